/*
 * polaris_types.h -- the data contract of the polaris tracer path, as plain C structs.
 *
 * Every struct below is byte-for-byte the layout the reference uploads to its device
 * buffers; citations are file:line under the reference tree (achilleasa/polaris):
 *
 *   PolarisBvhNode          asset/scene/optimized_scene.go:25-31   == CL/types.cl:26-48
 *   PolarisMeshInstance     asset/scene/optimized_scene.go:137-149 == CL/types.cl:50-65
 *   PolarisMaterialNode     asset/scene/optimized_scene.go:79-107  == CL/types.cl:103-163
 *   PolarisEmissive         asset/scene/optimized_scene.go:118-133 == CL/types.cl:165-186
 *   PolarisTextureMetadata  asset/scene/optimized_scene.go:152-163 == CL/types.cl:92-101
 *   PolarisSceneView        asset/scene/optimized_scene.go:167-190 (the 10 flat slices +
 *                           the two scene-global material indices)
 *   PolarisBlockRequest     tracer/tracer.go:6-34 (same 12 fields, same order)
 *
 * A Go caller can therefore pass &slice[0] / len(slice) of scene.Scene's slices and a
 * *tracer.BlockRequest reinterpretation straight through cgo.
 */
#ifndef POLARIS_TYPES_H
#define POLARIS_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BVH node, 32 bytes.
 *   inner node     : ldata > 0, rdata > 0  = left / right child node indices
 *   top-level leaf : ldata <= 0 = -(mesh instance index), rdata == 0
 *   bottom leaf    : ldata <= 0 = -(first triangle index), rdata > 0 = triangle count
 * (optimized_scene.go:14-64, CL/kernels/intersect.cl:6-11) */
typedef struct PolarisBvhNode {
	float min[3];
	int32_t ldata;
	float max[3];
	int32_t rdata;
} PolarisBvhNode;

/* Mesh instance, 80 bytes. `inv_transform` is the INVERSE instance matrix, column major
 * (compiler.go:185-192); rays are taken to mesh space with it (intersect.cl:88-89). */
typedef struct PolarisMeshInstance {
	uint32_t mesh_index;
	uint32_t bvh_root;
	uint32_t reserved[2];
	float inv_transform[16];
} PolarisMeshInstance;

/* Layered-material tree node, 64 bytes (types.cl:103-163).  Field use depends on `type`:
 * BxDF leaves (bxdf.cl:13-18): emissive 2, diffuse 4, conductor 8, rough conductor 16,
 * dielectric 32, rough dielectric 64; operators (material_sampler.cl:4-8): mix 10001,
 * mixMap 10002, bumpMap 10003, normalMap 10004, disperse 10005. */
typedef struct PolarisMaterialNode {
	uint32_t type;
	uint32_t left_child;
	int32_t right_child;   /* or transmittance texture index            */
	int32_t tex;           /* bump / mix-weight / reflectance / specularity / radiance texture */
	float k[4];            /* reflectance | specularity | radiance | int dispersion IORs; k[0] = mix weight */
	float t[4];            /* transmittance | ext dispersion IORs       */
	float int_ior;
	float ext_ior;
	float scale;           /* radiance scale | roughness                */
	int32_t roughness_tex;
} PolarisMaterialNode;

#define POLARIS_BXDF_INVALID          0u
#define POLARIS_BXDF_EMISSIVE         2u
#define POLARIS_BXDF_DIFFUSE          4u
#define POLARIS_BXDF_CONDUCTOR        8u
#define POLARIS_BXDF_ROUGH_CONDUCTOR  16u
#define POLARIS_BXDF_DIELECTRIC       32u
#define POLARIS_BXDF_ROUGH_DIELECTRIC 64u
#define POLARIS_MAT_OP_MIX            10001u
#define POLARIS_MAT_OP_MIX_MAP        10002u
#define POLARIS_MAT_OP_BUMP_MAP       10003u
#define POLARIS_MAT_OP_NORMAL_MAP     10004u
#define POLARIS_MAT_OP_DISPERSE       10005u

/* Emissive primitive, 80 bytes (types.cl:165-186). */
typedef struct PolarisEmissive {
	float transform[16];   /* column major; the owning instance's (inverse) matrix, compiler.go:208 */
	float area;
	uint32_t tri_index;
	uint32_t mat_node_index;
	uint32_t type;         /* 0 = area light, 1 = environment light (emissive_sampler.cl:4-5) */
} PolarisEmissive;

#define POLARIS_EMISSIVE_AREA 0u
#define POLARIS_EMISSIVE_ENVIRONMENT 1u

/* Texture metadata, 16 bytes; formats texture_sampler.cl:4-7. */
typedef struct PolarisTextureMetadata {
	uint32_t format;       /* 0 L8, 1 L32F, 2 RGBA8, 3 RGBA32F */
	uint32_t width;
	uint32_t height;
	uint32_t data_offset;  /* byte offset into the texture blob */
} PolarisTextureMetadata;

#define POLARIS_TEX_L8 0u
#define POLARIS_TEX_L32F 1u
#define POLARIS_TEX_RGBA8 2u
#define POLARIS_TEX_RGBA32F 3u

/* A borrowed view of scene.Scene's flat arrays (optimized_scene.go:167-190).  Nothing is
 * retained after the call that receives it returns. */
typedef struct PolarisSceneView {
	const PolarisBvhNode *bvh_nodes;          uint32_t num_bvh_nodes;
	const PolarisMeshInstance *mesh_instances; uint32_t num_mesh_instances;
	const PolarisMaterialNode *material_nodes; uint32_t num_material_nodes;
	const PolarisEmissive *emissives;         uint32_t num_emissives;
	const uint8_t *texture_data;              uint32_t texture_data_bytes;
	const PolarisTextureMetadata *texture_meta; uint32_t num_textures;
	const float *vertices;                    /* float4 per vertex, 3 per triangle */
	const float *normals;                     /* float4 per vertex                 */
	const float *uvs;                         /* float2 per vertex                 */
	const uint32_t *material_index;           /* root material node per triangle   */
	uint32_t num_triangles;
	int32_t scene_diffuse_mat_index;          /* -1 = none (pipeline.go:134)       */
	int32_t scene_emissive_mat_index;
} PolarisSceneView;

/* tracer.BlockRequest (tracer/tracer.go:6-34), same field order and widths. */
typedef struct PolarisBlockRequest {
	uint32_t frame_w, frame_h;
	uint32_t block_x, block_y, block_w, block_h;
	uint32_t samples_per_pixel;
	uint32_t num_bounces;
	uint32_t min_bounces_for_rr;
	float exposure;
	uint32_t seed;
	uint32_t accumulated_samples;
} PolarisBlockRequest;

#define POLARIS_MAX_BOUNCES 32

/* Ray and shading-event counters of one Trace call.  The counting rule is BASELINE.md
 * section 3 / SURVEY.md section 8d: rays = rays actually handed to an intersection kernel. */
typedef struct PolarisTraceStats {
	uint64_t primary_rays;        /* closest-hit queries on camera rays             */
	uint64_t indirect_rays;       /* closest-hit queries on bounce rays             */
	uint64_t occlusion_rays;      /* any-hit tests                                  */
	uint64_t shaded_hits;         /* hits entering the BxDF path of shadeHits       */
	uint64_t shaded_misses;       /* misses shaded against a background material    */
	uint64_t emitter_hits;        /* front-facing hits on an emissive (path ends)   */
	uint64_t unoccluded;          /* occlusion rays that reached the light          */
	/* per bounce, summed over samples: rays[b] = closest-hit queries issued for bounce b
	 * (b = 0: primaries), occl[b] = occlusion rays emitted by the shade step of bounce b */
	uint64_t rays_per_bounce[POLARIS_MAX_BOUNCES];
	uint64_t occl_per_bounce[POLARIS_MAX_BOUNCES];
	double device_ms;             /* device time of the call (HIP events); CPU: wall */
} PolarisTraceStats;

#ifdef __cplusplus
}
#endif
#endif /* POLARIS_TYPES_H */
