// ref_driver.cpp -- replays the reference's HOST sequencing around the reference's own
// device code, which is compiled in place from /root/reference/tracer/opencl/CL/main.cl
// (see Makefile in this directory; no reference source is copied into this repository).
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  Nothing in the product links this.
//
// What is restated here (this file is the only part of oracle/_ref that is OUR code, next to
// cl_builtins.cpp) is the launch order of
//     tracer/opencl/tracer.go:194-247   (Tracer.Trace: clear, per-sample loop)
//     tracer/opencl/pipeline.go:94-213  (MonteCarloIntegrator: bounce loop)
//     tracer/opencl/resources.go:127-360 (argument binding, the two counter resets before
//                                         shadeHits at :230-238)
// with the host PRNG (Go math/rand) replaced by an explicit seed list.  Work-items execute
// in ascending global id with work-group size 1 (pipeline.go:105-106).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle_api.h"

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint2 __attribute__((ext_vector_type(2)));

extern thread_local size_t polaris_ref_gid[3];

// Device-side structs of CL/types.cl as the host sees them (sizes checked below).
struct Ray { float origin[4]; float dir[4]; };                                   // types.cl:4-10
struct Path { float throughput[4]; uint32_t pixelIndex, flags, r1, r2; };        // types.cl:12-24
struct Intersection { float wuvt[4]; uint32_t meshInstance, triIndex, r1, r2; }; // types.cl:67-79
struct Surface { float3 point; float3 normal; float2 uv; uint32_t matNodeIndex; }; // types.cl:81-90
static_assert(sizeof(Ray) == 32 && sizeof(Path) == 32 && sizeof(Intersection) == 32, "layout");
static_assert(sizeof(Surface) == 48 && sizeof(PolarisMaterialNode) == 64, "layout");
static_assert(sizeof(PolarisEmissive) == 80 && sizeof(PolarisMeshInstance) == 80, "layout");
static_assert(sizeof(PolarisBvhNode) == 32 && sizeof(PolarisTextureMetadata) == 16, "layout");

// The reference's kernels (plain C symbol names in the compiled object).
extern "C" {
void generatePrimaryRays(void *rays, int *numRays, void *paths, float4 fTL, float4 fTR, float4 fBL,
                         float4 fBR, float3 eyePos, float2 texelDims, uint32_t blockY, uint32_t blockH,
                         uint32_t frameW, uint32_t frameH, uint32_t randSeed);
void rayIntersectionQuery(void *rays, const int *numRays, const void *bvhNodes, const void *meshInstances,
                          const void *vertexList, int *hitFlag, void *intersections);
void rayIntersectionTest(void *rays, const int *numRays, const void *bvhNodes, const void *meshInstances,
                         const void *vertexList, int *hitFlag);
void shadeHits(void *rays, const int *numRays, void *paths, void *hitFlags, void *intersections,
               const void *vertices, const void *normals, const void *uv, const void *materialIndices,
               const void *materialNodes, const void *emissives, uint32_t numEmissives, const void *texMeta,
               const void *texData, uint32_t bounce, uint32_t minBouncesForRR, uint32_t randSeed,
               void *occlusionRays, volatile int *numOcclusionRays, void *emissiveSamples, void *indirectRays,
               volatile int *numIndirectRays, void *accumulator);
void shadePrimaryRayMisses(void *rays, const int *numRays, void *paths, void *hitFlags, const void *materialNodes,
                           uint32_t bgIndex, const void *texMeta, const void *texData, void *accumulator);
void shadeIndirectRayMisses(void *rays, const int *numRays, void *paths, void *hitFlags, const void *materialNodes,
                            uint32_t bgIndex, const void *texMeta, const void *texData, void *accumulator);
void accumulateEmissiveSamples(void *rays, const int *numRays, void *paths, void *hitFlags, void *emissiveSamples,
                               void *accumulator);
void tonemapSimpleReinhard(const void *accumulator, void *paths, void *frameBuffer, float sampleWeight, float exposure);
}

// Reference helper functions (Itanium-mangled with address-space qualifiers).
float2 ref_randomGetSample2f(uint2 *state) __asm__("_Z17randomGetSample2fPU9CLgenericDv2_j");
float3 ref_bxdfGetSample(Surface *, PolarisMaterialNode *, const void *texMeta, const void *texData, float2 rnd,
                         float3 inDir, float3 *outDir, float *pdf)
    __asm__("_Z13bxdfGetSamplePU9CLgeneric7SurfacePU9CLgeneric12MaterialNodePU8CLglobal15TextureMetadataPU8CLglobalhDv2_fDv3_fPU9CLgenericSB_PU9CLgenericf");
float ref_bxdfGetPdf(Surface *, PolarisMaterialNode *, const void *texMeta, const void *texData, float3 inDir, float3 outDir)
    __asm__("_Z10bxdfGetPdfPU9CLgeneric7SurfacePU9CLgeneric12MaterialNodePU8CLglobal15TextureMetadataPU8CLglobalhDv3_fSA_");
float3 ref_bxdfEval(Surface *, PolarisMaterialNode *, const void *texMeta, const void *texData, float3 inDir, float3 outDir)
    __asm__("_Z8bxdfEvalPU9CLgeneric7SurfacePU9CLgeneric12MaterialNodePU8CLglobal15TextureMetadataPU8CLglobalhDv3_fSA_");
void ref_matSelectNode(Path *path, Surface *surface, float3 inRayDir, PolarisMaterialNode *selected, float3 *tint, const void *materialNodes,
                       uint2 *rndState, const void *texMeta, const void *texData)
    __asm__("_Z13matSelectNodePU8CLglobal4PathPU9CLgeneric7SurfaceDv3_fPU9CLgeneric12MaterialNodePU9CLgenericS5_PU8CLglobalS6_PU9CLgenericDv2_jPU8CLglobal15TextureMetadataPU8CLglobalh");
float3 ref_texGetSample3f(float2 uv, int texIndex, const void *texMeta, const void *texData)
    __asm__("_Z14texGetSample3fDv2_fiPU8CLglobal15TextureMetadataPU8CLglobalh");
float ref_texGetSample1f(float2 uv, int texIndex, const void *texMeta, const void *texData)
    __asm__("_Z14texGetSample1fDv2_fiPU8CLglobal15TextureMetadataPU8CLglobalh");
float3 ref_texGetBumpSample3f(float2 uv, int texIndex, const void *texMeta, const void *texData)
    __asm__("_Z18texGetBumpSample3fDv2_fiPU8CLglobal15TextureMetadataPU8CLglobalh");
float3 ref_emissiveGetSample(Surface *, const void *emissive, const void *vertices, const void *normals, const void *uv,
                             const void *materialNodes, const void *texMeta, const void *texData, float2 rnd,
                             float3 *outDir, float *pdf, float *dist)
    __asm__("_Z17emissiveGetSamplePU9CLgeneric7SurfacePU8CLglobal8EmissivePU8CLglobalDv4_fS7_PU8CLglobalDv2_fPU8CLglobal12MaterialNodePU8CLglobal15TextureMetadataPU8CLglobalhS8_PU9CLgenericDv3_fPU9CLgenericfSN_");
float ref_emissiveGetPdf(Surface *, const void *emissive, const void *vertices, const void *normals, const void *uv,
                         const void *materialNodes, const void *texMeta, const void *texData, float3 outDir)
    __asm__("_Z14emissiveGetPdfPU9CLgeneric7SurfacePU8CLglobal8EmissivePU8CLglobalDv4_fS7_PU8CLglobalDv2_fPU8CLglobal12MaterialNodePU8CLglobal15TextureMetadataPU8CLglobalhDv3_f");

static inline float4 f4(const float *p) { float4 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3]; return v; }
static inline float3 f3(const float *p) { float3 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; return v; }

extern "C" int polaris_ref_trace(const PolarisSceneView *sc, const float eye[3], const float frustum[16],
                                 const PolarisBlockRequest *req, const uint32_t *seeds, size_t n_seeds,
                                 float *trace_accum, PolarisTraceStats *stats, const PolarisOracleTaps *taps,
                                 uint32_t flags) {
	const uint32_t W = req->frame_w, H = req->frame_h, BH = req->block_h, BY = req->block_y;
	const uint32_t B = req->num_bounces, spp = req->samples_per_pixel;
	if (!sc || !req || !trace_accum || W == 0 || BH == 0 || BY + BH > H || B > POLARIS_MAX_BOUNCES) return 1;
	if (n_seeds < (size_t)spp * (1 + B)) return 2;
	const int N = (int)(W * BH);
	const size_t F = (size_t)W * H;

	std::vector<Ray> rays[3];
	for (auto &r : rays) r.resize(N);
	std::vector<Path> paths(N);
	std::vector<int> hitFlags(N);
	std::vector<Intersection> isects(N);
	std::vector<float> emissiveSamples((size_t)N * 4);
	int counters[3] = {0, 0, 0};
	PolarisTraceStats st;
	memset(&st, 0, sizeof st);

	// ClearTraceAccumulator (tracer.go:215, accumulator.cl:5-9)
	memset(trace_accum, 0, F * 4 * sizeof(float));
	// shadeHits indexes emitter hits by the block-local path index
	// (pt_integrator.cl:106).  Handing that kernel an accumulator pointer advanced by
	// block_y rows makes the same store land on pixelIndex without touching the source.
	float *shadeAccum = trace_accum + ((flags & POLARIS_ORACLE_FIX_EMITTER_INDEX) ? (size_t)BY * W * 4 : 0);

	const float2 texel = {1.0f / (float)W, 1.0f / (float)H}; // resources.go:130-133
	const int bg = sc->scene_diffuse_mat_index;

	for (uint32_t s = 0; s < spp; s++) {
		const uint32_t *sseed = seeds + (size_t)s * (1 + B);
		const bool tap = taps && taps->tap_sample == s;
		// PrimaryRayGenerator (pipeline.go:80-84, resources.go:127-156), 2-D NDRange (W, BH)
		for (uint32_t y = 0; y < BH; y++)
			for (uint32_t x = 0; x < W; x++) {
				polaris_ref_gid[0] = x;
				polaris_ref_gid[1] = y;
				generatePrimaryRays(rays[0].data(), &counters[0], paths.data(), f4(frustum), f4(frustum + 4),
				                    f4(frustum + 8), f4(frustum + 12), f3(eye), texel, BY, BH, W, H, sseed[0]);
			}
		polaris_ref_gid[1] = 0;
		st.primary_rays += (uint64_t)counters[0];
		if (tap && taps->primary_rays) memcpy(taps->primary_rays, rays[0].data(), (size_t)N * sizeof(Ray));

		uint32_t cur = 0;
		// primary query: the CPU-device branch of pipeline.go:107-111
		{
			const int n = counters[cur];
#pragma omp parallel for schedule(dynamic, 256)
			for (int g = 0; g < n; g++) {
				polaris_ref_gid[0] = (size_t)g;
				rayIntersectionQuery(rays[cur].data(), &counters[cur], sc->bvh_nodes, sc->mesh_instances, sc->vertices,
				                     hitFlags.data(), isects.data());
			}
		}
		if (tap) {
			for (int g = 0; g < N; g++) {
				if (taps->primary_hit) taps->primary_hit[g] = hitFlags[g];
				if (hitFlags[g]) {
					if (taps->primary_wuvt) memcpy(taps->primary_wuvt + 4 * g, isects[g].wuvt, 16);
					if (taps->primary_tri) {
						taps->primary_tri[2 * g] = (int32_t)isects[g].meshInstance;
						taps->primary_tri[2 * g + 1] = (int32_t)isects[g].triIndex;
					}
				}
			}
		}

		for (uint32_t b = 0; b < B; b++) {
			const int n = counters[cur];
			st.rays_per_bounce[b] += (uint64_t)n;
			if (b > 0) st.indirect_rays += (uint64_t)n;
			int nhit = 0;
			for (int g = 0; g < n; g++) nhit += hitFlags[g] ? 1 : 0;
			st.shaded_hits += (uint64_t)nhit;
			// miss shading (pipeline.go:134-143)
			if (bg != -1) {
				st.shaded_misses += (uint64_t)(n - nhit);
				for (int g = 0; g < n; g++) {
					polaris_ref_gid[0] = (size_t)g;
					if (b == 0)
						shadePrimaryRayMisses(rays[cur].data(), &counters[cur], paths.data(), hitFlags.data(),
						                      sc->material_nodes, (uint32_t)bg, sc->texture_meta, sc->texture_data, trace_accum);
					else
						shadeIndirectRayMisses(rays[cur].data(), &counters[cur], paths.data(), hitFlags.data(),
						                       sc->material_nodes, (uint32_t)bg, sc->texture_meta, sc->texture_data, trace_accum);
				}
			}
			// ShadeHits (resources.go:226-273): the two counter resets, then the kernel
			counters[2] = 0;
			counters[1 - cur] = 0;
			for (int g = 0; g < n; g++) {
				polaris_ref_gid[0] = (size_t)g;
				shadeHits(rays[cur].data(), &counters[cur], paths.data(), hitFlags.data(), isects.data(), sc->vertices,
				          sc->normals, sc->uvs, sc->material_index, sc->material_nodes, sc->emissives, sc->num_emissives,
				          sc->texture_meta, sc->texture_data, b, req->min_bounces_for_rr, sseed[1 + b], rays[2].data(),
				          &counters[2], emissiveSamples.data(), rays[1 - cur].data(), &counters[1 - cur], shadeAccum);
			}
			if (tap && b == 0 && taps->throughput0)
				for (int g = 0; g < N; g++) memcpy(taps->throughput0 + 4 * g, paths[g].throughput, 16);
			if (tap && taps->num_rays) {
				taps->num_rays[2 * b] = n;
				taps->num_rays[2 * b + 1] = counters[2];
			}
			// occlusion rays (pipeline.go:160-168)
			const int nocc = counters[2];
			st.occl_per_bounce[b] += (uint64_t)nocc;
			st.occlusion_rays += (uint64_t)nocc;
#pragma omp parallel for schedule(dynamic, 256)
			for (int g = 0; g < nocc; g++) {
				polaris_ref_gid[0] = (size_t)g;
				rayIntersectionTest(rays[2].data(), &counters[2], sc->bvh_nodes, sc->mesh_instances, sc->vertices,
				                    hitFlags.data());
			}
			for (int g = 0; g < nocc; g++) {
				polaris_ref_gid[0] = (size_t)g;
				if (!hitFlags[g]) st.unoccluded++;
				accumulateEmissiveSamples(rays[2].data(), &counters[2], paths.data(), hitFlags.data(), emissiveSamples.data(),
				                          trace_accum);
			}
			// next bounce (pipeline.go:203-208)
			if (b + 1 < B) {
				cur = 1 - cur;
				const int m = counters[cur];
#pragma omp parallel for schedule(dynamic, 256)
				for (int g = 0; g < m; g++) {
					polaris_ref_gid[0] = (size_t)g;
					rayIntersectionQuery(rays[cur].data(), &counters[cur], sc->bvh_nodes, sc->mesh_instances, sc->vertices,
					                     hitFlags.data(), isects.data());
				}
			}
		}
	}
	polaris_ref_gid[0] = 0;
	if (stats) *stats = st;
	return 0;
}

extern "C" int polaris_ref_tonemap(const float *accum, uint32_t n_pixels, float sample_weight, float exposure,
                                   uint8_t *rgba) {
	for (uint32_t g = 0; g < n_pixels; g++) {
		polaris_ref_gid[0] = g;
		tonemapSimpleReinhard(accum, nullptr, rgba, sample_weight, exposure);
	}
	polaris_ref_gid[0] = 0;
	return 0;
}

extern "C" void polaris_ref_random(uint32_t state[2], float out[2]) {
	uint2 st = {state[0], state[1]};
	float2 r = ref_randomGetSample2f(&st);
	state[0] = st.x;
	state[1] = st.y;
	out[0] = r.x;
	out[1] = r.y;
}

extern "C" int polaris_ref_intersect_probe(const PolarisSceneView *sc, const float *rays, uint32_t n, int any_hit, int32_t *hit,
                                           float *wuvt, int32_t *inst_tri) {
	std::vector<Ray> r(n);
	memcpy(r.data(), rays, (size_t)n * sizeof(Ray));
	std::vector<int> flags(n, 0);
	std::vector<Intersection> isects(any_hit ? 0 : n);
	int count = (int)n;
#pragma omp parallel for schedule(dynamic, 256)
	for (long g = 0; g < (long)n; g++) {
		polaris_ref_gid[0] = (size_t)g;
		if (any_hit) rayIntersectionTest(r.data(), &count, sc->bvh_nodes, sc->mesh_instances, sc->vertices, flags.data());
		else rayIntersectionQuery(r.data(), &count, sc->bvh_nodes, sc->mesh_instances, sc->vertices, flags.data(), isects.data());
	}
	for (uint32_t i = 0; i < n; i++) {
		hit[i] = flags[i];
		if (!any_hit && flags[i]) {
			if (wuvt) memcpy(wuvt + 4 * (size_t)i, &isects[i], 16);
			if (inst_tri) memcpy(inst_tri + 2 * (size_t)i, (const char *)&isects[i] + 16, 8);
		}
	}
	return 0;
}

extern "C" void polaris_ref_bxdf_probe(const PolarisMaterialNode *node, const PolarisTextureMetadata *tex_meta,
                                       const uint8_t *tex_data, const float normal[3], const float uv[2],
                                       const float in_dir[3], const float sample[2], const float eval_dir[3],
                                       float out[11]) {
	Surface sf;
	sf.point = (float3){0.0f, 0.0f, 0.0f};
	sf.normal = f3(normal);
	sf.uv = (float2){uv[0], uv[1]};
	sf.matNodeIndex = 0;
	PolarisMaterialNode m = *node;
	float3 od = {0, 0, 0};
	float pdf = 0.0f;
	float3 v = ref_bxdfGetSample(&sf, &m, tex_meta, tex_data, (float2){sample[0], sample[1]}, f3(in_dir), &od, &pdf);
	out[0] = v.x; out[1] = v.y; out[2] = v.z;
	out[3] = od.x; out[4] = od.y; out[5] = od.z;
	out[6] = pdf;
	out[7] = ref_bxdfGetPdf(&sf, &m, tex_meta, tex_data, f3(in_dir), f3(eval_dir));
	float3 e = ref_bxdfEval(&sf, &m, tex_meta, tex_data, f3(in_dir), f3(eval_dir));
	out[8] = e.x; out[9] = e.y; out[10] = e.z;
}

extern "C" void polaris_ref_material_probe(const PolarisSceneView *sc, uint32_t root, const float normal[3], const float uv[2],
                                           const uint32_t rng_state[2], uint32_t path_flags, float out[18]) {
	Path path;
	memset(&path, 0, sizeof path);
	path.flags = path_flags;
	Surface sf;
	sf.point = (float3){0.0f, 0.0f, 0.0f};
	sf.normal = f3(normal);
	sf.uv = (float2){uv[0], uv[1]};
	sf.matNodeIndex = root;
	PolarisMaterialNode m;
	memset(&m, 0, sizeof m);
	float3 tint = {1.0f, 1.0f, 1.0f};
	uint2 rnd = {rng_state[0], rng_state[1]};
	ref_matSelectNode(&path, &sf, (float3){0.0f, 0.0f, 0.0f}, &m, &tint, sc->material_nodes, &rnd, sc->texture_meta, sc->texture_data);
	memcpy(&out[0], &m.type, 4);
	out[1] = m.int_ior; out[2] = m.ext_ior;
	out[3] = sf.normal.x; out[4] = sf.normal.y; out[5] = sf.normal.z;
	out[6] = tint.x; out[7] = tint.y; out[8] = tint.z;
	memcpy(&out[9], &path.flags, 4);
	uint32_t r0 = rnd.x, r1 = rnd.y;
	memcpy(&out[10], &r0, 4); memcpy(&out[11], &r1, 4);
	for (int k = 0; k < 3; k++) { out[12 + k] = m.k[k]; out[15 + k] = m.t[k]; }
}

extern "C" void polaris_ref_tex_probe(const PolarisTextureMetadata *tex_meta, const uint8_t *tex_data,
                                      int32_t tex_index, const float uv[2], float out[7]) {
	float2 u = {uv[0], uv[1]};
	float3 a = ref_texGetSample3f(u, tex_index, tex_meta, tex_data);
	out[0] = a.x; out[1] = a.y; out[2] = a.z;
	out[3] = ref_texGetSample1f(u, tex_index, tex_meta, tex_data);
	float3 b = ref_texGetBumpSample3f(u, tex_index, tex_meta, tex_data);
	out[4] = b.x; out[5] = b.y; out[6] = b.z;
}

extern "C" void polaris_ref_emissive_probe(const PolarisSceneView *sc, uint32_t emissive_index, const float point[3],
                                           const float normal[3], const float sample[2], const float pdf_dir[3],
                                           float out[9]) {
	Surface sf;
	sf.point = f3(point);
	sf.normal = f3(normal);
	sf.uv = (float2){0.0f, 0.0f};
	sf.matNodeIndex = 0;
	const PolarisEmissive *em = sc->emissives + emissive_index;
	float3 od = {0, 0, 0};
	float pdf = 0.0f, dist = 0.0f;
	float3 v = ref_emissiveGetSample(&sf, em, sc->vertices, sc->normals, sc->uvs, sc->material_nodes, sc->texture_meta,
	                                 sc->texture_data, (float2){sample[0], sample[1]}, &od, &pdf, &dist);
	out[0] = v.x; out[1] = v.y; out[2] = v.z;
	out[3] = od.x; out[4] = od.y; out[5] = od.z;
	out[6] = pdf;
	out[7] = dist;
	out[8] = ref_emissiveGetPdf(&sf, em, sc->vertices, sc->normals, sc->uvs, sc->material_nodes, sc->texture_meta,
	                            sc->texture_data, f3(pdf_dir));
}

extern "C" const char *polaris_ref_describe(void) {
#ifdef POLARIS_REF_LIBM
	return "reference OpenCL C (tracer/opencl/CL/main.cl) compiled for x86-64; built-ins: glibc libm";
#else
	return "reference OpenCL C (tracer/opencl/CL/main.cl) compiled for x86-64; built-ins: polaris_math.h";
#endif
}
