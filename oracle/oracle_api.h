/*
 * oracle_api.h -- C API shared by the two CPU checkers under oracle/.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the libraries that implement this API; the product
 * (polaris_amd/, include/polaris_hip.h) never does.
 *
 *   libpolaris_oracle.so   (oracle/polaris_oracle.cpp)  prefix polaris_oracle_
 *        the CPU restatement of the reference path; travels to the GPU box.
 *   libpolaris_ref_pm.so / libpolaris_ref_libm.so  (oracle/refbuild)  prefix polaris_ref_
 *        the reference's own OpenCL C compiled for the host; exists only where
 *        /root/reference exists.
 *
 * Both expose the same entry points so a test can drive either through one binding.
 */
#ifndef POLARIS_ORACLE_API_H
#define POLARIS_ORACLE_API_H

#include "polaris_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Optional taps into the state of ONE sample (index tap_sample) of a trace call.
 * Any pointer may be NULL.  N = frame_w * block_h. */
typedef struct PolarisOracleTaps {
	uint32_t tap_sample;
	float *primary_rays;    /* [N][8]: origin.xyz, maxDist, dir.xyz, (float)pathIndex   */
	int32_t *primary_hit;   /* [N]   : hit flag of the primary query                    */
	float *primary_wuvt;    /* [N][4]: barycentrics w,u,v and distance t (hits only)    */
	int32_t *primary_tri;   /* [N][2]: mesh instance, triangle index (hits only)        */
	float *throughput0;     /* [N][4]: path throughput after the first shade step       */
	int32_t *num_rays;      /* [2*bounces]: per bounce b of the tapped sample:
	                           [2b] = active rays entering shade, [2b+1] = occlusion rays */
} PolarisOracleTaps;

/* flags for *_trace */
#define POLARIS_ORACLE_FIX_EMITTER_INDEX 1u /* index direct emitter hits by pixelIndex
                                                instead of the block-local
                                               path index of pt_integrator.cl:106 */
#define POLARIS_ORACLE_SERIAL 2u            /* do not use OpenMP (restatement only)  */
#define POLARIS_ORACLE_PARALLEL_SAMPLES 4u  /* restatement only, CPU-baseline mode: threads take whole
                                               samples (private buffers), per-pixel sums re-associated */

/*
 * One tracer.Trace call (tracer/opencl/tracer.go:194-247 + pipeline.go:94-213) on a single
 * block:  clears trace_accum (frame_w*frame_h float4, stride 4 floats), then for every
 * sample s < samples_per_pixel runs generatePrimaryRays with seeds[s*(1+B)] and the bounce
 * loop with seeds[s*(1+B)+1+b]  (B = num_bounces).
 * Returns 0 on success.
 */
#define POLARIS_ORACLE_DECL(prefix)                                                          \
	int prefix##_trace(const PolarisSceneView *scene, const float eye[3],                    \
	                   const float frustum[16], const PolarisBlockRequest *req,              \
	                   const uint32_t *seeds, size_t n_seeds, float *trace_accum,            \
	                   PolarisTraceStats *stats, const PolarisOracleTaps *taps,              \
	                   uint32_t flags);                                                      \
	/* tonemapSimpleReinhard (kernels/hdr.cl:5-28) over n pixels */                           \
	int prefix##_tonemap(const float *accum, uint32_t n_pixels, float sample_weight,          \
	                     float exposure, uint8_t *rgba);                                      \
	/* randomGetSample2f (samplers/random_sampler.cl:7-16): advances state, writes 2 draws */ \
	void prefix##_random(uint32_t state[2], float out[2]);                                    \
	/* BxDF probe: out[0..2]=bxdfGetSample value, out[3..5]=sampled dir, out[6]=pdf,          \
	 * out[7]=bxdfGetPdf(eval_dir), out[8..10]=bxdfEval(eval_dir) (bxdf/bxdf.cl:31-105) */    \
	void prefix##_bxdf_probe(const PolarisMaterialNode *node,                                 \
	                         const PolarisTextureMetadata *tex_meta, const uint8_t *tex_data, \
	                         const float normal[3], const float uv[2], const float in_dir[3], \
	                         const float sample[2], const float eval_dir[3], float out[11]);  \
	/* texGetSample3f / texGetSample1f / texGetBumpSample3f (texture_sampler.cl:14-252):      \
	 * out[0..2], out[3], out[4..6] */                                                        \
	void prefix##_tex_probe(const PolarisTextureMetadata *tex_meta, const uint8_t *tex_data,  \
	                        int32_t tex_index, const float uv[2], float out[7]);              \
	/* emissiveGetSample / emissiveGetPdf (emissive_sampler.cl:176-223) from a surface point: \
	 * out[0..2]=sample, out[3..5]=dir, out[6]=pdf, out[7]=dist, out[8]=emissiveGetPdf(pdf_dir) */ \
	void prefix##_emissive_probe(const PolarisSceneView *scene, uint32_t emissive_index,      \
	                             const float point[3], const float normal[3],                 \
	                             const float sample[2], const float pdf_dir[3], float out[9]); \
	/* matSelectNode (samplers/material_sampler.cl:21-95) from material node `root` of the scene, with the path's dispersion    \
	 * flags and the shading PRNG state as they stand: out[0] = type of the selected leaf (bits), out[1..2] = its int / ext IOR \
	 * after the dispersion override, out[3..5] = surface normal after bump / normal maps, out[6..8] = tint, out[9] = path      \
	 * flags (bits), out[10..11] = PRNG state (bits), out[12..14] / [15..17] = the leaf's k / t vectors (which leaf was chosen) */ \
	void prefix##_material_probe(const PolarisSceneView *scene, uint32_t root, const float normal[3], const float uv[2],      \
	                             const uint32_t rng_state[2], uint32_t path_flags, float out[18]);                           \
	/* rayIntersectionQuery (any_hit = 0, intersect.cl:184-347) or rayIntersectionTest (any_hit \
	 * != 0, :26-180) over n arbitrary rays [n][8] = origin.xyz, maxDist, dir.xyz, unused:       \
	 * hit[n]; for closest hits also wuvt[n][4] and inst_tri[n][2] (either may be NULL) */       \
	int prefix##_intersect_probe(const PolarisSceneView *scene, const float *rays, uint32_t n,    \
	                             int any_hit, int32_t *hit, float *wuvt, int32_t *inst_tri);      \
	const char *prefix##_describe(void);

POLARIS_ORACLE_DECL(polaris_oracle)
POLARIS_ORACLE_DECL(polaris_ref)

#ifdef __cplusplus
}
#endif
#endif
