// kernels.h -- the gfx950 kernels of the polaris wavefront path tracer.
//
// Formulation (DESIGN.md has the full picture).  K samples of the block are traced as ONE
// wavefront batch: slot = sample * Npad + index, Npad = N rounded up to the 256-thread
// workgroup, so a workgroup never straddles two samples.  Every stream is a float4 array
// (one 16 B access per lane = the coalescing sweet spot of the guide):
//     ray_o  = origin.xyz | max distance          ray_d = direction.xyz | path word
//     thr    = path throughput.xyz | -            hit   = u | v | t | triangle (-1 = miss)
//     (inside a Trace ray_o and hit are 12-byte records -- origin.xyz, and u | v | triangle: Streams::o12, hit12)
//     occ_o / occ_d / occ_e = shadow ray origin|maxDist, dir|accumulator cell, NEE radiance|accumulator cell
//     vis    = one byte per shadow ray: 1 = unoccluded (batched mode: what the any-hit kernels write; k_fold_nee reads it)
//     lsum   = per-path radiance of this batch (resolved into the trace accumulator in
//              sample order at the end of the batch)
// path word = path index (24 bit, the reference keeps it as a float in dir.w,
// util/ray.cl:9-12) | dispersion flags << 24 (util/path.cl:4-6).
//
// Rays stay in their 256-slot chunk; their ORDER is data.  The rays a chunk emits are appended to the front of the same 256
// slots in whatever order its waves finish (cnt[wg] says how many are live); what the reference's stable compaction would
// have made of them is carried along: every indirect ray holds its parent's canonical index in thr.w, the chunk's 256-bit
// emit mask (Streams::emask) turns that into the ray's own canonical index, and a tiny segmented scan (k_scan) turns the
// per-chunk counts into pfx[wg], the position the chunk's first ray WOULD have in the reference's globally compacted
// buffer.  pfx + canonical index is therefore exactly the `globalId` the reference seeds the shading PRNG with
// (kernels/pt_integrator.cl:81) when its atomic compaction runs in work-item order -- without ever moving a ray out of
// its pixel neighbourhood, without a grid-wide dependency inside the shading kernel, and without host round trips.
// (Round 1 kept the physical order equal to the reference order; see k_shade for why that went.)
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "scene_layout.h"
#include "shading.h"

namespace pol {

constexpr int WG = 256;            // threads per workgroup (4 wave64)
constexpr int kExitMarker = (int)0x80000000;

// 1 / det of the Moeller-Trumbore test in three instructions (v_rcp_f32 + one Newton step) instead of the eleven of the
// correctly rounded division: bit-equal to 1.0f / x for EVERY float with 2^-126 <= |x| < 2^126 (all 2^32 patterns swept on the
// GPU: polaris_hip_selftest_rcp, tests/test_gpu_probes.py).  Outside that interval the two differ (zeros, infinities,
// denormal x, denormal results) -- the triangle tests only consume the quotient when |det| >= kEps, and the scene validation
// bounds the coordinates so that |det| stays far below 2^126 (scene_layout.h, kMaxCoordinate).
__device__ __forceinline__ float rcp_det(float x) {
	const float r = __builtin_amdgcn_rcpf(x);
	return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}


// native_recip(ray.dir) of intersect.cl:302 as the correctly rounded 1 / x (polaris_math.h pm_rcp) for the three components of
// a direction: where every lane of the wave has all three magnitudes inside [2^-126, 2^126) -- i.e. always, except for rays that
// run exactly along an axis plane -- the three-instruction form above is that quotient, bit for bit (the same sweep); otherwise
// the wave takes the division.  (A NaN component slips through the min / max: both forms return a NaN for it, and no NaN's
// payload is ever looked at.)
__device__ __forceinline__ void rcp_dir(float dx, float dy, float dz, float &ix, float &iy, float &iz) {
	const float ax = __builtin_fabsf(dx), ay = __builtin_fabsf(dy), az = __builtin_fabsf(dz);
	const float lo = __builtin_fminf(__builtin_fminf(ax, ay), az), hi = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
	const bool fast = lo >= 1.17549435e-38f && hi < 8.50705917e37f; // 2^-126, 2^126
	if (__ballot(!fast) == 0ull) { ix = rcp_det(dx); iy = rcp_det(dy); iz = rcp_det(dz); }
	else { ix = pm_rcp(dx); iy = pm_rcp(dy); iz = pm_rcp(dz); }
}

struct PairNode { float4 lo0, hi0, lo1, hi1; };  // .w of lo0/lo1 carry the child refs (int bits)
struct TriRec { float4 v0, e1, e2; };             // v0.w = DFS rank, e1.w = scene triangle index (uint bits)
struct InstRec { float4 r0, r1, r2; int4 meta; }; // meta.x = root ref, meta.y = rank

#ifndef POLARIS_LDS_TOP_NODES
#define POLARIS_LDS_TOP_NODES 64
#endif
constexpr int kLdsTopNodes = POLARIS_LDS_TOP_NODES;  // (64 x 64 B = 4 KB: with the 16 KB stack a CU holds 8 workgroups) // pair records (8 KB) of the top of the tree staged in LDS per workgroup

struct BvhDev {
	const PairNode *pairs; // inner nodes in breadth-first order
	uint32_t num_pairs;
	const int2 *leaves;
	const TriRec *tris;
	const InstRec *insts;
	int root_ref;
	// Single-instance scenes (the top-level tree is ONE instance leaf): that instance's record travels by value, i.e. in
	// scalar registers, and k_trace enters it when it sets a ray up instead of chasing LeafInfo -> InstRec through two
	// dependent global loads per ray.  root_is_instance = 0 otherwise.
	int root_is_instance;
	InstRec root_inst;
	// Tiny-scene mode (kNodesLdsAll): the workgroup's dynamic LDS block, sized by the host for the uploaded scene (polaris_hip.hip,
	// plan_tiny_lds; layout in k_trace).
	uint32_t tiny_stack_off, lds_tris; // byte offset of the first real stack row (>= one row); triangle slots kept in LDS
};

struct Streams {
	float4 *ray_o, *ray_d, *thr, *hit;
	float4 *occ_o, *occ_d, *occ_e;
	uint8_t *vis; // batched mode: visibility flag of the shadow ray in the same slot of occ_* (written for EVERY shadow ray: dense, whole sectors)
	float4 *lsum;
	uint32_t *cnt_ray, *cnt_occ, *pfx;
	uint32_t *wg_stat; // per workgroup, written by shade: hits | misses << 10 | emitter hits << 20 (summed by k_scan)
	// Canonical order (see k_shade).  Rays sit in their chunk in ANY order; what the reference's stable compaction would
	// have made of them is carried as data: every indirect ray holds its PARENT's canonical index (position within the
	// chunk in reference order) in thr.w, and emask[bounce parity][chunk][8] has bit c set when the parent with canonical
	// index c emitted a ray -- so a ray's own canonical index is the number of set bits below its parent's.
	uint32_t *emask[2];
	int *hit_inst; // optional (test tap): instance id per slot, may be null
	// hit12 != 0: the hit records are 12 bytes (u, v, triangle word), not 16 (u, v, t, triangle word) -- nothing in a Trace reads t
	// (shadeHits rebuilds the hit point from the barycentrics, intersect.cl:283-286 / pt_integrator.cl), so the traversal kernels
	// do not store it and the shade kernels do not load it: 8 B per ray less through HBM.  The taps and probes keep 16 (they return t).
	uint32_t hit12;
	// o12 != 0: the origins of the closest-hit rays (ray_o) are 12 bytes -- inside a Trace their max distance is FLT_MAX for every
	// one of them (camera rays, camera.cl; bounce rays, pt_integrator.cl:209), so it is neither stored nor loaded.  (Shadow rays
	// carry a real distance in occ_o.w; the probes and taps pass arbitrary ones: 16 bytes there.)
	uint32_t o12;
};
__device__ __forceinline__ float4 load_ray_o(const Streams &st, size_t slot) {
	if (st.o12) { const float *p = reinterpret_cast<const float *>(st.ray_o) + 3 * slot; return make_float4(p[0], p[1], p[2], 3.402823466e+38f); }
	return st.ray_o[slot];
}
__device__ __forceinline__ void store_ray_o(const Streams &st, size_t slot, float4 o) {
	if (st.o12) { float *p = reinterpret_cast<float *>(st.ray_o) + 3 * slot; p[0] = o.x; p[1] = o.y; p[2] = o.z; }
	else st.ray_o[slot] = o;
}
__device__ __forceinline__ void store_hit(const Streams &st, size_t slot, float u, float v, float t, int tri) {
	if (st.hit12) { float *h = reinterpret_cast<float *>(st.hit) + 3 * slot; h[0] = u; h[1] = v; h[2] = __int_as_float(tri); }
	else st.hit[slot] = make_float4(u, v, t, __int_as_float(tri));
}
__device__ __forceinline__ float4 load_hit(const Streams &st, size_t slot) {
	if (st.hit12) { const float *h = reinterpret_cast<const float *>(st.hit) + 3 * slot; return make_float4(h[0], h[1], 0.0f, h[2]); }
	return st.hit[slot];
}

// device-side counters of one Trace call (mirrors PolarisTraceStats, all uint64)
// (shaded hits / misses / emitter hits are kept per bounce: the totals of PolarisTraceStats are their sums, and bench.py prices
// every shade launch -- i.e. every kernel symbol -- with its own counts, polaris_hip_shade_counts)
enum StatSlot { ST_UNOCCLUDED = 0, ST_RAYS_BOUNCE, ST_OCCL_BOUNCE = ST_RAYS_BOUNCE + POLARIS_MAX_BOUNCES, ST_HITS_BOUNCE = ST_OCCL_BOUNCE + POLARIS_MAX_BOUNCES,
                ST_MISSES_BOUNCE = ST_HITS_BOUNCE + POLARIS_MAX_BOUNCES, ST_EMITTERS_BOUNCE = ST_MISSES_BOUNCE + POLARIS_MAX_BOUNCES,
                ST_DEBUG = ST_EMITTERS_BOUNCE + POLARIS_MAX_BOUNCES, ST_COUNT = ST_DEBUG + 32 };

__device__ __forceinline__ int fbits(float f) { return __float_as_int(f); }
__device__ __forceinline__ float ibits(int i) { return __int_as_float(i); }

// ------------------------------------------------------------------------------------------
// generatePrimaryRays, kernels/camera.cl:5-58  (+ the per-workgroup bookkeeping of this design)
// ------------------------------------------------------------------------------------------
struct CameraArgs { float4 tl, tr, bl, br; float3 eye; float2 texel; };

__global__ __launch_bounds__(WG) void k_generate(Streams st, CameraArgs cam, const uint32_t *seeds, uint32_t seed_stride,
                                                  uint32_t first_sample, uint32_t N, uint32_t Npad, uint32_t W, uint32_t blockY,
                                                  int zero_lsum, int write_origin) {
	const uint32_t wgs_per_sample = Npad / WG;
	const uint32_t s = blockIdx.x / wgs_per_sample;
	const uint32_t idx0 = (blockIdx.x % wgs_per_sample) * WG;
	const uint32_t idx = idx0 + threadIdx.x;
	const size_t slot = (size_t)blockIdx.x * WG + threadIdx.x;
	if (threadIdx.x == 0) {
		st.cnt_ray[blockIdx.x] = idx0 < N ? min((uint32_t)WG, N - idx0) : 0u;
		st.pfx[blockIdx.x] = idx0; // rays are dense at bounce 0: reference position == index
	}
	if (idx >= N) return;
	const uint32_t seed = seeds[(size_t)(first_sample + s) * seed_stride];
	const uint32_t gx = idx % W, gy = idx / W;
	Rng rng = {gx + seed, gy + seed}; // camera.cl:38
	f2 s0 = rng_next(rng);
	float ox = s0.x < 0.5f ? pm_sqrt(2.0f * s0.x) - 0.5f : 1.5f - pm_sqrt(2.0f - 2.0f * s0.x);
	float oy = s0.y < 0.5f ? pm_sqrt(2.0f * s0.y) - 0.5f : 1.5f - pm_sqrt(2.0f - 2.0f * s0.y);
	float tx = ((float)gx + ox) * cam.texel.x;
	float ty = ((float)(gy + blockY) + oy) * cam.texel.y;
	// mix(mix(TL, BL, ty), mix(TR, BR, ty), tx), then normalize over all four lanes (w = 0)
	float lx = pm_mix(cam.tl.x, cam.bl.x, ty), ly = pm_mix(cam.tl.y, cam.bl.y, ty), lz = pm_mix(cam.tl.z, cam.bl.z, ty), lw = pm_mix(cam.tl.w, cam.bl.w, ty);
	float rx = pm_mix(cam.tr.x, cam.br.x, ty), ry = pm_mix(cam.tr.y, cam.br.y, ty), rz = pm_mix(cam.tr.z, cam.br.z, ty), rw = pm_mix(cam.tr.w, cam.br.w, ty);
	float dx = pm_mix(lx, rx, tx), dy = pm_mix(ly, ry, tx), dz = pm_mix(lz, rz, tx), dw = pm_mix(lw, rw, tx);
	float inv = 1.0f / pm_sqrt(dx * dx + dy * dy + dz * dz + dw * dw);
	if (write_origin) store_ray_o(st, slot, make_float4(cam.eye.x, cam.eye.y, cam.eye.z, kFltMax)); // (the wave-packet kernel takes the eye from its arguments)
	st.ray_d[slot] = make_float4(dx * inv, dy * inv, dz * inv, ibits((int)idx));
	// (the throughput of a camera ray is 1: the first shade step does not read it, so it is not written)
	if (zero_lsum) st.lsum[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------
// BVH traversal.  Results follow kernels/intersect.cl:184-347 (closest hit) and :26-180 (any
// hit): identical slab test, identical Moeller-Trumbore arithmetic, so the set of candidate
// hits and every t/u/v are bit-identical; the ORDER of traversal is ours (near child first,
// subtrees beyond the current best culled with a conservative margin), which cannot change the
// minimum -- exact ties are resolved by DFS rank = "first tested wins" of the reference.
// One lane = one ray; the node stack is a per-lane column of an LDS array ([depth][lane], so
// a wave's accesses are conflict free).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float slab_entry(float4 lo, float4 hi, f3 o, f3 inv, float maxDist) { // intersect.cl:301-309
	float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
	float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
	float minmax = pm_fmin(pm_fmin(pm_fmax(t0x, t1x), pm_fmax(t0y, t1y)), pm_fmax(t0z, t1z));
	float maxmin = pm_fmax(pm_fmax(pm_fmin(t0x, t1x), pm_fmin(t0y, t1y)), pm_fmin(t0z, t1z));
	return (minmax < 0 || maxmin > minmax) ? kFltMax : (maxmin >= maxDist ? kFltMax : maxmin);
}

struct HitRec { float t, u, v; int tri; int inst; uint32_t irank, trank; };

// A leaf reference is ~code (scene_layout.h).  code & 15 = triangle count (1..15) and code >> 4 = first triangle slot: the
// common case costs no fetch.  code & 15 == 0: a top-level leaf, code >> 4 = the mesh instance -- unless kBigLeafFlag is set:
// a leaf of more than 15 triangles, looked up by node id.  Returns (-first | -instance, count).
__device__ __forceinline__ int2 leaf_of(const BvhDev &B, int ref) {
	const uint32_t code = (uint32_t)~ref;
	const uint32_t cnt = code & 15u;
	if (cnt) return make_int2(-(int)(code >> 4), (int)cnt);
	if (!(code & kBigLeafFlag)) return make_int2(-(int)(code >> 4), 0);
	return B.leaves[(code & (kBigLeafFlag - 1u)) >> 4];
}

template <bool ANY_HIT>
__device__ __forceinline__ bool traverse(const BvhDev &B, f3 O, f3 D, float maxDist, int (*stk)[WG], HitRec &best) {
	const int lane = threadIdx.x;
	int sp = 0;
	int cur = B.root_ref;
	f3 o = O, d = D;
	f3 inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)}; // native_recip(ray.dir), intersect.cl:302
	int inst = 0;
	uint32_t irank = 0;
	best.t = maxDist; best.tri = -1; best.inst = 0; best.u = best.v = 0.0f; best.irank = best.trank = 0;
	for (;;) {
		if (cur >= 0) { // inner node: test both children
			const PairNode P = B.pairs[cur];
			float t0 = slab_entry(P.lo0, P.hi0, o, inv, maxDist);
			float t1 = slab_entry(P.lo1, P.hi1, o, inv, maxDist);
			if (!ANY_HIT) { // cull subtrees that start beyond the best hit (factor 1.001 >> rounding of t; +inf = box does not bound its subtree)
				if (t0 > best.t * P.hi0.w) t0 = kFltMax;
				if (t1 > best.t * P.hi1.w) t1 = kFltMax;
			}
			int c0 = fbits(P.lo0.w), c1 = fbits(P.lo1.w);
			const bool h0 = t0 < kFltMax, h1 = t1 < kFltMax;
			if (h0 && h1) {
				if (t1 < t0) { int t = c0; c0 = c1; c1 = t; }
				stk[sp++][lane] = c1;
				cur = c0;
				continue;
			}
			if (h0 || h1) { cur = h0 ? c0 : c1; continue; }
		} else { // leaf (cur = ~node)
			const int2 li = leaf_of(B, cur);
			if (li.y == 0) { // top-level leaf: enter the mesh instance (intersect.cl:239-252)
				inst = -li.x;
				const InstRec I = B.insts[inst];
				irank = (uint32_t)I.meta.y;
				stk[sp++][lane] = kExitMarker;
				// mul4x1 / mul3x1, util/transform.cl:9-26
				f3 no = {I.r0.x * o.x + I.r0.y * o.y + I.r0.z * o.z + I.r0.w, I.r1.x * o.x + I.r1.y * o.y + I.r1.z * o.z + I.r1.w,
				         I.r2.x * o.x + I.r2.y * o.y + I.r2.z * o.z + I.r2.w};
				f3 nd = {I.r0.x * d.x + I.r0.y * d.y + I.r0.z * d.z, I.r1.x * d.x + I.r1.y * d.y + I.r1.z * d.z,
				         I.r2.x * d.x + I.r2.y * d.y + I.r2.z * d.z};
				o = no; d = nd;
				inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
				cur = I.meta.x;
				continue;
			}
			const int first = -li.x;
			for (int t = first; t < first + li.y; t++) { // Moeller-Trumbore, intersect.cl:255-292
				const TriRec T = B.tris[t];
				f3 e1 = xyz(T.e1), e2 = xyz(T.e2);
				f3 pv = cross(d, e2);
				float det = dot(e1, pv);
				if (pm_fabs(det) < kEps) continue;
				float idet = rcp_det(det);
				f3 tv = o - xyz(T.v0);
				float u = dot(tv, pv) * idet;
				if (u < 0.0f || u > 1.0f) continue;
				f3 qv = cross(tv, e1);
				float v = dot(d, qv) * idet;
				if (v < 0.0f || u + v > 1.0f) continue;
				float tt = dot(e2, qv) * idet;
				if (ANY_HIT) {
					if (tt > kEps && tt < maxDist) return true;
				} else if (tt > kEps) {
					const uint32_t trank = (uint32_t)fbits(T.v0.w);
					const bool closer = tt < best.t;
					const bool tie = tt == best.t && best.tri >= 0 && (irank < best.irank || (irank == best.irank && trank < best.trank));
					if (closer || tie) { best.t = tt; best.u = u; best.v = v; best.tri = fbits(T.e1.w); best.inst = inst; best.irank = irank; best.trank = trank; }
				}
			}
		}
		// pop
		for (;;) {
			if (sp == 0) return best.tri >= 0;
			cur = stk[--sp][lane];
			if (cur != kExitMarker) break;
			if (sp == 0) return best.tri >= 0; // nothing pending above the instance: done, skip the restore
			o = O; d = D; // leaving the instance: restore the world-space ray (intersect.cl:330-335)
			inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
		}
	}
}

// rayIntersectionQuery: closest hit for the live rays of every workgroup.
__global__ __launch_bounds__(WG) void k_intersect(Streams st, BvhDev B) {
	__shared__ int stk[kTraversalStack][WG];
	if (threadIdx.x >= st.cnt_ray[blockIdx.x]) return;
	const size_t slot = (size_t)blockIdx.x * WG + threadIdx.x;
	const float4 o4 = load_ray_o(st, slot), d4 = st.ray_d[slot];
	HitRec h;
	traverse<false>(B, xyz(o4), xyz(d4), o4.w, stk, h);
	store_hit(st, slot, h.u, h.v, h.t, h.tri);
	if (st.hit_inst) st.hit_inst[slot] = h.inst;
}

// What an unoccluded shadow ray does (accumulateEmissiveSamples, pt_integrator.cl:278-296), in one of two ways:
//   acc != null  (exact mode: one sample per batch, the trace accumulator itself) -- the NEE radiance is added to the ray's
//                accumulator cell right here, in the reference's order; one path per cell and launch: plain read-modify-write;
//   acc == null  (batched mode) -- DEFERRED: every shadow ray writes ONE BYTE, vis[slot] = unoccluded (a dense array: the
//                stores of a chunk's rays fill whole sectors -- a 4-byte flag inside the 16-byte record, tried first, cost 19
//                bytes of write traffic per flag); the NEE records of every bounce are kept (one occ_e / vis array per
//                bounce) and k_fold_nee adds the unoccluded ones to the per-path radiance once per batch, bounce after
//                bounce -- the same sums in the same order.  Until round 4 every unoccluded ray did a read-modify-write of its
//                16-byte cell of lsum, scattered over the chunk's 4 KB: whole lines moved per bounce (1.55 x the kernel's
//                algorithmic bytes), and the any-hit kernel loaded the NEE radiance and the cell beside every ray.
__device__ __forceinline__ void nee_result(const Streams &st, float4 *acc, size_t slot, uint32_t cell, bool unoccluded) {
	if (acc) {
		if (unoccluded) {
			const float4 e = st.occ_e[slot];
			float4 a = acc[cell];
			a.x += e.x; a.y += e.y; a.z += e.z;
			acc[cell] = a;
		}
	} else {
		st.vis[slot] = unoccluded ? (uint8_t)1 : (uint8_t)0;
	}
}

// rayIntersectionTest + accumulateEmissiveSamples (intersect.cl:26-180, pt_integrator.cl:278-296)
// fused: an unoccluded shadow ray adds its NEE radiance straight to its accumulator cell (or records its visibility: nee_result).
__global__ __launch_bounds__(WG) void k_occlusion(Streams st, BvhDev B, float4 *acc, unsigned long long *stats) {
	__shared__ int stk[kTraversalStack][WG];
	if (threadIdx.x >= st.cnt_occ[blockIdx.x]) return;
	const size_t slot = (size_t)blockIdx.x * WG + threadIdx.x;
	const float4 o4 = st.occ_o[slot], d4 = st.occ_d[slot];
	HitRec h;
	const bool occluded = traverse<true>(B, xyz(o4), xyz(d4), o4.w, stk, h);
	nee_result(st, acc, slot, (uint32_t)fbits(d4.w), !occluded);
	const unsigned long long m = __ballot(!occluded);
	if (m != 0 && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&stats[ST_UNOCCLUDED], (unsigned long long)__popcll(m));
}

// ------------------------------------------------------------------------------------------
// Persistent traversal with lane refill ("k_trace").
//
// PMC counters of the one-ray-per-lane kernels above show the limiter: only ~28 % of the lanes
// of an issued VALU instruction are live (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU) --
// rays of a wave finish at very different times and lanes sit in different phases (box tests
// vs triangle tests).  This kernel attacks both:
//   * a workgroup is a persistent worker: the 256-slot chunks (= one workgroup's compacted rays) are
//     dealt statically (chunk c belongs to workgroup c % gridDim.x; its 4 waves share an LDS cursor)
//     and, whenever >= kRefillMin of a wave's lanes are idle, they are handed the next rays of the
//     chunk (ballot/popcount ranks) -- the wave's tail is filled with new work instead of waiting
//     for its longest ray.  Because the deal is static the grid must not exceed what the GPU holds at
//     once (the host sizes it from the occupancy API).  A single global ticket counter was measured
//     first: it saturates at ~88 dequeues/us (MI355X_MICROARCH.md, "dequeue"), a ~190 us floor under
//     every launch of a 16 Ki-chunk batch;
//   * "while-while" phases: all lanes first descend through inner nodes (lanes that already
//     reached a leaf wait), then all lanes with a leaf test triangles.
// Per-ray arithmetic is the same as traverse<> above (same slab test, same Moeller-Trumbore,
// same tie rule), so results are bit-identical; only the schedule changes.
//
// The body is written branch-light (round 2).  PMC on the headline frame: the first version issued
// 83 VALU + 39 SALU + 12 branch instructions per wave-iteration -- a third of it control flow the
// structurizer made out of nested `if`s (exec save / restore, branches over empty halves):
//   * the inner step is straight-line code under ONE exec mask: the far child is stored to the stack
//     slot unconditionally (the slot is free), the pop value is read unconditionally, and child order /
//     push / pop are selects;
//   * everything rare leaves the inner loop as a negative node code and is handled once per round in
//     the leaf phase: "ray finished" (kDone), "leaving the instance" (kExitMarker), instance leaves,
//     leaves of more than 15 triangles;
//   * Moeller-Trumbore runs without early exits (a lane's early exit saves the wave nothing): one
//     predicate accumulates the rejections, the hit record is updated with selects.
// -6 % (closest hit) / -13.5 % (any hit, which now also fits 7 waves per SIMD) kernel time.
// ------------------------------------------------------------------------------------------
// idle lanes of a wave before it fetches new rays (re-tuned once the kernel had become issue-bound: 24-32 x 16 is flat for
// closest hits, shadow rays gain 3 % over 40; 48 costs 6 %)
#ifndef POLARIS_REFILL_MIN
#define POLARIS_REFILL_MIN 32
#endif
constexpr int kRefillMin = POLARIS_REFILL_MIN;
#ifndef POLARIS_STRAGGLERS
#define POLARIS_STRAGGLERS 16
#endif
constexpr int kStragglers = POLARIS_STRAGGLERS;
#ifndef POLARIS_REFILL_MIN_ANY
#define POLARIS_REFILL_MIN_ANY POLARIS_REFILL_MIN
#endif
#ifndef POLARIS_STRAGGLERS_ANY
#define POLARIS_STRAGGLERS_ANY POLARIS_STRAGGLERS
#endif
constexpr int kRefillMinAny = POLARIS_REFILL_MIN_ANY, kStragglersAny = POLARIS_STRAGGLERS_ANY; // the same two for shadow rays

__device__ __forceinline__ float slab_entry_hw(float4 lo, float4 hi, f3 o, f3 inv, float maxDist) {
	// identical to slab_entry except that min/max are the hardware's IEEE minNum/maxNum
	// (v_min_f32/v_max3_f32): they differ from pm_fmin/pm_fmax only in the sign of a zero result,
	// which no comparison below can see.
	float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
	float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
	float minmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
	float maxmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
	return (minmax < 0 || maxmin > minmax) ? kFltMax : (maxmin >= maxDist ? kFltMax : maxmin);
}

// The same test as a predicate: slab_entry_hw(..) < kFltMax, i.e. "the ray enters the box before maxDist", with the entry
// distance in t -- the compares combined as booleans (scalar mask operations) instead of through a selected kFltMax.
__device__ __forceinline__ bool slab_hit_hw(float4 lo, float4 hi, f3 o, f3 inv, float maxDist, float &t) {
	float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
	float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
	float minmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
	float maxmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
	t = maxmin;
	return !(minmax < 0) && !(maxmin > minmax) && !(maxmin >= maxDist) && maxmin < kFltMax;
}

// The same predicate with fewer compares, for a caller that hands in boxDist = minNum(maxDist, FLT_MAX) (v_min_f32: a NaN or
// infinite maxDist becomes FLT_MAX):
//   !(maxmin >= maxDist) && maxmin < FLT_MAX  ==  maxmin < boxDist     for every maxmin and maxDist: a finite maxDist <= FLT_MAX
//       makes the second compare redundant for ordered maxmin and both sides are false for a NaN maxmin; for maxDist = +inf or
//       NaN the left side is "maxmin < FLT_MAX", which is the right side with boxDist = FLT_MAX;
// (Folding the other two compares into !(maxNum(maxmin, 0) > minmax) is equivalent as well, but costs an instruction more than it
// saves: the compiler quiets the max3's result with a v_max x, x in front of the maxNum.)
__device__ __forceinline__ bool slab_hit_fast(float4 lo, float4 hi, f3 o, f3 inv, float boxDist, float &t) {
	float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
	float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
	float minmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
	float maxmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
	t = maxmin;
	return !(minmax < 0) && !(maxmin > minmax) && maxmin < boxDist;
}

// The closest-hit kernel with the 16-entry stack is asked for 7 waves per SIMD (20 KB of LDS per workgroup: -2.4 % kernel
// time on the Cornell box, -9 % on the sphere scene); the any-hit variant fits 7 by itself since the rewrite.
// negative node codes that are not leaf references (a leaf code first << 4 | count never has all of bits 4..30 set)
constexpr int kDone = (int)0x80000001;  // the ray has nothing pending: write its result
constexpr int kIdle = (int)0x80000002;  // the lane holds no ray
constexpr int kFirstLeafRef = (int)0x80000010;

// Where the node records and the stack live (template parameter NODES of k_trace), chosen by the host from the scene's size:
//   kNodesGlobal  plain global_load (big trees: nearly every fetch misses an LDS copy and would pay the FLAT path)
//   kNodesLdsTop  the first kLdsTopNodes records (breadth-first = the hot top of the tree) staged in LDS per workgroup; the
//                 two-way fetch compiles to ONE flat_load through a selected pointer (small trees: -13 % kernel time)
//   kNodesLdsAll  "tiny scene" mode (<= kTinyPairs pair records, < 2047 triangle slots and nodes, stack <= 16): the WHOLE tree
//                 sits in LDS and is read with ds_read_b128 (no FLAT path, no L1 gathers for nodes).  What makes that fit at
//                 full occupancy: workgroups of 1024 threads share ONE copy of the tree (<= 32 KB) and the stack entries are
//                 16-bit node codes (2 KB per row for the 1024 lanes).  Since round 4 the block is laid out per scene -- the
//                 rows the tree needs, its pair records, and the triangle records that still fit half a CU's 160 KB
//                 (polaris_hip.hip plan_tiny_lds; the layout is in k_trace): two workgroups per CU = 8 waves per SIMD.  This
//                 mode exists for gfx950's LDS.  (Round 1 tried the whole tree in LDS with 256-thread workgroups: 48 KB each,
//                 3 per CU, +-0.)
enum NodeMode { kNodesGlobal = 0, kNodesLdsTop = 1, kNodesLdsAll = 2 };
constexpr int kTinyPairs = 512;
constexpr int kTinyBlock = 1024;
constexpr uint32_t kTinyMaxIndex = 2046; // triangle slots and leaf ids must fit the 11-bit field of a 16-bit stack entry (2047 | 15 is the exit marker)

// 16-bit stack entries of the tiny mode: every reference the tiny mode can push is the sign extension of its low 16 bits --
// inner nodes are indices <= 511, a leaf reference is ~(first << 4 | count) with first <= kTinyMaxIndex, i.e. >= -32752 -- so a
// push is a ds_write_b16 of the register and a pop a ds_read_i16, no conversion.  Only the instance exit marker needs a value of
// its own there: -32768 (= ~(2047 << 4 | 15), a leaf code kTinyMaxIndex rules out).
constexpr int kTinyExitMarker = -32768;

// Tiny mode keeps ONE word per triangle slot beside the 36 bytes of its vertices: rank << 19 | shading class << 11 | triangle
// (TriRec.e2.w, written at upload for scenes of <= kTinyMaxIndex triangle slots: scene_layout.h tiny_meta_word).  -1 = no hit.
__device__ __forceinline__ int tiny_meta_tri(int word) { return word < 0 ? -1 : (int)(((uint32_t)word & 0x7FFu) | ((uint32_t)word >> 11 & 0xFFu) << 24); }

// ONE = the scene IS one mesh instance (BvhDev::root_is_instance) and every box of its tree bounds its subtree -- the usual
// small scene.  The tiny mode has a variant compiled for that: no instance leaves, no exit markers, no instance rank to carry
// or compare, one cull limit per ray (best distance x 1.001, refreshed when a leaf changes it) instead of one product per child.
template <bool ANY_HIT, int STACK, int NODES, bool ONE = false>
__global__ __launch_bounds__(NODES == kNodesLdsAll ? kTinyBlock : WG)
__attribute__((amdgpu_waves_per_eu(NODES == kNodesLdsAll ? 8 : ((!ANY_HIT && STACK == 16) ? 7 : 1), NODES == kNodesLdsAll ? 8 : ((!ANY_HIT && STACK == 16) ? 7 : 10))))
void k_trace(Streams st, BvhDev B, uint32_t num_chunks, float4 *acc, unsigned long long *stats, uint32_t o_mask) {
	// o_mask: ~0, or 0 for the camera rays of a batch -- they all start at the eye with no distance limit, so k_generate does
	// not write 16 bytes of origin per path and this kernel does not read them: st.ray_o then points at ONE record (eye | FLT_MAX)
	// and every lane reads slot (its slot & 0).
	constexpr bool LDS_TOP = NODES == kNodesLdsTop, TINY = NODES == kNodesLdsAll;
	static_assert(!ONE || TINY, "the single-instance variant exists for the tiny mode only");
	constexpr int BLOCK = TINY ? kTinyBlock : WG; // threads per workgroup (a CHUNK of rays is always WG = 256 slots)
	typedef typename std::conditional<TINY, int16_t, int>::type StackEntry;
	constexpr int EXIT = TINY ? kTinyExitMarker : kExitMarker; // the instance exit marker as this variant's stack holds it
	// A ray that leaves an instance with more to do gets its world-space ray back (intersect.cl:330-335).  Where the register
	// budget allows (every variant but the tiny mode, which sits at 64 VGPRs for 8 waves) it is kept in six registers; the tiny
	// mode re-reads it from the ray streams.  In a scene of many instances a ray enters and leaves several: two dependent
	// gathers per exit (and, until round 4, one more per entry for the instance id) were a third of the lines such a ray touched.
	constexpr bool KEEP_WORLD = !TINY;
	// The per-lane node stack: a column of an LDS array.  A lane keeps its stack pointer as the BYTE OFFSET of its top entry
	// (sp0 = empty, + kRow per entry), so a pop reads at `sp` and a push writes at `sp + kRow` with no address arithmetic;
	// row 0 is a dummy, so that "the entry below an empty stack" can be read (and ignored) without a clamp.
	// Tiny mode: ONE dynamic LDS block laid out by the host for the uploaded scene (polaris_hip.hip, plan_tiny_lds):
	//   [num_pairs pair records, 64 B][lds_tris triangle records, 36 B: v0, e1, e2][lds_tris packed words, 4 B][stack rows]
	// i.e. exactly the stack rows the tree needs, its pair records, and as many triangle records as the rest of half a CU's LDS
	// holds (slots are in descending order of how often a ray reaches their leaf, scene_layout.h: what does not fit -- the tail of
	// a small sphere's facets -- stays in global memory).  The dummy row below the stack is whatever precedes the stack (it is
	// read and ignored, never written).  Two such workgroups share a CU's 160 KB (gfx950) = 8 waves per SIMD.
	extern __shared__ __attribute__((aligned(16))) char tiny_lds[];
	__shared__ StackEntry stk[TINY ? 1 : STACK + 1][TINY ? 1 : BLOCK];
	constexpr uint32_t kRow = BLOCK * sizeof(StackEntry);
	__shared__ uint32_t wg_cursor;
	if (threadIdx.x == 0) wg_cursor = 0;
	__shared__ float4 top_static[LDS_TOP ? kLdsTopNodes * 4 : 1];
	float4 *const top = TINY ? reinterpret_cast<float4 *>(tiny_lds) : top_static;
	float *const ltri = TINY ? reinterpret_cast<float *>(tiny_lds + (size_t)B.num_pairs * sizeof(PairNode)) : nullptr; // 9 floats per slot, slots [0, B.lds_tris)
	uint32_t *const lmeta = TINY ? reinterpret_cast<uint32_t *>(ltri + 9 * (size_t)B.lds_tris) : nullptr;            // tiny_meta word per slot
	if (LDS_TOP || TINY) {
		const uint32_t n4 = (TINY ? B.num_pairs : min((uint32_t)kLdsTopNodes, B.num_pairs)) * 4;
		const float4 *src = reinterpret_cast<const float4 *>(B.pairs);
		for (uint32_t i = threadIdx.x; i < n4; i += BLOCK) top[i] = src[i];
		if (TINY) {
			for (uint32_t i = threadIdx.x; i < B.lds_tris; i += BLOCK) { // (one thread per slot: three 16-byte loads, ten LDS words)
				const TriRec T = B.tris[i];
				float *t9 = ltri + 9 * i;
				t9[0] = T.v0.x; t9[1] = T.v0.y; t9[2] = T.v0.z; t9[3] = T.e1.x; t9[4] = T.e1.y; t9[5] = T.e1.z; t9[6] = T.e2.x; t9[7] = T.e2.y; t9[8] = T.e2.z;
				lmeta[i] = (uint32_t)fbits(T.e2.w);
			}
		}
	}
	__syncthreads();
	// (tiny mode: a lane's stack pointer is its offset in the dynamic block itself -- the block's address is a link-time constant,
	// so it folds into the instruction's offset field and no access pays an add for the stack's run-time position)
	char *const stk_bytes = TINY ? tiny_lds : reinterpret_cast<char *>(&stk[0][0]);
	const int tid = threadIdx.x;
	const uint32_t sp0 = (TINY ? B.tiny_stack_off - kRow : 0u) + (uint32_t)tid * (uint32_t)sizeof(StackEntry);
	auto push_ref = [&](uint32_t at, int ref) { *reinterpret_cast<StackEntry *>(stk_bytes + at + kRow) = (StackEntry)ref; }; // onto a stack whose pointer is `at`
	auto read_ref = [&](uint32_t at) -> int { return (int)*reinterpret_cast<const StackEntry *>(stk_bytes + at); };         // the top entry of such a stack
	const uint32_t lane = tid & 63;
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t *cnts = ANY_HIT ? st.cnt_occ : st.cnt_ray;
	const float4 *src_o = ANY_HIT ? st.occ_o : st.ray_o;
	const float4 *src_d = ANY_HIT ? st.occ_d : st.ray_d;

	uint32_t chunk = 0, off = 0, cnt = 0; // wave-uniform queue state
	bool drained = false;
	uint32_t slot = 0; // per-lane ray state
	f3 o = {0, 0, 0}, d = {0, 0, 0}, inv = {0, 0, 0};
	float maxDist = 0.0f;
	float boxDist = 0.0f, best_cull = 0.0f; // ONE: minNum(maxDist, FLT_MAX) for the box tests (slab_hit_fast); best_t x kCullMargin
	uint32_t sp = sp0;
	int cur = kIdle, cell = 0;
	uint32_t irank = 0, unocc = 0;
	float best_t = 0.0f, best_u = 0.0f, best_v = 0.0f;
	int best_tri = -1;
	uint32_t best_irank = 0, best_trank = 0;
	f3 wo = {0, 0, 0}, wd = {0, 0, 0}; // KEEP_WORLD: the world-space ray
	f3 nee = {0, 0, 0}, acc_old = {0, 0, 0}; // any hit: the NEE radiance of the shadow ray and its accumulator cell, both fetched at set-up

	// next pending node of the lane's ray: the popped reference, or kDone when nothing is pending -- an instance's exit
	// marker with nothing above it ends the ray too (no need to restore the world-space ray first)
	auto pop = [&]() {
		const int popped = read_ref(sp);
		const bool empty = sp == sp0;
		const uint32_t spm = sp - kRow;
		cur = (empty || (!ONE && popped == EXIT && spm == sp0)) ? kDone : popped;
		sp = empty ? sp : spm;
	};

	// a lane takes up the ray in `ray_slot` (its origin | max distance and direction | path word are in o4, d4)
	auto start_ray = [&](uint32_t ray_slot, float4 o4, float4 d4) {
		slot = ray_slot;
		o = xyz(o4); d = xyz(d4);
		if (KEEP_WORLD) { wo = o; wd = d; }
		maxDist = o4.w;
		cell = fbits(d4.w);
		if (ANY_HIT) {
			// what an unoccluded ray adds, and the cell it adds to (one path per cell and launch: nobody else touches it):
			// fetched here, beside the ray, so that finishing a ray is a store and not two dependent round trips
			// (round 3 A/B: reading the cell only when a ray ends unoccluded moves no fewer bytes -- 2.44 vs 2.40 GB of FETCH_SIZE
			// per frame: on this scene 90 % of the shadow rays do reach the light, and the 16-byte cells move as whole lines
			// either way -- and costs 3 % of the kernel's time)
			if (acc) { // (exact mode only; batched mode writes the ray's visibility byte at the end instead: nee_result)
				const float4 e4 = st.occ_e[slot], a4 = acc[cell];
				nee = xyz(e4); acc_old = xyz(a4);
			}
		}
		sp = sp0;
		cur = B.root_ref;
		irank = 0;
		if (ONE || B.root_is_instance) { // enter the scene's one instance right away (intersect.cl:239-252; mul4x1 / mul3x1, util/transform.cl:9-26)
			const InstRec &I = B.root_inst;
			const f3 no = {I.r0.x * o.x + I.r0.y * o.y + I.r0.z * o.z + I.r0.w, I.r1.x * o.x + I.r1.y * o.y + I.r1.z * o.z + I.r1.w,
			               I.r2.x * o.x + I.r2.y * o.y + I.r2.z * o.z + I.r2.w};
			const f3 nd = {I.r0.x * d.x + I.r0.y * d.y + I.r0.z * d.z, I.r1.x * d.x + I.r1.y * d.y + I.r1.z * d.z,
			               I.r2.x * d.x + I.r2.y * d.y + I.r2.z * d.z};
			o = no; d = nd;
			irank = (uint32_t)I.meta.y;
			cur = I.meta.x; // (no exit marker: nothing is ever pending below this instance, so an empty stack ends the ray -- and the
			                //  stack needs one row less: polaris_hip.hip plan_tiny_lds)
		}
		rcp_dir(d.x, d.y, d.z, inv.x, inv.y, inv.z); // native_recip(ray.dir), intersect.cl:302
		best_t = maxDist; best_tri = -1; best_u = best_v = 0.0f; best_irank = best_trank = 0;
		if (ONE) { boxDist = __builtin_fminf(maxDist, kFltMax); best_cull = best_t * kCullMargin; }
	};
	// the next rays of the workgroup's chunks go to the lanes for which wants() holds (take(slot) must make it false)
	auto draw = [&](auto wants, auto take) {
		for (;;) {
			if (off >= cnt) {
				uint32_t c = 0;
				if (lane == 0) c = atomicAdd(&wg_cursor, 1u);
				c = blockIdx.x + __builtin_amdgcn_readfirstlane(c) * gridDim.x;
				if (c >= num_chunks) { drained = true; break; }
				chunk = c;
				off = 0;
				cnt = cnts[chunk];
				continue;
			}
			const unsigned long long m = __ballot(wants());
			const uint32_t n = __popcll(m);
			if (n == 0) break;
			const uint32_t share = min(cnt - off, n);
			const uint32_t rank = __popcll(m & below);
			if (wants() && rank < share) take(chunk * WG + off + rank);
			off += share;
		}
	};
	for (;;) {
		// ---- refill idle lanes from the workgroup's chunks -------------------------------------------
		// (Round 3 tried the asynchronous version once more, now that the kernel has registers to spare: every lane keeps its
		// NEXT ray staged in eight registers, loaded in front of the inner phase -- which issues no vector-memory instruction
		// in the tiny mode -- and an idle lane restarts from them without touching memory.  60 / 63 VGPRs, no spill, bit-exact,
		// and closest hit 4.4 against 4.0 ms, any hit 2.7 against 2.4: vmcnt is in order, so the first triangle fetch after
		// a top-up waits for the top-up as well, and the wave is parked there instead of here.  Round 4 tried it a fourth time
		// with the triangle records in LDS -- no vector-memory wait left in either loop, checked in the ISA -- and it is still
		// 5 % slower per kernel: with 8 waves per SIMD the parked wave costs nothing, the kernel is bound by instruction issue,
		// and the staging adds instructions.  EXPERIMENTS.md.)
		{
			const unsigned long long freem = __ballot(cur == kIdle);
			if (!drained && (freem == ~0ull || __popcll(freem) >= (ANY_HIT ? kRefillMinAny : kRefillMin))) {
				draw([&]() { return cur == kIdle; }, [&](uint32_t ray_slot) { start_ray(ray_slot, ANY_HIT ? src_o[ray_slot] : load_ray_o(st, ray_slot & o_mask), src_d[ray_slot]); });
			}
			if (__ballot(cur != kIdle) == 0ull) {
				if (drained) break;
				continue;
			}
		}
		// ---- phase 1: descend through inner nodes (intersect.cl:296-328) ---------------------------------
		// left early once fewer than kStragglers lanes are still descending (they continue next round)
		// (one step always if anybody descends, further steps while at least kStragglers lanes still do)
		if (__ballot(cur >= 0) != 0ull) do {
			if (cur >= 0) {
				PairNode P;
				if (TINY || (LDS_TOP && cur < kLdsTopNodes)) { P.lo0 = top[4 * cur]; P.hi0 = top[4 * cur + 1]; P.lo1 = top[4 * cur + 2]; P.hi1 = top[4 * cur + 3]; }
				else P = B.pairs[cur];
				// what a pop would deliver: read NOW, beside the node record (the push below writes the entry above it), so the
				// stack's LDS round trip is off the step's dependent chain
				const int popped = read_ref(sp);
				const bool empty = sp == sp0;
				const uint32_t spm = sp - kRow;
				float t0, t1;
				const bool e0 = ONE ? slab_hit_fast(P.lo0, P.hi0, o, inv, boxDist, t0) : slab_hit_hw(P.lo0, P.hi0, o, inv, maxDist, t0);
				const bool e1 = ONE ? slab_hit_fast(P.lo1, P.hi1, o, inv, boxDist, t1) : slab_hit_hw(P.lo1, P.hi1, o, inv, maxDist, t1);
				// closest hit: cull subtrees that start beyond the best hit (+inf factor = box does not bound its subtree; ONE: every
				// factor is kCullMargin, and best_cull is that product)
				const bool h0 = e0 && (ANY_HIT || !(t0 > (ONE ? best_cull : best_t * P.hi0.w)));
				const bool h1 = e1 && (ANY_HIT || !(t1 > (ONE ? best_cull : best_t * P.hi1.w)));
				// nearer child first for closest hits (what makes the cull bite); stored order for shadow rays
				const bool second_first = h1 && (!h0 || (!ANY_HIT && t1 < t0)); // (shadow rays: nearer-first / farther-first measured, profiles/r05_any_hit_order_ab.txt: within noise except the terrain, -5 %)
				const int c0 = fbits(P.lo0.w), c1 = fbits(P.lo1.w);
				const int nearc = second_first ? c1 : c0, farc = second_first ? c0 : c1;
				push_ref(sp, farc); // kept only when both children are hit (sp advances); otherwise the slot stays free
				const bool both = h0 && h1, none = !(h0 || h1);
				const int after_pop = (empty || (!ONE && popped == EXIT && spm == sp0)) ? kDone : popped;
				cur = none ? after_pop : nearc;
				sp = both ? sp + kRow : ((none && !empty) ? spm : sp);
			}
		} while (__popcll(__ballot(cur >= 0)) >= (ANY_HIT ? kStragglersAny : kStragglers));
		// ---- phase 2: everything that is not an inner node ------------------------------------------------
		if (cur == kDone) { // the ray is finished
			if (ANY_HIT) { // unoccluded: accumulateEmissiveSamples, pt_integrator.cl:278-296 (both operands were fetched at set-up)
				if (acc) {
					float *c = reinterpret_cast<float *>(acc + cell);
					c[0] = acc_old.x + nee.x; c[1] = acc_old.y + nee.y; c[2] = acc_old.z + nee.z;
				} else {
					st.vis[slot] = 1; // deferred: k_fold_nee adds the ray's NEE record
				}
				unocc++;
			} else {
				store_hit(st, slot, best_u, best_v, best_t, TINY ? tiny_meta_tri(best_tri) : best_tri);
			}
			cur = kIdle;
		}
		if (!ONE && cur == EXIT) { // leaving the instance: back to the world-space ray (intersect.cl:330-335)
			if (KEEP_WORLD) { o = wo; d = wd; }
			else {
				const float4 o4 = ANY_HIT ? src_o[slot] : load_ray_o(st, slot & o_mask), d4 = src_d[slot];
				o = xyz(o4); d = xyz(d4);
			}
			rcp_dir(d.x, d.y, d.z, inv.x, inv.y, inv.z);
			pop();
		}
		if (!ONE && cur < 0 && cur >= kFirstLeafRef && (((uint32_t)~cur) & 15u) == 0u) { // a top-level leaf, or a leaf of more than 15 triangles
			const uint32_t code = (uint32_t)~cur;
			if (!(code & kBigLeafFlag)) { // top-level leaf: enter the mesh instance (intersect.cl:239-252); its id is in the reference
				const InstRec I = B.insts[code >> 4];
				irank = (uint32_t)I.meta.y;
				push_ref(sp, EXIT);
				sp += kRow;
				// mul4x1 / mul3x1, util/transform.cl:9-26
				const f3 no = {I.r0.x * o.x + I.r0.y * o.y + I.r0.z * o.z + I.r0.w, I.r1.x * o.x + I.r1.y * o.y + I.r1.z * o.z + I.r1.w,
				               I.r2.x * o.x + I.r2.y * o.y + I.r2.z * o.z + I.r2.w};
				const f3 nd = {I.r0.x * d.x + I.r0.y * d.y + I.r0.z * d.z, I.r1.x * d.x + I.r1.y * d.y + I.r1.z * d.z,
				               I.r2.x * d.x + I.r2.y * d.y + I.r2.z * d.z};
				o = no; d = nd;
				rcp_dir(d.x, d.y, d.z, inv.x, inv.y, inv.z);
				cur = I.meta.x;
			} else { // more than 15 triangles: re-filed as a run of inline leaves is not possible (count > 15): walk it here
				const int2 li = B.leaves[(code & (kBigLeafFlag - 1u)) >> 4];
				bool occluded = false;
				for (int t = -li.x; t < -li.x + li.y && !occluded; t++) {
					const TriRec T = B.tris[t];
					const f3 e1 = xyz(T.e1), e2 = xyz(T.e2);
					const f3 pv = cross(d, e2);
					const float det = dot(e1, pv);
					if (pm_fabs(det) < kEps) continue;
					const float idet = rcp_det(det);
					const f3 tv = o - xyz(T.v0);
					const float u = dot(tv, pv) * idet;
					if (u < 0.0f || u > 1.0f) continue;
					const f3 qv = cross(tv, e1);
					const float v = dot(d, qv) * idet;
					if (v < 0.0f || u + v > 1.0f) continue;
					const float tt = dot(e2, qv) * idet;
					if (ANY_HIT) {
						if (tt > kEps && tt < maxDist) occluded = true;
					} else if (tt > kEps) {
						const uint32_t trank = (uint32_t)fbits(T.v0.w);
						const bool closer = tt < best_t;
						const bool tie = tt == best_t && best_tri >= 0 && (irank < best_irank || (irank == best_irank && trank < best_trank));
						if (closer || tie) { best_t = tt; best_u = u; best_v = v; best_tri = fbits(T.e1.w); best_irank = irank; best_trank = trank; }
					}
				}
				if (occluded) { cur = kIdle; if (!acc) st.vis[slot] = 0; } // shadow ray blocked: nothing to add
				else pop();
			}
		}
		// ---- inline leaves (1..15 triangles): Moeller-Trumbore, intersect.cl:255-292, without early exits ----
		{
			const bool tl = cur < 0 && cur >= kFirstLeafRef && (ONE || (((uint32_t)~cur) & 15u) != 0u);
			if (__ballot(tl) != 0ull) {
				const uint32_t code = (uint32_t)~cur;
				const uint32_t first = code >> 4, ntri = tl ? (code & 15u) : 0u;
				const int popped = read_ref(sp); // what follows the leaf: read beside the triangles, not after them
				bool occluded = false;
				uint32_t i = 0; // (bottom-tested: some lane holds a leaf, and a leaf has at least one triangle)
				// one triangle through Moeller-Trumbore (intersect.cl:255-292) without early exits; updates the lane's hit record
				// (tiny mode: `word` is the slot's tiny_meta word -- rank | class | triangle -- and best_tri holds the best hit's word:
				// ranks are unique within a mesh and sit in the top bits, so comparing the words compares the ranks)
				auto test_tri = [&](f3 v0, f3 e1, f3 e2, uint32_t trank, int word) {
					const f3 pv = cross(d, e2);
					const float det = dot(e1, pv);
					bool ok = !(pm_fabs(det) < kEps);
					const float idet = rcp_det(det);
					const f3 tv = o - v0;
					const float u = dot(tv, pv) * idet;
					ok = ok && !(u < 0.0f || u > 1.0f);
					const f3 qv = cross(tv, e1);
					const float v = dot(d, qv) * idet;
					ok = ok && !(v < 0.0f || u + v > 1.0f);
					const float tt = dot(e2, qv) * idet;
					ok = ok && tt > kEps;
					if (ANY_HIT) {
						occluded = occluded || (ok && tt < maxDist);
					} else {
						const bool closer = tt < best_t;
						const bool lower = TINY ? (uint32_t)word < (uint32_t)best_tri : trank < best_trank;
						const bool tie = tt == best_t && best_tri >= 0 && (ONE ? lower : (irank < best_irank || (irank == best_irank && lower)));
						const bool take = ok && (closer || tie);
						best_t = take ? tt : best_t; best_u = take ? u : best_u; best_v = take ? v : best_v;
						best_tri = take ? word : best_tri;
						if (!ONE) best_irank = take ? irank : best_irank;
						if (!TINY) best_trank = take ? trank : best_trank;
					}
				};
				// (round 4 A/B: two triangles per round -- six loads in flight, half the rounds -- where the tree is read from global
				// memory, and 64-byte triangle records that never straddle a line: both within +-2 % on the terrain, C4 and C5,
				// profiles/r04_tri_variants_ab.txt)
				do {
					if (i < ntri && !occluded) {
						const uint32_t s = first + i;
						if (TINY && s < B.lds_tris) { // (two explicit paths: one fetch through a selected pointer would be a FLAT load)
							const float *t9 = ltri + 9 * s;
							test_tri(f3{t9[0], t9[1], t9[2]}, f3{t9[3], t9[4], t9[5]}, f3{t9[6], t9[7], t9[8]}, 0u, (int)lmeta[s]);
						} else {
							const TriRec T = B.tris[s];
							test_tri(xyz(T.v0), xyz(T.e1), xyz(T.e2), (uint32_t)fbits(T.v0.w), fbits(TINY ? T.e2.w : T.e1.w));
						}
					}
					i++;
				} while (__ballot(i < ntri && !occluded) != 0ull);
				if (ONE && !ANY_HIT) best_cull = best_t * kCullMargin;
				if (tl) {
					if (ANY_HIT && occluded) { cur = kIdle; if (!acc) st.vis[slot] = 0; } // blocked: nothing to add
					else {
						const bool empty = sp == sp0;
						const uint32_t spm = sp - kRow;
						cur = (empty || (!ONE && popped == EXIT && spm == sp0)) ? kDone : popped;
						sp = empty ? sp : spm;
					}
				}
			}
		}
	}
	if (ANY_HIT) {
		// wave sum -> workgroup sum in LDS -> ONE global atomic per workgroup (contended atomics on a single address run
		// at ~80/us; see k_shade)
		uint32_t v = unocc;
#pragma unroll
		for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
		__syncthreads();
		if (threadIdx.x == 0) wg_cursor = 0;
		__syncthreads();
		if (lane == 0 && v) atomicAdd(&wg_cursor, v);
		__syncthreads();
		if (threadIdx.x == 0 && wg_cursor) atomicAdd(&stats[ST_UNOCCLUDED], (unsigned long long)wg_cursor);
	}
}


// ------------------------------------------------------------------------------------------
// Wave-packet traversal for PRIMARY rays (the role of rayPacketIntersectionQuery,
// kernels/intersect.cl:353-575, without its two defects -- SURVEY.md 8a-3).
//
// One wave64 = one packet of 64 consecutive pixels walking the tree together: the node index is
// wave-uniform, so the 64 B pair record, the leaf header and the triangles are fetched ONCE per
// packet through the scalar cache instead of 64 per-lane gathers; the stack (node, 64-bit active
// mask) is wave-shared in LDS; child order and "does anybody want this child" are decided by
// __ballot votes.  Every lane keeps its own active bit: a lane takes part in a subtree only if ITS
// slab test passed at every ancestor, which is exactly the set of leaves its own traversal (and the
// reference's rayIntersectionQuery) visits -- so per-ray results are bit-identical to k_trace.
// ------------------------------------------------------------------------------------------
// CAMERA: every ray starts at the eye with no distance limit (camera rays of a batch): the origin stream is neither written
// by k_generate nor read here.
template <bool ANY_HIT, bool CAMERA = false>
__global__ __launch_bounds__(WG) void k_trace_packet(Streams st, BvhDev B, float4 *acc, unsigned long long *stats, float3 eye) {
	__shared__ int p_ref[4][kTraversalStack];
	__shared__ unsigned long long p_mask[4][kTraversalStack];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t cnt = (ANY_HIT ? st.cnt_occ : st.cnt_ray)[blockIdx.x];
	const float4 *src_o = ANY_HIT ? st.occ_o : st.ray_o;
	const float4 *src_d = ANY_HIT ? st.occ_d : st.ray_d;
	if (wave * 64 >= cnt) return; // nothing live in this wave's 64 slots (uniform per wave)
	const bool valid = tid < cnt;
	const size_t slot = (size_t)blockIdx.x * WG + tid;
	const float4 o4 = CAMERA ? make_float4(eye.x, eye.y, eye.z, kFltMax) : (valid ? (ANY_HIT ? src_o[slot] : load_ray_o(st, slot)) : make_float4(0, 0, 0, 0));
	const float4 d4 = valid ? src_d[slot] : make_float4(1, 1, 1, 0);
	bool occluded = false; // ANY_HIT: a lane that found its blocker leaves the packet for good
	const f3 O = xyz(o4), D = xyz(d4);
	const float maxDist = o4.w;
	f3 o = O, d = D;
	const f3 INV = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)}; // native_recip(ray.dir), intersect.cl:302 (kept for the way out of an instance)
	f3 inv = INV;
	HitRec best;
	best.t = maxDist; best.tri = -1; best.inst = 0; best.u = best.v = 0.0f; best.irank = best.trank = 0;
	int inst = 0;
	uint32_t irank = 0;
	int sp = 0;
	int cur = B.root_ref;
	bool active = valid;
	int *refs = p_ref[wave];
	unsigned long long *masks = p_mask[wave];
	for (;;) {
		bool need_pop = false;
		if (cur >= 0) {
			const int ucur = __builtin_amdgcn_readfirstlane(cur);
			const PairNode P = B.pairs[ucur]; // uniform address: scalar loads
			float t0 = slab_entry_hw(P.lo0, P.hi0, o, inv, maxDist);
			float t1 = slab_entry_hw(P.lo1, P.hi1, o, inv, maxDist);
			const float lim0 = ANY_HIT ? kFltMax : best.t * P.hi0.w, lim1 = ANY_HIT ? kFltMax : best.t * P.hi1.w;
			const bool w0 = active && t0 < kFltMax && !(t0 > lim0), w1 = active && t1 < kFltMax && !(t1 > lim1);
			const unsigned long long m0 = __ballot(w0), m1 = __ballot(w1);
			const int c0 = fbits(P.lo0.w), c1 = fbits(P.lo1.w);
			if (m0 != 0ull && m1 != 0ull) {
				// packet-coherence vote: visit first the child that more lanes reach first
				const int near0 = __popcll(__ballot(w0 && (!w1 || t0 <= t1))), near1 = __popcll(__ballot(w1 && (!w0 || t1 < t0)));
				const bool first0 = near0 >= near1;
				if (lane == 0) { refs[sp] = first0 ? c1 : c0; masks[sp] = first0 ? m1 : m0; }
				__builtin_amdgcn_wave_barrier();
				sp++;
				cur = first0 ? c0 : c1;
				active = first0 ? w0 : w1;
			} else if (m0 != 0ull) { cur = c0; active = w0; }
			else if (m1 != 0ull) { cur = c1; active = w1; }
			else need_pop = true;
		} else {
			const int unode = __builtin_amdgcn_readfirstlane(~cur);
			const int2 li = leaf_of(B, ~unode);
			if (li.y == 0) { // enter the instance (all lanes transform; only active ones matter)
				inst = -li.x;
				const InstRec I = B.insts[inst];
				irank = (uint32_t)I.meta.y;
				const unsigned long long am = __ballot(active);
				if (lane == 0) { refs[sp] = kExitMarker; masks[sp] = am; }
				__builtin_amdgcn_wave_barrier();
				sp++;
				f3 no = {I.r0.x * o.x + I.r0.y * o.y + I.r0.z * o.z + I.r0.w, I.r1.x * o.x + I.r1.y * o.y + I.r1.z * o.z + I.r1.w,
				         I.r2.x * o.x + I.r2.y * o.y + I.r2.z * o.z + I.r2.w};
				f3 nd = {I.r0.x * d.x + I.r0.y * d.y + I.r0.z * d.z, I.r1.x * d.x + I.r1.y * d.y + I.r1.z * d.z,
				         I.r2.x * d.x + I.r2.y * d.y + I.r2.z * d.z};
				o = no; d = nd;
				inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
				cur = I.meta.x;
			} else {
				const int first = -li.x;
				int t = first; // (bottom-tested: a triangle leaf holds at least one triangle)
				do {
					const TriRec T = B.tris[t]; // uniform address: scalar loads
					// Moeller-Trumbore without early exits (as in k_trace): a packet only skips what ALL its lanes skip, which is rare,
					// and with nested exits the compiler sinks each load of the record into the branch that first needs it -- three
					// dependent scalar round trips per triangle instead of one
					if (active) {
						const f3 e1 = xyz(T.e1), e2 = xyz(T.e2);
						const f3 pv = cross(d, e2);
						const float det = dot(e1, pv);
						bool ok = !(pm_fabs(det) < kEps);
						const float idet = rcp_det(det);
						const f3 tv = o - xyz(T.v0);
						const float u = dot(tv, pv) * idet;
						ok = ok && !(u < 0.0f || u > 1.0f);
						const f3 qv = cross(tv, e1);
						const float v = dot(d, qv) * idet;
						ok = ok && !(v < 0.0f || u + v > 1.0f);
						const float tt = dot(e2, qv) * idet;
						ok = ok && tt > kEps;
						if (ANY_HIT) {
							if (ok && tt < maxDist) { occluded = true; active = false; }
						} else {
							const uint32_t trank = (uint32_t)fbits(T.v0.w);
							const bool closer = tt < best.t;
							const bool tie = tt == best.t && best.tri >= 0 && (irank < best.irank || (irank == best.irank && trank < best.trank));
							const bool take = ok && (closer || tie);
							best.t = take ? tt : best.t; best.u = take ? u : best.u; best.v = take ? v : best.v;
							best.tri = take ? fbits(T.e1.w) : best.tri; best.inst = take ? inst : best.inst;
							best.irank = take ? irank : best.irank; best.trank = take ? trank : best.trank;
						}
					}
				} while (++t < first + li.y);
				need_pop = true;
			}
		}
		if (need_pop) {
			bool done = false, leave = false;
			for (;;) {
				if (sp == 0) { done = true; break; }
				sp--;
				__builtin_amdgcn_wave_barrier();
				cur = refs[sp];
				const unsigned long long m = masks[sp];
				active = ((m >> lane) & 1ull) != 0ull && !occluded;
				if (cur != kExitMarker) {
					if (__ballot(active) == 0ull) continue; // nobody left for this subtree
					break;
				}
				leave = true; // leaving the instance (the ray is restored below, outside this loop: nothing in it reads the ray)
			}
			if (done) break;
			if (leave) { o = O; d = D; inv = INV; }
		}
	}
	if (ANY_HIT) {
		const bool clear = valid && !occluded;
		if (valid) nee_result(st, acc, slot, (uint32_t)fbits(d4.w), clear);
		const unsigned long long m = __ballot(clear);
		if (lane == 0 && m) atomicAdd(&stats[ST_UNOCCLUDED], (unsigned long long)__popcll(m));
	} else if (valid) {
		store_hit(st, slot, best.u, best.v, best.t, best.tri);
		if (st.hit_inst) st.hit_inst[slot] = best.inst;
	}
}

// ------------------------------------------------------------------------------------------
// shadeHits (+ shadePrimaryRayMisses / shadeIndirectRayMisses), kernels/pt_integrator.cl:17-275
// ------------------------------------------------------------------------------------------
struct ShadeArgs {
	const uint32_t *seeds; uint32_t seed_stride, first_sample; // shade seed of sample s: seeds[(first+s)*stride + 1 + bounce]
	uint32_t N, Npad, W, blockY;
	uint32_t bounce, min_rr;
	int last_bounce;   // no closest-hit query follows (pipeline.go:203): do not emit indirect rays
	int exact;         // accumulate into acc[pixelIndex] (trace accumulator) instead of lsum[path slot]
	float4 *acc;       // trace accumulator (exact) or lsum (batched)
	const uint32_t *emask_in; // emit masks of the previous shade step (null at bounce 0: canonical index = slot index)
	uint32_t *emask_out;      // ... of this one
};

struct ShadeOut {
	bool emit_ind;
	float4 ro, rd, thr;             // new indirect ray (origin|maxDist, dir|path word, throughput)
	uint32_t hit, miss, emit;       // event flags for the counters
};

// A path's TERMINAL contribution -- background radiance of a missed ray (pt_integrator.cl:214-275) or the radiance of a directly
// hit emitter (:101-107): a path gets at most one, at its end.  Exact mode adds it to the trace accumulator in the reference's
// order.  Batched mode: the per-path cell of lsum holds nothing else until k_fold_nee (the NEE terms are deferred), so it is a
// plain 16-byte store, not a read-modify-write.
__device__ __forceinline__ void terminal_add(const ShadeArgs &A, uint32_t cell, f3 add) {
	if (A.exact) {
		float4 a = A.acc[cell];
		a.x += add.x; a.y += add.y; a.z += add.z;
		A.acc[cell] = a;
	} else {
		A.acc[cell] = make_float4(add.x, add.y, add.z, 0.0f);
	}
}

// One ray through shadeHits / shade*RayMisses.  `gid_ref` is the ray's position in the reference's
// compacted buffer (PRNG state, pt_integrator.cl:81), `sample` the sample the workgroup belongs to.
// The shadow ray (origin|maxDist, dir|accumulator cell, NEE radiance) is handed to `occ_sink` the moment it exists -- the
// kernels store it right there (shadow rays have no order to keep), so its 11 registers are free for the rest of the ray.
template <bool LDS, class OccSink>
__device__ __forceinline__ void shade_ray(const SceneT<LDS> &S, const ShadeArgs &A, uint32_t sample, uint32_t seed, uint32_t gid_ref,
                                          float4 d4, float4 t4, float4 h4, ShadeOut &R, OccSink occ_sink) {
	R.emit_ind = false;
	R.hit = R.miss = R.emit = 0;
	const uint32_t pword = (uint32_t)fbits(d4.w);
	const uint32_t path_index = pword & 0xFFFFFFu;
	uint32_t flags = pword >> 24;
	const uint32_t pixel_index = A.blockY * A.W + path_index; // camera.cl:32-33
	const uint32_t cell = A.exact ? pixel_index : (uint32_t)(sample * A.Npad + path_index);
	f3 thr = xyz(t4);
	const int tri_word = fbits(h4.w);
	const int tri = tri_word & (int)((1u << S.tri_bits) - 1u); // (the shading class rides above the index)
	if (tri_word < 0) {
		if (S.bg_node >= 0) { // pt_integrator.cl:214-275 (throughput is exactly 1 for primaries)
			typename Tbl<LDS>::Node bg = S.nodes + S.bg_node;
			f2 uv = latlong_uv(xyz(d4));
			f3 kd = mat_color(uv, bg->k, bg->tex, S);
			f3 add = A.bounce == 0 ? kd : thr * kd;
			terminal_add(A, cell, add);
			R.miss = 1;
		}
		return;
	}
	R.hit = 1;
	Rng rng = {seed, gid_ref}; // pt_integrator.cl:81-84
	const f2 sample0 = rng_next(rng), sample1 = rng_next(rng), sample2 = rng_next(rng);
	const f3 in_dir = -xyz(d4);
	// surfaceInit, util/surface.cl:12-33
	const float bu = h4.x, bv = h4.y, bw = 1.0f - (bu + bv); // intersect.cl:283-286
	const uint32_t off = (uint32_t)tri * 3;
	Surf sf;
	{
		float4 a = S.vertices[off], b = S.vertices[off + 1], c = S.vertices[off + 2];
		sf.p = mk3(bw * a.x + bu * b.x + bv * c.x, bw * a.y + bu * b.y + bv * c.y, bw * a.z + bu * b.z + bv * c.z);
		a = S.normals[off]; b = S.normals[off + 1]; c = S.normals[off + 2];
		sf.n = normalize(mk3(bw * a.x + bu * b.x + bv * c.x, bw * a.y + bu * b.y + bv * c.y, bw * a.z + bu * b.z + bv * c.z));
		float2 ua = S.uvs[off], ub = S.uvs[off + 1], uc = S.uvs[off + 2];
		sf.uv = {bw * ua.x + bu * ub.x + bv * uc.x, bw * ua.y + bu * ub.y + bv * uc.y};
	}
	f3 tint = splat(1.0f);
	MatT<LDS> m = select_material(S.mat_index[tri], sf, flags, tint, rng, S);
	material_params(sf, m, S);
	const float in_dot_n = dot(in_dir, sf.n);
	if (m.type == POLARIS_BXDF_EMISSIVE) { // pt_integrator.cl:101-107 (indexed by pixel: SURVEY 5.8)
		if (in_dot_n > 0.0f) {
			f3 add = thr * m.nd->scale * m.kcol;
			terminal_add(A, cell, add);
			R.emit = 1;
		}
		return;
	}
	bool reject = m.type == POLARIS_BXDF_INVALID;
	if (A.bounce >= A.min_rr) { // Russian roulette, :113-125
		float p = pm_max(pm_min(0.5f, 0.2126f * thr.x + 0.7152f * thr.y + 0.0722f * thr.z), 0.01f);
		if (p < sample2.x) reject = true;
		else thr = thr / p;
	}
	if (reject) return;
	f3 out_dir = splat(0.0f);
	float bxdf_pdf = 1.0f, bxdf_weight = 1.0f;
	const f3 bxdf_val = bxdf_sample(sf, m, S, sample0, in_dir, out_dir, bxdf_pdf);
	const float displace = pm_sign(dot(sf.n, out_dir));
	const f3 ind_origin = sf.p + (sf.n * displace) * kEps;  // DISPLACE_BY_EPSILON, :134
	const f3 occ_origin = sf.p + sf.n * kEps;               // :136
	// light selection + sampling + MIS, :139-155
	f3 e_dir = splat(0.0f), e_rad = splat(0.0f);
	float e_pdf = 0.0f, sel_pdf = 0.0f, e_weight = 0.0f, e_dist = 0.0f;
	typename Tbl<LDS>::Light em = nullptr;
	uint32_t e_index = 0;
	if (S.num_emissives > 0) {
		sel_pdf = S.sel_pdf; // native_recip((float)numEmissives), emissiveSelect, emissive_sampler.cl:226-237: the same for every ray
		const int ei = pm_clampi((int)(sample1.x * (int)S.num_emissives), 0, (int)S.num_emissives - 1);
		em = S.emissives + ei;
		e_index = (uint32_t)ei;
		const LightSample L = light_sample(sf, em, (uint32_t)ei, S, sample1);
		e_dir = L.dir; e_rad = L.radiance; e_pdf = L.pdf; e_dist = L.dist;
	}
	const float n_dot_e = pm_max(0.0f, dot(sf.n, e_dir));
	const bool want_nee = maxcomp(e_rad) > 0.0f && e_pdf > 0.0f && n_dot_e > 0.0f; // :158
	if (em) {
		float bxdf_e_pdf;
		f3 bxdf_e_val;
		bxdf_pdf_eval(sf, m, S, in_dir, e_dir, want_nee, bxdf_e_pdf, bxdf_e_val);
		e_weight = (e_pdf * e_pdf) / (e_pdf * e_pdf + bxdf_e_pdf * bxdf_e_pdf);       // POWER_HEURISTIC, :149
		const float e_bxdf_pdf = light_pdf(sf, em, e_index, S, out_dir);
		bxdf_weight = (bxdf_pdf * bxdf_pdf) / (bxdf_pdf * bxdf_pdf + e_bxdf_pdf * e_bxdf_pdf); // :154
		if (want_nee) {
			e_rad = e_rad * (e_weight * bxdf_e_val * thr * n_dot_e / (e_pdf * sel_pdf)); // :160
			if (maxcomp(e_rad) > 0.0f) {
				occ_sink(make_float4(occ_origin.x, occ_origin.y, occ_origin.z, e_dist - kLightEps), // :203
				         make_float4(e_dir.x, e_dir.y, e_dir.z, ibits((int)cell)), make_float4(e_rad.x, e_rad.y, e_rad.z, ibits((int)cell)));
			}
		}
	}
	if ((m.type & (POLARIS_BXDF_CONDUCTOR | POLARIS_BXDF_DIELECTRIC)) != 0) bxdf_weight = 1.0f; // :166-168
	const f3 tp = bxdf_weight * bxdf_val * tint * pm_fabs(dot(sf.n, out_dir)); // :173
	if (maxcomp(tp) > 0.0f && bxdf_pdf > 0.0f && !A.last_bounce) {
		const f3 nt = thr * tp / bxdf_pdf; // :175
		R.emit_ind = true;
		R.ro = make_float4(ind_origin.x, ind_origin.y, ind_origin.z, kFltMax); // :209
		R.rd = make_float4(out_dir.x, out_dir.y, out_dir.z, ibits((int)(path_index | (flags << 24))));
		R.thr = make_float4(nt.x, nt.y, nt.z, 0.0f);
	}
}

constexpr uint32_t kLdsMatNodes = 64, kLdsLights = 16, kLdsTextures = 16;

// Material nodes, emissive records and texture metadata are tiny tables that EVERY ray walks
// through dependent loads (tree node -> texture record -> texel; light -> its material).  When they
// all fit they are staged in LDS once per workgroup (kernel variant LDS = true, chosen by the host),
// which takes those round trips out of the wave's latency chain: the shading kernels are bound by
// exactly that chain times their occupancy (PMC: waves wait on memory 59 % of their life, 4.1 cycles
// per VALU instruction, 4 waves per SIMD).
struct ShadeLds {
	float4 nodes[kLdsMatNodes * 4];
	float4 lights[kLdsLights * 5];
	float4 light_geo[kLdsLights * (kLightGeoFloats / 4)];
	float4 texmeta[kLdsTextures];
};
// Ends in a __syncthreads() in both variants (k_shade_wave relies on it to publish its cursor).
// The three tables are at most 256 + 80 + 16 float4s: every thread issues its (up to) three loads back to back and stores
// them afterwards -- one memory round trip in front of the barrier, not one per table.
static_assert(kLdsMatNodes * 4 <= WG && kLdsLights * 5 <= WG && kLdsTextures <= WG && kLdsLights * (kLightGeoFloats / 4) <= WG, "stage_scene copies each table in one pass");
// `before_barrier` runs after the staging loads are issued and before the barrier (k_shade pins its ray loads there).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <bool LDS, class Hook = NoHook>
__device__ __forceinline__ SceneT<LDS> stage_scene(const SceneDev &Sg, ShadeLds &L, Hook before_barrier = Hook()) {
	SceneT<LDS> S;
	S.vertices = Sg.vertices; S.normals = Sg.normals; S.uvs = Sg.uvs; S.mat_index = Sg.mat_index; S.tex_data = Sg.tex_data;
	S.num_emissives = Sg.num_emissives; S.bg_node = Sg.bg_node; S.num_nodes = Sg.num_nodes; S.num_textures = Sg.num_textures;
	S.tri_bits = Sg.tri_bits; S.sel_pdf = Sg.sel_pdf;
	if constexpr (LDS) { // the host launches this variant only when all three tables fit
		const uint32_t tid = threadIdx.x;
		const bool has_n = tid < Sg.num_nodes * 4, has_l = tid < Sg.num_emissives * 5, has_t = tid < Sg.num_textures;
		const bool has_g = tid < Sg.num_emissives * (kLightGeoFloats / 4);
		float4 vn = make_float4(0, 0, 0, 0), vl = vn, vt = vn, vg = vn;
		if (has_n) vn = reinterpret_cast<const float4 *>(Sg.nodes)[tid];
		if (has_l) vl = reinterpret_cast<const float4 *>(Sg.emissives)[tid];
		if (has_t) vt = reinterpret_cast<const float4 *>(Sg.tex_meta)[tid];
		if (has_g) vg = reinterpret_cast<const float4 *>(Sg.light_geo)[tid];
		if (has_n) L.nodes[tid] = vn;
		if (has_l) L.lights[tid] = vl;
		if (has_t) L.texmeta[tid] = vt;
		if (has_g) L.light_geo[tid] = vg;
		S.light_geo = (typename Tbl<true>::F)(reinterpret_cast<float *>(L.light_geo));
		S.nodes = (typename Tbl<true>::Node)(L.nodes);
		S.emissives = (typename Tbl<true>::Light)(L.lights);
		S.tex_meta = (typename Tbl<true>::TexMeta)(L.texmeta);
	} else {
		S.nodes = Sg.nodes; S.emissives = Sg.emissives; S.tex_meta = Sg.tex_meta; S.light_geo = Sg.light_geo;
	}
	before_barrier();
	__syncthreads();
	return S;
}

// Canonical index of a ray within its chunk = its position in the reference's compacted buffer relative to the chunk's
// first ray: the number of emitting parents (bits of the previous step's emit mask) below its own parent.
__device__ __forceinline__ uint32_t canonical_index(const uint32_t (&mask)[8], uint32_t parent) {
	const uint32_t word = parent >> 5, lower = (1u << (parent & 31)) - 1u;
	uint32_t c = 0;
#pragma unroll
	for (uint32_t w = 0; w < 8; w++) c += w < word ? __popc(mask[w]) : (w == word ? __popc(mask[w] & lower) : 0);
	return c;
}

// k_shade: one workgroup per chunk, one lane per live ray.
//
// What the reference fixes is the ORDER of a sample's live rays (its PRNG stream is keyed by a ray's position in the
// compacted buffer, pt_integrator.cl:81, and compaction is stable in work-item order) -- not where a ray physically sits.
// Round 1 kept the physical order equal to the reference order (stable in-place compaction: ranks from ballots, wave
// totals through LDS, one barrier).  In-kernel stamps showed what that barrier costs: a wave spent 25-30 % of its life
// waiting there for the slowest wave of its workgroup (the one holding the rough-dielectric rays), wave slots and
// registers held all the while.  Now the order is DATA (Streams::emask): a ray carries its parent's canonical index, the
// chunk's emit mask turns it into the ray's own, and the rays themselves are appended to the chunk in whatever order the
// waves finish (one LDS atomic per wave and stream reserves the slots).  No barrier after shading: a wave stores its rays
// and retires; the last wave of the workgroup to finish publishes the chunk's counts and mask.
//
// SORT (the bounces after the first): with the physical order free, the chunk's rays are also shaded in the order of their
// material's SHADING CLASS (scene_layout.h, shading_classes: which BxDF leaves and texture operators the material tree can
// reach; it rides in the hit record above the triangle index).  A wave of bounce rays holds diffuse walls, glass, metal
// and misses side by side and executes the union of their code paths: 25 % of the lanes of a VALU instruction were live
// there (PMC), and the same box with diffuse materials only shades in half the time.  Counting sort over <= 16 classes:
// per-wave ballots, the 64 (class, wave) counts scanned by every wave with shuffles, the rays (already in registers)
// moved to their lane through LDS -- no memory round trip is added.  Most waves then run ONE short path and retire early.
//
// In-place safety: a chunk's input streams (ray_d, thr, hit) are overwritten by its own outputs.  Every wave's inputs are
// in registers before the workgroup's last barrier (SORT: the exchange barrier; otherwise the staging barrier, in front
// of which every wave waits for its loads), and no wave stores before that barrier.
//
// Occupancy: built for 6 waves per SIMD (<= 80 VGPRs, no spill since the SLP vectoriser is off: its register pairs cost ~10
// VGPRs; with it the kernel needed 90 and 6 waves spilled).  21 KB of LDS per workgroup: 6 workgroups per CU fit.
#ifndef POLARIS_SHADE_WAVES
#define POLARIS_SHADE_WAVES 6
#endif
template <bool LDS, bool SORT, bool FIRST>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(POLARIS_SHADE_WAVES, POLARIS_SHADE_WAVES))) void k_shade(Streams st, SceneDev Sg, ShadeArgs A) {
	__shared__ ShadeLds lds;
	__shared__ uint32_t s_cnt[SORT ? 4 : 1][16];       // rays per (wave, class)
	__shared__ float4 x_d[SORT ? WG : 1], x_t[SORT ? WG : 1], x_h[SORT ? WG : 1]; // the rays in class order
	__shared__ uint32_t s_emit[8];                     // this step's emit mask: bit c = the ray with canonical index c emitted an indirect ray
	__shared__ uint32_t s_tot[6];                      // indirect rays, shadow rays, hits, misses, emitter hits of the chunk; waves finished
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// The lane's ray is requested first, from its own slot whether or not the slot is live (a chunk always has its 256 slots: a
	// dead slot's bytes are read and never looked at), so the request does not wait for the chunk's live count: count, ray, emit
	// mask, seed and tables are one round trip.  (Round 4 timing builds: a quarter of the sorted kernel's time passes before the
	// sort, 7 % in the sort, the rest in shade_ray and its stores.  That first quarter is the ray streams themselves coming from
	// HBM -- taking the count's scalar round trip out of it, as here, measured +-0.)
	const size_t base = (size_t)blockIdx.x * WG;
	const size_t my = base + tid;
#ifndef POLARIS_SHADE_LOAD
#define POLARIS_SHADE_LOAD 1
#endif
	// Only LIVE slots are requested (POLARIS_SHADE_LOAD = 1, round 5).  Round 4 requested every lane's slot at once, live or not, so that
	// the ray loads did not wait for the chunk's count -- and read the 44 bytes of every dead slot (10 / 30 / 42 % of a chunk at
	// bounces 1 / 2 / 3): 1.07 GB per headline frame, the larger part of this kernel's 1.30 x traffic over its algorithmic bytes
	// (now 1.13 x), for no measurable time either way (profiles/r05_shade_dead_slots_ab.txt; 0 = the round-4 form, 2 = slots 0..127 at
	// once and the rest only if live: A/B aids).  Camera rays (FIRST) fill every slot: they keep the unconditional form.
	float4 d4 = make_float4(0, 0, 0, 0), t4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f), h4 = make_float4(0, 0, 0, 0);
	const bool early = FIRST || POLARIS_SHADE_LOAD == 0 || (POLARIS_SHADE_LOAD == 2 && tid < 128);
	if (early) { d4 = st.ray_d[my]; if (!FIRST) t4 = st.thr[my]; h4 = load_hit(st, my); } // (camera rays carry no throughput: it is 1)
	// (the sample's seed and the chunk's position in the reference's buffer: requested with everything else, used by shade_ray)
	const uint32_t s = blockIdx.x / (A.Npad / WG);
	const uint32_t seed = A.seeds[(size_t)(A.first_sample + s) * A.seed_stride + 1 + A.bounce], pfx0 = st.pfx[blockIdx.x];
	const uint32_t cnt = st.cnt_ray[blockIdx.x];
	if (!early && tid < cnt) { d4 = st.ray_d[my]; t4 = st.thr[my]; h4 = load_hit(st, my); }
	if (cnt == 0) { // uniform exit: nothing live in this workgroup
		if (tid == 0) { st.cnt_occ[blockIdx.x] = 0; st.wg_stat[blockIdx.x] = 0; } // (no rays come out of it: its emit mask will not be read)
		return;
	}
	uint32_t pmask[8]; // the previous step's emit mask of this chunk (uniform)
#pragma unroll
	for (int w = 0; w < 8; w++) pmask[w] = FIRST ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)A.emask_in[(size_t)blockIdx.x * 8 + w]); // (kept in scalar registers)
	if (tid < 8) s_emit[tid] = 0; // (published by the barrier that ends stage_scene)
	if (tid < 6) s_tot[tid] = 0;
	// in-place safety without SORT: this wave's rays are in registers before the staging barrier (the empty asm consumes them)
	const SceneT<LDS> S = stage_scene<LDS>(Sg, lds, [&]() { if (!SORT) asm volatile("" ::"v"(d4.w), "v"(t4.w), "v"(h4.w)); });
	const unsigned long long below = (1ull << lane) - 1ull;
	if (SORT) {
		uint32_t key = 16; // no ray
		if (tid < cnt) {
			const int w = fbits(h4.w);
			key = w < 0 ? 0u : min((uint32_t)w >> S.tri_bits, 15u);
		}
		if (lane < 16) s_cnt[wave][lane] = 0; // (LDS operations of one wave execute in order)
		uint32_t rank_in = 0;
		unsigned long long todo = __ballot(key < 16);
		while (todo != 0ull) { // one round per class present in this wave
			const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, __ffsll((long long)todo) - 1);
			const unsigned long long m = __ballot(key == k);
			if (key == k) rank_in = __popcll(m & below);
			if (lane == 0) s_cnt[wave][k] = __popcll(m);
			todo &= ~m;
		}
		__syncthreads();
		// exclusive scan of the 64 counts in (class, wave) order, by every wave for itself
		const uint32_t mine = s_cnt[lane & 3][lane >> 2];
		uint32_t incl = mine;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t up = __shfl_up(incl, d);
			if ((int)lane >= d) incl += up;
		}
		const uint32_t dest = __shfl(incl - mine, (int)((key & 15u) * 4 + wave)) + rank_in;
		if (key < 16) { x_d[dest] = d4; x_t[dest] = t4; x_h[dest] = h4; }
		__syncthreads();
		if (tid < cnt) { d4 = x_d[tid]; t4 = x_t[tid]; h4 = x_h[tid]; }
	}
	if (wave * 64 < cnt) { // (uniform per wave)
		ShadeOut R;
		R.emit_ind = false;
		R.hit = R.miss = R.emit = 0;
		uint32_t canon = 0;
		if (tid < cnt) {
			canon = FIRST ? tid : canonical_index(pmask, (uint32_t)fbits(t4.w));
			shade_ray(S, A, s, seed, pfx0 + canon, d4, t4, h4, R, [&](float4 oo, float4 od, float4 oe) {
				// the lanes that get here emit a shadow ray: one of them reserves their slots, all store at once
				const unsigned long long m = __ballot(true);
				const int first = __ffsll((long long)m) - 1;
				uint32_t at = 0;
				if ((int)lane == first) at = atomicAdd(&s_tot[1], (uint32_t)__popcll(m));
				const size_t d = base + (uint32_t)__builtin_amdgcn_readlane((int)at, first) + __popcll(m & below);
				st.occ_o[d] = oo; st.occ_d[d] = od; st.occ_e[d] = oe;
			});
		}
		// ---- append the wave's rays to the chunk ------------------------------------------------
		const unsigned long long m_ind = __ballot(R.emit_ind);
		const unsigned long long mh = __ballot(R.hit != 0), mm = __ballot(R.miss != 0), me = __ballot(R.emit != 0);
		uint32_t at_ind = 0;
		if (lane == 0) {
			if (m_ind) at_ind = atomicAdd(&s_tot[0], (uint32_t)__popcll(m_ind));
			if (mh) atomicAdd(&s_tot[2], (uint32_t)__popcll(mh));
			if (mm) atomicAdd(&s_tot[3], (uint32_t)__popcll(mm));
			if (me) atomicAdd(&s_tot[4], (uint32_t)__popcll(me));
		}
		at_ind = __builtin_amdgcn_readfirstlane(at_ind);
		if (R.emit_ind) {
			atomicOr(&s_emit[canon >> 5], 1u << (canon & 31));
			const size_t d = base + at_ind + __popcll(m_ind & below);
			R.thr.w = ibits((int)canon); // the child's parent, in canonical order
			store_ray_o(st, d, R.ro); st.ray_d[d] = R.rd; st.thr[d] = R.thr;
		}
	}
	// ---- the last wave to get here publishes the chunk's counts and emit mask -------------------
	__threadfence_block();
	uint32_t arrived = 0;
	if (lane == 0) arrived = atomicAdd(&s_tot[5], 1u);
	if (__builtin_amdgcn_readfirstlane(arrived) == 3u) {
		__threadfence_block();
		if (lane == 0) {
			st.cnt_ray[blockIdx.x] = s_tot[0];
			st.cnt_occ[blockIdx.x] = s_tot[1];
			// no global atomics here: 32 Ki workgroups adding to three shared counters serialise at the
			// memory side (~80 atomics/us on one address); k_scan sums these words instead
			st.wg_stat[blockIdx.x] = s_tot[2] | (s_tot[3] << 10) | (s_tot[4] << 20);
		}
		if (lane < 8) A.emask_out[(size_t)blockIdx.x * 8 + lane] = s_emit[lane];
	}
}

// k_shade_wave: the same shading with a WAVE as the unit of work, for the sparse late bounces.  k_shade's launch time is
// (#chunks / resident workgroups) x (latency chain of one workgroup) however few rays a chunk still holds, so the sparse
// bounces cost as much as the dense first one.  Here persistent waves pull GROUPS of kSparseGroup consecutive chunks
// (dealt round-robin to workgroups, LDS cursor inside) and walk the group's live rays 64 at a time, chunk after chunk:
// after Russian roulette a chunk holds ~20 rays, so one pass shades the survivors of about three chunks with full lanes
// (round 1 walked one chunk per pass: 11 % of the lanes of a VALU instruction were live).  Every lane knows its ray's
// chunk (its sample, seed, reference position and emit mask are per lane); outputs go to the ray's own chunk through
// per-chunk LDS counters (same canonical-order protocol as k_shade; never the first bounce).
// In-place safety: a chunk's rays are read in slot order, a pass's reads precede its writes (program order of one wave),
// and a chunk never holds more emitted rays than rays already read from it.
constexpr int kSparseGroup = 8;
template <bool LDS>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(4, 4))) // (fits 128 VGPRs without a spill: 4 waves per SIMD instead of 3)
void k_shade_wave(Streams st, SceneDev Sg, ShadeArgs A, uint32_t num_chunks) {
	constexpr int G = kSparseGroup;
	__shared__ ShadeLds lds;
	__shared__ uint32_t wg_cursor;
	__shared__ uint32_t w_emit[4][G][8]; // per wave and chunk of its group: the emit mask
	__shared__ uint32_t w_cnt[4][G][3];  // ... indirect rays, shadow rays, event counters (hits | misses << 10 | emitter hits << 20)
	if (threadIdx.x == 0) wg_cursor = 0;
	for (uint32_t i = threadIdx.x; i < 4 * G * 8; i += WG) (&w_emit[0][0][0])[i] = 0;
	for (uint32_t i = threadIdx.x; i < 4 * G * 3; i += WG) (&w_cnt[0][0][0])[i] = 0;
	const SceneT<LDS> S = stage_scene<LDS>(Sg, lds); // (ends in the __syncthreads that also publishes the LDS words above)
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t wgs_per_sample = A.Npad / WG;
	const uint32_t num_groups = (num_chunks + G - 1) / G;
	for (;;) {
		uint32_t c = 0;
		if (lane == 0) c = atomicAdd(&wg_cursor, 1u);
		const uint32_t group = blockIdx.x + __builtin_amdgcn_readfirstlane(c) * gridDim.x;
		if (group >= num_groups) break;
		const uint32_t chunk0 = group * G;
		// lane k < G holds chunk k's live count and the group's exclusive prefix over them
		const uint32_t my_cnt = (lane < (uint32_t)G && chunk0 + lane < num_chunks) ? st.cnt_ray[chunk0 + lane] : 0u;
		uint32_t incl = my_cnt;
#pragma unroll
		for (int d = 1; d < G; d <<= 1) {
			const uint32_t up = __shfl_up(incl, d);
			if ((int)lane >= d) incl += up;
		}
		const uint32_t excl = incl - my_cnt;
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, G - 1);
		for (uint32_t j = 0; j < total; j += 64) {
			const uint32_t r = j + lane; // the group's r-th live ray
			const bool live = r < total;
			uint32_t k = 0, first = 0; // its chunk within the group, and the rays of the group before that chunk
#pragma unroll
			for (int q = 1; q < G; q++) {
				const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)excl, q);
				if (r >= e) { k = q; first = e; }
			}
			ShadeOut R;
			R.emit_ind = false;
			R.hit = R.miss = R.emit = 0;
			uint32_t canon = 0;
			const uint32_t chunk = chunk0 + k;
			const size_t base = (size_t)chunk * WG;
			if (live) {
				const uint32_t idx = r - first;
				const float4 t4 = st.thr[base + idx];
				const uint4 m0 = reinterpret_cast<const uint4 *>(A.emask_in)[(size_t)chunk * 2], m1 = reinterpret_cast<const uint4 *>(A.emask_in)[(size_t)chunk * 2 + 1];
				const uint32_t pmask[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
				canon = canonical_index(pmask, (uint32_t)fbits(t4.w));
				const uint32_t s = chunk / wgs_per_sample;
				const uint32_t seed = A.seeds[(size_t)(A.first_sample + s) * A.seed_stride + 1 + A.bounce];
				shade_ray(S, A, s, seed, st.pfx[chunk] + canon, st.ray_d[base + idx], t4, load_hit(st, base + idx), R, [&](float4 oo, float4 od, float4 oe) {
					const size_t d = base + atomicAdd(&w_cnt[wave][k][1], 1u); // (shadow rays have no order to keep)
					st.occ_o[d] = oo; st.occ_d[d] = od; st.occ_e[d] = oe;
				});
			}
			if (R.emit_ind) {
				atomicOr(&w_emit[wave][k][canon >> 5], 1u << (canon & 31));
				const size_t d = base + atomicAdd(&w_cnt[wave][k][0], 1u); // any free slot of the ray's chunk: the order is in thr.w
				R.thr.w = ibits((int)canon);
				store_ray_o(st, d, R.ro); st.ray_d[d] = R.rd; st.thr[d] = R.thr;
			}
			if (live && (R.hit | R.miss | R.emit) != 0) atomicAdd(&w_cnt[wave][k][2], R.hit | (R.miss << 10) | (R.emit << 20));
		}
		// publish the group's chunks (LDS operations of one wave execute in order: the atomics above are done) and clear the wave's words
		if (lane < (uint32_t)G && chunk0 + lane < num_chunks) {
			st.cnt_ray[chunk0 + lane] = w_cnt[wave][lane][0];
			st.cnt_occ[chunk0 + lane] = w_cnt[wave][lane][1];
			st.wg_stat[chunk0 + lane] = w_cnt[wave][lane][2];
			w_cnt[wave][lane][0] = 0; w_cnt[wave][lane][1] = 0; w_cnt[wave][lane][2] = 0;
		}
		for (uint32_t i = lane; i < (uint32_t)G * 8; i += 64) {
			if (chunk0 + i / 8 < num_chunks) A.emask_out[(size_t)chunk0 * 8 + i] = (&w_emit[wave][0][0])[i];
			(&w_emit[wave][0][0])[i] = 0;
		}
	}
}

// ------------------------------------------------------------------------------------------
// Function-level probes (test taps, polaris_hip_probe): the device functions of shading.h on caller-supplied inputs, one
// lane per probe, through the SAME table staging the shade kernels use -- so the samplers and BxDFs are checked against
// the CPU oracle on the GPU one function at a time, not only through whole traces.
// ------------------------------------------------------------------------------------------
enum ProbeKind { kProbeBxdf = 0, kProbeTexture = 1, kProbeEmissive = 2, kProbeMaterial = 3 };
constexpr uint32_t kProbeIn[4] = {13, 2, 11, 8}, kProbeOut[4] = {11, 7, 9, 18};

template <bool LDS>
__global__ __launch_bounds__(WG) void k_probe(SceneDev Sg, int kind, uint32_t index, uint32_t n, const float *in, float *out) {
	__shared__ ShadeLds lds;
	const SceneT<LDS> S = stage_scene<LDS>(Sg, lds);
	const uint32_t i = blockIdx.x * WG + threadIdx.x;
	if (i >= n) return;
	if (kind == kProbeBxdf) { // in: normal[3] uv[2] in_dir[3] sample[2] eval_dir[3]; out: bxdfGetSample value[3] dir[3] pdf | bxdfGetPdf | bxdfEval[3] (bxdf/bxdf.cl:31-105)
		const float *p = in + (size_t)i * 13;
		float *o = out + (size_t)i * 11;
		Surf sf;
		sf.p = splat(0.0f); sf.n = mk3(p[0], p[1], p[2]); sf.uv = {p[3], p[4]};
		const f3 wi = mk3(p[5], p[6], p[7]), wo = mk3(p[10], p[11], p[12]);
		MatT<LDS> m;
		m.nd = S.nodes + index;
		m.type = m.nd->type;
		m.int_ior = m.nd->int_ior; m.ext_ior = m.nd->ext_ior;
		material_params(sf, m, S);
		f3 dir = splat(0.0f);
		float pdf = 0.0f;
		const f3 v = bxdf_sample(sf, m, S, f2{p[8], p[9]}, wi, dir, pdf);
		float pdf2;
		f3 ev;
		bxdf_pdf_eval(sf, m, S, wi, wo, true, pdf2, ev);
		o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = dir.x; o[4] = dir.y; o[5] = dir.z; o[6] = pdf;
		o[7] = pdf2; o[8] = ev.x; o[9] = ev.y; o[10] = ev.z;
	} else if (kind == kProbeTexture) { // in: uv[2]; out: texGetSample3f[3] | texGetSample1f | texGetBumpSample3f[3] (texture_sampler.cl:14-252)
		const float *p = in + (size_t)i * 2;
		float *o = out + (size_t)i * 7;
		const TexQuad q = tex_fetch(f2{p[0], p[1]}, (int)index, S);
		const f3 c = quad_sample3(q), b = quad_bump3(q);
		o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.x; o[4] = b.x; o[5] = b.y; o[6] = b.z;
	} else if (kind == kProbeMaterial) { // in: normal[3] uv[2] PRNG state[2] (bits) path flags (bits); out: see polaris_hip.h (matSelectNode, material_sampler.cl:21-95)
		const float *p = in + (size_t)i * 8;
		float *o = out + (size_t)i * 18;
		Surf sf;
		sf.p = splat(0.0f); sf.n = mk3(p[0], p[1], p[2]); sf.uv = {p[3], p[4]};
		Rng rng = {(uint32_t)fbits(p[5]), (uint32_t)fbits(p[6])};
		uint32_t flags = (uint32_t)fbits(p[7]);
		f3 tint = splat(1.0f);
		const MatT<LDS> m = select_material(index, sf, flags, tint, rng, S);
		o[0] = ibits((int)m.type); o[1] = m.int_ior; o[2] = m.ext_ior;
		o[3] = sf.n.x; o[4] = sf.n.y; o[5] = sf.n.z; o[6] = tint.x; o[7] = tint.y; o[8] = tint.z;
		o[9] = ibits((int)flags); o[10] = ibits((int)rng.sx); o[11] = ibits((int)rng.sy);
		for (int k = 0; k < 3; k++) { o[12 + k] = m.nd->k[k]; o[15 + k] = m.nd->t[k]; }
	} else { // in: point[3] normal[3] sample[2] pdf_dir[3]; out: emissiveGetSample radiance[3] dir[3] pdf dist | emissiveGetPdf (emissive_sampler.cl:176-223)
		const float *p = in + (size_t)i * 11;
		float *o = out + (size_t)i * 9;
		Surf sf;
		sf.p = mk3(p[0], p[1], p[2]); sf.n = mk3(p[3], p[4], p[5]); sf.uv = {0.0f, 0.0f};
		typename Tbl<LDS>::Light em = S.emissives + index;
		const LightSample L = light_sample(sf, em, index, S, f2{p[6], p[7]});
		o[0] = L.radiance.x; o[1] = L.radiance.y; o[2] = L.radiance.z; o[3] = L.dir.x; o[4] = L.dir.y; o[5] = L.dir.z;
		o[6] = L.pdf; o[7] = L.dist;
		o[8] = light_pdf(sf, em, index, S, mk3(p[8], p[9], p[10]));
	}
}

// polaris_hip_selftest_rcp: rcp_det against the correctly rounded 1.0f / x over ALL 2^32 float bit patterns.
// out[0] = patterns with lo <= |x| <= hi that differ, out[1] = patterns outside that differ, out[2] = one differing pattern inside.
__global__ __launch_bounds__(WG) void k_rcp_sweep(float lo, float hi, unsigned long long *out) {
	const uint64_t stride = (uint64_t)gridDim.x * WG;
	unsigned long long bad_in = 0, bad_out = 0, sample = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < (1ull << 32); i += stride) {
		const float x = ibits((int)(uint32_t)i);
		const float want = pm_rcp(x), got = rcp_det(x);
		if (fbits(want) != fbits(got) && !(want != want && got != got)) { // (any NaN equals any NaN)
			const float ax = pm_fabs(x);
			if (ax >= lo && ax <= hi) { bad_in++; sample = i; }
			else bad_out++;
		}
	}
	if (bad_in) { atomicAdd(&out[0], bad_in); out[2] = sample; }
	if (bad_out) atomicAdd(&out[1], bad_out);
}

// Arbitrary rays into the stream layout of the traversal kernels (polaris_hip_probe_intersect): slot i = ray i.
__global__ __launch_bounds__(WG) void k_probe_rays(Streams st, const float *rays, uint32_t n, int any_hit) {
	const uint32_t i = blockIdx.x * WG + threadIdx.x;
	if (threadIdx.x == 0) {
		const uint32_t live = blockIdx.x * WG < n ? min((uint32_t)WG, n - blockIdx.x * WG) : 0u;
		st.cnt_ray[blockIdx.x] = any_hit ? 0u : live;
		st.cnt_occ[blockIdx.x] = any_hit ? live : 0u;
	}
	if (i >= n) return;
	const float *r = rays + (size_t)i * 8;
	const float4 o = make_float4(r[0], r[1], r[2], r[3]), d = make_float4(r[4], r[5], r[6], ibits((int)i));
	if (any_hit) {
		st.occ_o[i] = o; st.occ_d[i] = d; st.occ_e[i] = make_float4(1.0f, 0.0f, 0.0f, 0.0f); // an unoccluded ray adds 1 to lsum[i].x
		st.lsum[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	} else {
		st.ray_o[i] = o; st.ray_d[i] = d;
		st.hit[i] = make_float4(0.0f, 0.0f, 0.0f, ibits(-1));
	}
}

// ------------------------------------------------------------------------------------------
// Segmented exclusive scan of the per-workgroup live-ray counts, one workgroup per sample:
// pfx[wg] = number of live rays in earlier workgroups of the same sample = position of this
// workgroup's first ray in the reference's compacted buffer.  Also books the ray counters.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(Streams st, uint32_t wgs_per_sample, uint32_t bounce, int count_next_rays,
                                               unsigned long long *stats) {
	__shared__ uint32_t part[1024];
	__shared__ uint32_t part_occ[1024];
	__shared__ uint32_t red[3][16];
	const uint32_t tid = threadIdx.x;
	const uint32_t per = (wgs_per_sample + 1023) / 1024;
	const uint32_t b0 = blockIdx.x * wgs_per_sample;
	const uint32_t lo = tid * per, hi = min(lo + per, wgs_per_sample);
	uint32_t sum = 0, socc = 0, s_hit = 0, s_miss = 0, s_emit = 0;
	for (uint32_t i = lo; i < hi; i++) {
		sum += st.cnt_ray[b0 + i];
		socc += st.cnt_occ[b0 + i];
		const uint32_t w = st.wg_stat[b0 + i]; // shading statistics of the workgroup (k_shade)
		s_hit += w & 1023u; s_miss += (w >> 10) & 1023u; s_emit += (w >> 20) & 1023u;
	}
	part[tid] = sum;
	part_occ[tid] = socc;
	// wave reduction of the three shading counters, then 16 wave totals through LDS
	for (int s = 32; s > 0; s >>= 1) { s_hit += __shfl_xor(s_hit, s); s_miss += __shfl_xor(s_miss, s); s_emit += __shfl_xor(s_emit, s); }
	if ((tid & 63) == 0) { red[0][tid >> 6] = s_hit; red[1][tid >> 6] = s_miss; red[2][tid >> 6] = s_emit; }
	__syncthreads();
	// Hillis-Steele inclusive scan over the 1024 partials
	for (uint32_t off = 1; off < 1024; off <<= 1) {
		uint32_t v = tid >= off ? part[tid - off] : 0, vo = tid >= off ? part_occ[tid - off] : 0;
		__syncthreads();
		part[tid] += v;
		part_occ[tid] += vo;
		__syncthreads();
	}
	uint32_t run = part[tid] - sum;
	for (uint32_t i = lo; i < hi; i++) { st.pfx[b0 + i] = run; run += st.cnt_ray[b0 + i]; }
	if (tid == 1023) {
		if (count_next_rays && part[1023]) atomicAdd(&stats[ST_RAYS_BOUNCE + bounce + 1], (unsigned long long)part[1023]);
		if (part_occ[1023]) atomicAdd(&stats[ST_OCCL_BOUNCE + bounce], (unsigned long long)part_occ[1023]);
	}
	if (tid < 3) {
		uint32_t t = 0;
		for (int w = 0; w < 16; w++) t += red[tid][w];
		if (t) atomicAdd(&stats[(tid == 0 ? ST_HITS_BOUNCE : (tid == 1 ? ST_MISSES_BOUNCE : ST_EMITTERS_BOUNCE)) + bounce], (unsigned long long)t);
	}
}

// accumulateEmissiveSamples (pt_integrator.cl:278-296) for a whole batch at once: one workgroup per chunk adds the NEE records its
// unoccluded shadow rays left (nee_result: vis), bounce after bounce, to the chunk's 256 per-path cells and writes them back
// as one coalesced 4 KB store.  A path has at most one shadow ray per bounce, so the adds of one bounce never collide, and the
// barrier between bounces keeps a path's terms in the reference's order: ((nee_0 + nee_1) + ...) + terminal contribution -- the
// sums the per-bounce read-modify-writes used to produce, bit for bit.
struct FoldArgs { const float4 *nee[POLARIS_MAX_BOUNCES]; const uint8_t *vis[POLARIS_MAX_BOUNCES]; const uint32_t *cnt[POLARIS_MAX_BOUNCES]; uint32_t bounces; };
__global__ __launch_bounds__(WG) void k_fold_nee(FoldArgs F, float4 *lsum) {
	__shared__ float acc[3][WG];
	const uint32_t tid = threadIdx.x;
	const uint32_t base = blockIdx.x * WG;
	acc[0][tid] = 0.0f; acc[1][tid] = 0.0f; acc[2][tid] = 0.0f;
	const float4 term = lsum[base + tid]; // the path's terminal contribution (or 0): requested before the records
	__syncthreads();
	uint32_t any = 0;
	// four bounces at a time: their counts, then their records, are requested together (two memory round trips per group, not
	// two per bounce: the kernel is a chain of dependent loads, 1 GB per batch), the adds stay in bounce order
	for (uint32_t b0 = 0; b0 < F.bounces; b0 += 4) {
		uint32_t n[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; k++) n[k] = b0 + k < F.bounces ? F.cnt[b0 + k][blockIdx.x] : 0u;
		float4 r[4];
		uint32_t v[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; k++) {
			v[k] = 0;
			r[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (tid < n[k]) { v[k] = F.vis[b0 + k][base + tid]; r[k] = F.nee[b0 + k][base + tid]; }
		}
#pragma unroll
		for (uint32_t k = 0; k < 4; k++) {
			if (v[k]) {
				const uint32_t local = (uint32_t)fbits(r[k].w) - base; // the ray's path: a cell of this chunk (rays never leave their chunk)
				acc[0][local] += r[k].x; acc[1][local] += r[k].y; acc[2][local] += r[k].z;
			}
			any |= n[k];
			__syncthreads();
		}
	}
	if (any == 0) return; // (uniform) no shadow ray in any bounce: the cells already hold what they should
	lsum[base + tid] = make_float4(acc[0][tid] + term.x, acc[1][tid] + term.y, acc[2][tid] + term.z, 0.0f);
}

// Batch epilogue: trace accumulator += per-path radiance, samples added in ascending order.
__global__ __launch_bounds__(WG) void k_resolve(const float4 *lsum, float4 *acc, uint32_t K, uint32_t N, uint32_t Npad, uint32_t pixel0) {
	const uint32_t idx = blockIdx.x * WG + threadIdx.x;
	if (idx >= N) return;
	float4 a = acc[pixel0 + idx];
	for (uint32_t s = 0; s < K; s++) {
		const float4 l = lsum[(size_t)s * Npad + idx];
		a.x += l.x; a.y += l.y; a.z += l.z;
	}
	acc[pixel0 + idx] = a;
}

// aggregateAccumulator, kernels/accumulator.cl:13-19 (rows of one block)
__global__ __launch_bounds__(WG) void k_aggregate(const float4 *src, float4 *dst, uint32_t n) {
	const uint32_t i = blockIdx.x * WG + threadIdx.x;
	if (i >= n) return;
	float4 a = dst[i];
	const float4 b = src[i];
	a.x += b.x; a.y += b.y; a.z += b.z;
	dst[i] = a;
}

// tonemapSimpleReinhard, kernels/hdr.cl:5-28
__global__ __launch_bounds__(WG) void k_tonemap(const float4 *acc, uchar4 *fb, uint32_t n, float weight, float exposure) {
	const uint32_t i = blockIdx.x * WG + threadIdx.x;
	if (i >= n) return;
	const float4 a = acc[i];
	const float e = 1.0f / 2.2f;
	float c[3] = {a.x * weight * exposure, a.y * weight * exposure, a.z * weight * exposure};
	unsigned char o[3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		float m = c[k] / (c[k] + 1.0f);
		float v = pm_clamp(pm_pow(m, e), 0.0f, 1.0f) * 255.0f;
		o[k] = (unsigned char)v; // truncation (hdr.cl:22-27)
	}
	fb[i] = make_uchar4(o[0], o[1], o[2], 255);
}

} // namespace pol
