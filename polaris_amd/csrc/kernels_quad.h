// kernels_quad.h -- k_trace4: BVH traversal with FOUR LANES PER RAY over the four-wide tree (scene_layout.h build_quads).
//
// Why (profiles/r03_*, DESIGN.md 3.1 "round 3"): on scenes whose tree does not fit LDS, k_trace -- one lane per ray, one
// 64-byte pair record per lane and step -- is bound by the CU's vector L1, not by HBM, latency or the ALUs.  A wave's node
// fetch is 4 x global_load_dwordx4 with 64 lanes on 64 different lines: 256 tag look-ups, ~75 ns of the CU's one TA / TCP
// per wave-step whatever the occupancy (tests/tools/gather_rate.hip; PMC: TA busy 57-66 %, TCP accesses == TA busy cycles,
// VALU 19-22 % busy, 1 M-triangle terrain and 1 024-instance scene alike).  The same bytes cost the L1 a quarter of that
// when four neighbouring lanes read ONE contiguous line (gather_rate `quad128`: 14.5 ns per wave-step of 16 records, 2.9 x
// the pair-step rate of `lane64`), and dead lanes cost it nothing.  So:
//
//   * a QUAD of lanes (4 q .. 4 q + 3) owns one ray; a wave walks 16 rays.  Lane p tests child p of the four-wide node: the
//     quad reads the node's 128 bytes as one line (2 x 16 B per lane), a triangle leaf's (up to) four triangles as 192
//     contiguous bytes, the ray / instance / leaf records once per quad (same address in all four lanes);
//   * the ray's state (origin, direction, limits, node, stack pointer) is replicated in the quad's four lanes -- registers
//     per LANE are what they were, and everything that decides control flow is quad-uniform, so cross-lane traffic is DPP
//     quad_perm operands only: the child order is a rank from three rotated compares, the next node an AND-reduction, the
//     hit count a field of the ballot;
//   * the node stack is ONE LDS column per ray (a quarter of k_trace's stack per lane): 64 entries fit eight workgroups per
//     CU, so the deep-stack variants no longer cost occupancy (k_trace<..,32,..>: 33 KB per workgroup, 4 per CU);
//   * every lane keeps the closest hit among the triangles IT tested; the quad's minimum distance (two DPP min) is what
//     culls, and the records are reduced once, when the ray ends, under the same total order (t, instance rank, triangle
//     rank) that resolves exact ties in k_trace -- so results stay bit-identical to k_trace, the packet kernel and the oracle.
//
// Per-ray arithmetic is kernels.h's: same slab test (slab_hit_hw), same Moeller-Trumbore, same 1 / det.  Which boxes a ray
// is tested against differs from the pair tree only as build_quads allows (boxes whose test is implied by their children's).
#pragma once

#include "kernels.h"

namespace pol {

struct QuadNode { float4 c[8]; }; // c[2 k] = child k's lo.xyz | ref, c[2 k + 1] = hi.xyz | cull factor

struct Bvh4Dev {
	const QuadNode *quads;
	const int2 *leaves;
	const TriRec *tris;
	const InstRec *insts; // meta.z = the instance's root in `quads` (meta.x is its root in the pair tree)
	int root_ref;
	int root_is_instance; // single-instance scenes: that instance's record by value (see BvhDev)
	InstRec root_inst;
};

// DPP quad_perm controls: lane p of every quad reads lane sel[p] of the same quad
constexpr int quad_perm(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int kQuadXor1 = quad_perm(1, 0, 3, 2), kQuadXor2 = quad_perm(2, 3, 0, 1);
constexpr int kQuadRot1 = quad_perm(1, 2, 3, 0), kQuadRot2 = quad_perm(2, 3, 0, 1), kQuadRot3 = quad_perm(3, 0, 1, 2);
template <int CTRL> __device__ __forceinline__ int qperm(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ float qperm(float v) { return ibits(qperm<CTRL>(fbits(v))); }
template <int K> __device__ __forceinline__ float qbcast(float v) { return qperm<quad_perm(K, K, K, K)>(v); }
template <int K> __device__ __forceinline__ int qbcast(int v) { return qperm<quad_perm(K, K, K, K)>(v); }

#ifndef POLARIS_REFILL_MIN4
#define POLARIS_REFILL_MIN4 8
#endif
#ifndef POLARIS_STRAGGLERS4
#define POLARIS_STRAGGLERS4 4
#endif
constexpr int kRefillMin4 = POLARIS_REFILL_MIN4;   // idle rays (of a wave's 16) before it fetches new ones
constexpr int kStragglers4 = POLARIS_STRAGGLERS4;  // the inner loop is left once fewer rays than this are still descending

template <bool ANY_HIT, int STACK>
__global__ __launch_bounds__(WG) void k_trace4(Streams st, Bvh4Dev B, uint32_t num_chunks, float4 *acc, unsigned long long *stats) {
	constexpr int RAYS = WG / 4;        // rays in flight per workgroup
	constexpr int kStride = STACK + 1;  // words per ray: entry 0 is the dummy below an empty stack; odd, so rays at equal depth use different banks
	__shared__ int stk[RAYS * kStride];
	__shared__ uint32_t wg_cursor;
	if (threadIdx.x == 0) wg_cursor = 0;
	__syncthreads();
	char *const stk_bytes = reinterpret_cast<char *>(&stk[0]);
	const uint32_t tid = threadIdx.x, lane = tid & 63, p = lane & 3;
	const uint32_t sp0 = (tid >> 2) * (uint32_t)(kStride * sizeof(int)); // byte offset of the ray's dummy entry = its empty stack
	auto read_ref = [&](uint32_t at) -> int { return *reinterpret_cast<const int *>(stk_bytes + at); };
	auto write_ref = [&](uint32_t at, int ref) { *reinterpret_cast<int *>(stk_bytes + at) = ref; };
	const unsigned long long kLead = 0x1111111111111111ull;              // lane 0 of every quad
	const unsigned long long below_quad = (1ull << (lane & 60u)) - 1ull;  // the lanes of the quads before this one
	const uint32_t *cnts = ANY_HIT ? st.cnt_occ : st.cnt_ray;
	const float4 *src_o = ANY_HIT ? st.occ_o : st.ray_o;
	const float4 *src_d = ANY_HIT ? st.occ_d : st.ray_d;
	const float4 *quads4 = reinterpret_cast<const float4 *>(B.quads) + 2 * p; // this lane's child slot of node 0
	const float kInf = __builtin_inff();

	uint32_t chunk = 0, off = 0, cnt = 0; // wave-uniform queue state
	bool drained = false;
	// per-ray state, identical in the four lanes of its quad
	uint32_t slot = 0, sp = sp0;
	f3 o = {0, 0, 0}, d = {0, 0, 0}, inv = {0, 0, 0};
	float maxDist = 0.0f, best_tq = 0.0f; // best_tq: the quad's closest hit distance so far (what culls)
	int cur = kIdle, cell = 0;
	uint32_t irank = 0, unocc = 0;
	f3 nee = {0, 0, 0}, acc_old = {0, 0, 0};
	// per-lane: the closest hit among the triangles THIS lane tested
	float best_t = 0.0f, best_u = 0.0f, best_v = 0.0f;
	int best_tri = -1;
	uint32_t best_irank = 0, best_trank = 0;

	// the ray enters an instance: mul4x1 / mul3x1 (util/transform.cl:9-26), intersect.cl:239-252
	auto enter = [&](float4 r0, float4 r1, float4 r2) {
		const f3 no = {r0.x * o.x + r0.y * o.y + r0.z * o.z + r0.w, r1.x * o.x + r1.y * o.y + r1.z * o.z + r1.w, r2.x * o.x + r2.y * o.y + r2.z * o.z + r2.w};
		const f3 nd = {r0.x * d.x + r0.y * d.y + r0.z * d.z, r1.x * d.x + r1.y * d.y + r1.z * d.z, r2.x * d.x + r2.y * d.y + r2.z * d.z};
		o = no; d = nd;
		inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
	};
	// next pending node of the ray (see k_trace)
	auto pop = [&]() {
		const int popped = read_ref(sp);
		const bool empty = sp == sp0;
		const uint32_t spm = sp - 4u;
		cur = (empty || (popped == kExitMarker && spm == sp0)) ? kDone : popped;
		sp = empty ? sp : spm;
	};
	// Moeller-Trumbore on triangle slot t (intersect.cl:255-292, no early exits); closest hit: into the LANE's record
	auto test_triangle = [&](uint32_t t, bool &occluded) {
		const TriRec T = B.tris[t];
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
			occluded = occluded || (ok && tt < maxDist);
		} else {
			const uint32_t trank = (uint32_t)fbits(T.v0.w);
			const bool closer = tt < best_t;
			const bool tie = tt == best_t && best_tri >= 0 && (irank < best_irank || (irank == best_irank && trank < best_trank));
			const bool take = ok && (closer || tie);
			best_t = take ? tt : best_t; best_u = take ? u : best_u; best_v = take ? v : best_v;
			best_tri = take ? fbits(T.e1.w) : best_tri;
			best_irank = take ? irank : best_irank; best_trank = take ? trank : best_trank;
		}
	};
	// after a leaf: shadow rays end at the first blocker any lane of the quad found; closest hits tighten the quad's cull distance
	auto after_leaf = [&](bool occluded) -> bool {
		if (ANY_HIT) {
			const unsigned long long m = __ballot(occluded);
			return (((uint32_t)(m >> (lane & 60u))) & 15u) != 0u;
		}
		float m = __builtin_fminf(best_t, qperm<kQuadXor1>(best_t));
		best_tq = __builtin_fminf(m, qperm<kQuadXor2>(m));
		return false;
	};

	for (;;) {
		// ---- refill idle quads from the workgroup's chunks (k_trace's scheme, 16 ray slots per wave) --------------------
		unsigned long long freem = __ballot(cur == kIdle);
		if (!drained && (freem == ~0ull || __popcll(freem & kLead) >= kRefillMin4)) {
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
				freem = __ballot(cur == kIdle);
				const uint32_t nfree = __popcll(freem & kLead);
				if (nfree == 0) break;
				const uint32_t take = min(cnt - off, nfree);
				const uint32_t rank = __popcll(freem & kLead & below_quad);
				if (cur == kIdle && rank < take) {
					slot = chunk * WG + off + rank;
					const float4 o4 = src_o[slot], d4 = src_d[slot]; // (one address per quad: one L1 access)
					o = xyz(o4); d = xyz(d4);
					maxDist = o4.w;
					cell = fbits(d4.w);
					if (ANY_HIT) {
						const float4 e4 = st.occ_e[slot], a4 = acc[cell]; // what an unoccluded ray adds, and where: fetched beside the ray (see k_trace)
						nee = xyz(e4); acc_old = xyz(a4);
					}
					sp = sp0;
					cur = B.root_ref;
					irank = 0;
					if (B.root_is_instance) {
						const InstRec &I = B.root_inst;
						irank = (uint32_t)I.meta.y;
						write_ref(sp0 + 4u, kExitMarker); // (nothing is ever pending below it: popping it ends the ray without a restore)
						sp = sp0 + 4u;
						cur = I.meta.z;
						enter(I.r0, I.r1, I.r2);
					} else {
						inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)}; // native_recip(ray.dir), intersect.cl:302
					}
					best_t = best_tq = maxDist; best_tri = -1; best_u = best_v = 0.0f; best_irank = best_trank = 0;
				}
				off += take;
			}
		}
		if (__ballot(cur != kIdle) == 0ull) {
			if (drained) break;
			continue;
		}
		// ---- phase 1: descend (intersect.cl:296-328, four children per step) ----------------------------------------------
		if (__ballot(cur >= 0) != 0ull) do {
			if (cur >= 0) {
				const float4 *rec = quads4 + (size_t)(uint32_t)cur * 8u;
				const float4 lo = rec[0], hi = rec[1];
				const int popped = read_ref(sp); // what a pop would deliver (the pushes below go above it)
				const bool empty = sp == sp0;
				const uint32_t spm = sp - 4u;
				float t;
				const bool e = slab_hit_hw(lo, hi, o, inv, maxDist, t); // (an unused slot holds a NaN box: never entered)
				const bool h = e && (ANY_HIT || !(t > best_tq * hi.w)); // closest hit: cull what starts beyond the best hit (+inf factor: never)
				const int ref = fbits(lo.w);
				const unsigned long long hm = __ballot(h);
				const uint32_t qm = ((uint32_t)(hm >> (lane & 60u))) & 15u; // the quad's hit children
				const uint32_t nhit = __popc(qm);
				uint32_t rank; // of this child among the quad's hit children, in visiting order
				if (ANY_HIT) {
					rank = __popc(qm & ((1u << p) - 1u)); // stored order
				} else { // nearest first, slot order among equals: the source lane of rotation r is (p + r) & 3, which precedes p iff it wrapped
					const float key = h ? t : kInf;
					const float k1 = qperm<kQuadRot1>(key), k2 = qperm<kQuadRot2>(key), k3 = qperm<kQuadRot3>(key);
					rank = (uint32_t)(k1 < key || (k1 == key && p == 3u)) + (uint32_t)(k2 < key || (k2 == key && p >= 2u)) + (uint32_t)(k3 < key || (k3 == key && p >= 1u));
				}
				// the first child becomes the current node, the others go on the stack, the next one to visit on top
				int first = (h && rank == 0u) ? ref : -1;
				first &= qperm<kQuadXor1>(first);
				first &= qperm<kQuadXor2>(first);
				if (h && rank != 0u) write_ref(sp + 4u * (nhit - rank), ref);
				const bool none = nhit == 0u;
				const int after_pop = (empty || (popped == kExitMarker && spm == sp0)) ? kDone : popped;
				cur = none ? after_pop : first;
				sp = none ? (empty ? sp : spm) : sp + 4u * (nhit - 1u);
			}
		} while (__popcll(__ballot(cur >= 0)) >= 4 * kStragglers4);
		// ---- phase 2: everything that is not an inner node ----------------------------------------------------------------
		if (cur == kDone) { // the ray is finished
			if (ANY_HIT) { // unoccluded: accumulateEmissiveSamples, pt_integrator.cl:278-296
				if (p == 0u) {
					float *c = reinterpret_cast<float *>(acc + cell);
					c[0] = acc_old.x + nee.x; c[1] = acc_old.y + nee.y; c[2] = acc_old.z + nee.z;
					unocc++;
				}
			} else { // the closest of the four lanes' records, ties by (instance rank, triangle rank) = "first tested wins" (intersect.cl:281)
#define POLARIS_QUAD_REDUCE(CTRL)                                                                                                \
				{                                                                                                                    \
					const float pt = qperm<CTRL>(best_t), pu = qperm<CTRL>(best_u), pv = qperm<CTRL>(best_v);                          \
					const int ptri = qperm<CTRL>(best_tri);                                                                           \
					const uint32_t pir = (uint32_t)qperm<CTRL>((int)best_irank), ptr = (uint32_t)qperm<CTRL>((int)best_trank);          \
					const bool better = pt < best_t || (pt == best_t && (pir < best_irank || (pir == best_irank && ptr < best_trank))); \
					best_t = better ? pt : best_t; best_u = better ? pu : best_u; best_v = better ? pv : best_v;                      \
					best_tri = better ? ptri : best_tri; best_irank = better ? pir : best_irank; best_trank = better ? ptr : best_trank; \
				}
				POLARIS_QUAD_REDUCE(kQuadXor1)
				POLARIS_QUAD_REDUCE(kQuadXor2)
#undef POLARIS_QUAD_REDUCE
				if (p == 0u) st.hit[slot] = make_float4(best_u, best_v, best_t, ibits(best_tri));
			}
			cur = kIdle;
		}
		if (cur == kExitMarker) { // leaving the instance: back to the world-space ray (intersect.cl:330-335)
			const float4 o4 = src_o[slot], d4 = src_d[slot];
			o = xyz(o4); d = xyz(d4);
			inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
			pop();
		}
		if (cur < 0 && cur >= kFirstLeafRef && (((uint32_t)~cur) & 15u) == 0u) { // leaf described by LeafInfo
			const int2 li = B.leaves[((uint32_t)~cur) >> 4];
			if (li.y == 0) { // top-level leaf: enter the mesh instance; the quad reads the record's 64 bytes as one line
				const float4 piece = reinterpret_cast<const float4 *>(B.insts + (-li.x))[p];
				const float4 r0 = make_float4(qbcast<0>(piece.x), qbcast<0>(piece.y), qbcast<0>(piece.z), qbcast<0>(piece.w));
				const float4 r1 = make_float4(qbcast<1>(piece.x), qbcast<1>(piece.y), qbcast<1>(piece.z), qbcast<1>(piece.w));
				const float4 r2 = make_float4(qbcast<2>(piece.x), qbcast<2>(piece.y), qbcast<2>(piece.z), qbcast<2>(piece.w));
				irank = (uint32_t)qbcast<3>(fbits(piece.y));
				const int root = qbcast<3>(fbits(piece.z));
				write_ref(sp + 4u, kExitMarker);
				sp += 4u;
				enter(r0, r1, r2);
				cur = root;
			} else { // more than 15 triangles (a caller's leaf kept whole): four at a time
				bool occluded = false;
				for (uint32_t i = p; i < (uint32_t)li.y; i += 4u) test_triangle((uint32_t)(-li.x) + i, occluded);
				if (after_leaf(occluded)) cur = kIdle; // shadow ray blocked: nothing to add
				else pop();
			}
		}
		// ---- inline leaves (1..15 triangles): lane p tests triangles p, p + 4, ... ------------------------------------------
		{
			const bool tl = cur < 0 && cur >= kFirstLeafRef && (((uint32_t)~cur) & 15u) != 0u;
			if (__ballot(tl) != 0ull) {
				const uint32_t code = (uint32_t)~cur;
				const uint32_t first = code >> 4, ntri = tl ? (code & 15u) : 0u;
				const int popped = read_ref(sp); // what follows the leaf: read beside the triangles
				bool occluded = false;
				uint32_t i = p;
				do {
					if (i < ntri) test_triangle(first + i, occluded);
					i += 4u;
				} while (__ballot(i < ntri) != 0ull);
				const bool blocked = after_leaf(occluded);
				if (tl) {
					if (ANY_HIT && blocked) cur = kIdle;
					else {
						const bool empty = sp == sp0;
						const uint32_t spm = sp - 4u;
						cur = (empty || (popped == kExitMarker && spm == sp0)) ? kDone : popped;
						sp = empty ? sp : spm;
					}
				}
			}
		}
	}
	if (ANY_HIT) { // one global atomic per workgroup (see k_trace)
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

} // namespace pol
