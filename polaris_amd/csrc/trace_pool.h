// trace_pool.h -- BVH traversal with a per-wave ray POOL in LDS ("k_trace_pool").
//
// What bounds k_trace (kernels.h) is divergence: rays of a wave sit in different phases (box tests
// vs triangle tests), so only ~43 % of the lanes of an inner-node iteration and ~35 % of a triangle
// iteration do useful work -- and every vector-memory instruction (4 gathers per node, 3 per
// triangle) is paid per wave, however few lanes it serves.  Here the binding of rays to lanes is
// given up:
//
//   * a wave owns a pool of R rays (R > 64) whose whole traversal state lives in LDS: origin | best t,
//     1/direction | rank of the best hit's instance, direction | accumulator cell, (u, v, triangle,
//     rank) of the best hit, (current node, stack depth, instance rank, stream slot), and the node
//     stack -- 80 + 4 x STACK bytes per ray;
//   * every ray is in exactly one of three wave-private ring queues of ray ids: INNER (next step is a
//     pair-of-boxes test), LEAF (next step is a triangle test / entering or leaving an instance) or
//     FREE;
//   * one iteration of the wave takes up to 64 ids from the INNER queue AND up to 64 from the LEAF
//     queue, advances each of those rays by exactly ONE step -- all lanes of a batch execute the same
//     code: no phase divergence, ~full waves per memory instruction -- and files every ray under its
//     next phase; finished rays write their result and their pool slot is refilled from the chunk
//     queue.  The two batches are independent (a ray is in one queue only), so their loads are in
//     flight together: memory-level parallelism inside the wave instead of across many resident
//     waves.
//
// Per-ray arithmetic (slab test, Moeller-Trumbore, near-child order, cull by the running best with
// the same margin, tie rule by DFS rank) is the same as traverse<> / k_trace in kernels.h, and a
// ray's traversal does not depend on any other ray, so results are bit-identical under this
// schedule too (tests/test_gpu_parity.py runs every traversal variant against the oracle).
// Everything is wave-private: no barrier and no atomic inside the loop; DS instructions of one wave
// execute in order, which is what makes the queue hand-over between lanes safe.
#pragma once

#include "kernels.h"

namespace pol {

constexpr int kDoneMarker = (int)0x80000001; // "ray finished" as a node reference (leaf codes never use the 16 lowest negative values)

template <bool ANY_HIT, int STACK, int R>
struct PoolLds {
	float4 A[R];                 // origin.xyz | best t (closest hit) or max distance (any hit)
	float4 Bv[R];                // 1/direction | instance rank of the best hit (uint bits)
	float4 C[R];                 // direction.xyz | accumulator cell (any hit, bits) or the ray's max distance (closest hit)
	float4 D[ANY_HIT ? 1 : R];   // u, v, triangle (int bits), triangle rank (uint bits) of the best hit
	int4 M[R];                   // current node reference, stack depth, rank of the instance the ray is inside, stream slot
	int stk[STACK][R];
	uint8_t q_in[R], q_lf[R], q_free[R];
};

#ifndef POLARIS_POOL_REFILL
#define POLARIS_POOL_REFILL 32
#endif

template <bool ANY_HIT, int STACK, int R>
__global__ __launch_bounds__(WG) void k_trace_pool(Streams st, BvhDev B, uint32_t num_chunks, float4 *acc, unsigned long long *stats) {
	static_assert(R >= 64 && R <= 256, "pool ids are bytes");
	__shared__ PoolLds<ANY_HIT, STACK, R> pools[WG / 64];
	__shared__ uint32_t wg_cursor;
	if (threadIdx.x == 0) wg_cursor = 0;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	PoolLds<ANY_HIT, STACK, R> &P = pools[wave];
	for (uint32_t i = lane; i < (uint32_t)R; i += 64) P.q_free[i] = (uint8_t)i;
	__syncthreads();
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t *cnts = ANY_HIT ? st.cnt_occ : st.cnt_ray;
	const float4 *src_o = ANY_HIT ? st.occ_o : st.ray_o;
	const float4 *src_d = ANY_HIT ? st.occ_d : st.ray_d;

	// wave-uniform queue state
	uint32_t in_head = 0, in_cnt = 0, lf_head = 0, lf_cnt = 0, fr_head = 0, fr_cnt = R;
	uint32_t chunk = 0, off = 0, cnt = 0;
	bool drained = false;
	uint32_t unocc = 0;

	auto push = [&](uint8_t *q, uint32_t head, uint32_t &n, bool pred, uint32_t id) {
		const unsigned long long m = __ballot(pred);
		if (pred) q[(head + n + (uint32_t)__popcll(m & below)) % (uint32_t)R] = (uint8_t)id;
		n += (uint32_t)__popcll(m);
	};
	// next pending node of ray `id`, or kDoneMarker.  An instance's exit marker with nothing pending above it ends the ray too
	// (no need to restore the world-space ray first).
	auto pop = [&](uint32_t id, int &cur, int &sp) {
		if (sp == 0) { cur = kDoneMarker; return; }
		cur = P.stk[--sp][id];
		if (cur == kExitMarker && sp == 0) cur = kDoneMarker;
	};
	auto finish = [&](uint32_t id, uint32_t slot, bool occluded) {
		if (ANY_HIT) {
			if (!occluded) {
				const float4 e = st.occ_e[slot];
				const uint32_t cell = (uint32_t)fbits(P.C[id].w);
				float4 a = acc[cell]; // one path per cell and launch: plain read-modify-write
				a.x += e.x; a.y += e.y; a.z += e.z;
				acc[cell] = a;
				unocc++;
			}
		} else {
			const float4 d = P.D[id];
			st.hit[slot] = make_float4(d.x, d.y, P.A[id].w, d.z);
		}
	};

#ifdef POLARIS_TRACE_COUNTERS
	uint32_t c_node = 0, c_leaf = 0, c_iter1 = 0, c_iter2 = 0, c_outer = 0, c_refill = 0; // wave-uniform
#define TCP(x) x
#else
#define TCP(x)
#endif
#ifdef POLARIS_STAMPS
	unsigned long long sk_refill = 0, sk_pop = 0, sk_state = 0, sk_inner = 0, sk_leaf = 0, sk_push = 0, sk_total = 0, sk_t = 0, sk_t0 = __builtin_amdgcn_s_memtime();
#endif
	for (;;) {
		TCP(c_outer++;)
		STAMP_BEGIN();
		// ---- refill free pool slots from the chunk queue -------------------------------------------
		if (!drained && (fr_cnt >= (uint32_t)POLARIS_POOL_REFILL || in_cnt + lf_cnt == 0)) {
			TCP(c_refill++;)
			while (fr_cnt > 0) {
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
				const uint32_t take = min(min(cnt - off, fr_cnt), 64u);
				const bool mine = lane < take;
				uint32_t id = 0;
				if (mine) {
					id = P.q_free[(fr_head + lane) % (uint32_t)R];
					const uint32_t slot = chunk * WG + off + lane;
					const float4 o4 = src_o[slot], d4 = src_d[slot];
					P.A[id] = o4; // .w = max distance = the initial best t
					P.Bv[id] = make_float4(pm_rcp(d4.x), pm_rcp(d4.y), pm_rcp(d4.z), 0.0f); // native_recip(ray.dir), intersect.cl:302
					P.C[id] = ANY_HIT ? d4 : make_float4(d4.x, d4.y, d4.z, o4.w);
					if (!ANY_HIT) P.D[id] = make_float4(0.0f, 0.0f, ibits(-1), 0.0f);
					P.M[id] = make_int4(B.root_ref, 0, 0, (int)slot);
				}
				if (B.root_ref >= 0) push(P.q_in, in_head, in_cnt, mine, id);
				else push(P.q_lf, lf_head, lf_cnt, mine, id);
				fr_head = (fr_head + take) % (uint32_t)R;
				fr_cnt -= take;
				off += take;
			}
		}
		STAMP(sk_refill)
		if (in_cnt + lf_cnt == 0) {
			if (drained) break;
			continue;
		}
		// ---- take one batch from each queue ------------------------------------------------------------
		const uint32_t n_i = min(in_cnt, 64u), n_l = min(lf_cnt, 64u);
		const bool act_i = lane < n_i, act_l = lane < n_l;
		TCP(c_node += n_i; c_leaf += n_l; c_iter1 += n_i ? 1 : 0; c_iter2 += n_l ? 1 : 0;)
		uint32_t id_i = 0, id_l = 0;
		if (act_i) id_i = P.q_in[(in_head + lane) % (uint32_t)R];
		if (act_l) id_l = P.q_lf[(lf_head + lane) % (uint32_t)R];
		in_head = (in_head + n_i) % (uint32_t)R; in_cnt -= n_i;
		lf_head = (lf_head + n_l) % (uint32_t)R; lf_cnt -= n_l;

#ifdef POLARIS_STAMPS
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(id_i), "+v"(id_l) :: "memory");
		STAMP(sk_pop)
#endif
		// ---- state of both batches from LDS, then the loads of both: they are in flight together ---------
		// (inactive lanes read pool slot 0 / node 0 / triangle 0: always valid addresses, results unused)
		const float4 a_i = P.A[id_i], b_i = P.Bv[id_i];
		const int4 m_i = P.M[id_i];
		const float max_dist_i = ANY_HIT ? a_i.w : P.C[id_i].w; // the slab test compares with the ray's MAX distance (intersect.cl:309)
		const int4 m_l = P.M[id_l];
		const float4 a_l = P.A[id_l], c_l = P.C[id_l];
		int cur_i = act_i ? m_i.x : 0, sp_i = m_i.y;
		int cur_l = act_l ? m_l.x : kExitMarker, sp_l = m_l.y;
		const uint32_t code_l = (uint32_t)~cur_l;
		const bool inline_leaf = act_l && cur_l != kExitMarker && (code_l & 15u) != 0u; // 1..15 triangles, the common case
		const PairNode N = B.pairs[cur_i];
		const TriRec T0 = B.tris[inline_leaf ? (code_l >> 4) : 0u];

#ifdef POLARIS_STAMPS
		{
			PairNode &Nn = const_cast<PairNode &>(N);
			TriRec &Tt = const_cast<TriRec &>(T0);
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(Nn.lo0.x), "+v"(Nn.hi0.x), "+v"(Nn.lo1.x), "+v"(Nn.hi1.x), "+v"(Tt.v0.x), "+v"(Tt.e1.x), "+v"(Tt.e2.x) :: "memory");
			STAMP(sk_state)
		}
#endif
		// ---- INNER batch: one pair-of-boxes step (intersect.cl:296-328) -------------------------------
		if (act_i) {
			const f3 o = xyz(a_i), inv = xyz(b_i);
			float t0 = slab_entry_hw(N.lo0, N.hi0, o, inv, max_dist_i);
			float t1 = slab_entry_hw(N.lo1, N.hi1, o, inv, max_dist_i);
			if (!ANY_HIT) { // cull subtrees that start beyond the best hit (+inf factor = box does not bound its subtree)
				if (t0 > a_i.w * N.hi0.w) t0 = kFltMax;
				if (t1 > a_i.w * N.hi1.w) t1 = kFltMax;
			}
			int c0 = fbits(N.lo0.w), c1 = fbits(N.lo1.w);
			const bool h0 = t0 < kFltMax, h1 = t1 < kFltMax;
			if (h0 && h1) {
				if (!ANY_HIT && t1 < t0) { const int t = c0; c0 = c1; c1 = t; }
				P.stk[sp_i++][id_i] = c1;
				cur_i = c0;
			} else if (h0 || h1) {
				cur_i = h0 ? c0 : c1;
			} else {
				pop(id_i, cur_i, sp_i);
			}
			if (cur_i == kDoneMarker) finish(id_i, (uint32_t)m_i.w, false);
			else { P.M[id_i].x = cur_i; P.M[id_i].y = sp_i; }
		} else cur_i = kDoneMarker;

		STAMP(sk_inner)
		// ---- LEAF batch: one triangle (or entering / leaving an instance) --------------------------------
		if (act_l) {
			const uint32_t slot = (uint32_t)m_l.w;
			bool occluded = false;
			// One Moeller-Trumbore test (intersect.cl:255-292) of the ray (o, d) against T; updates the pool on a closer hit.
			float best_t = a_l.w;
			auto test_tri = [&](const TriRec &T, f3 o, f3 d, uint32_t irank) {
				const f3 e1 = xyz(T.e1), e2 = xyz(T.e2);
				const f3 pv = cross(d, e2);
				const float det = dot(e1, pv);
				if (pm_fabs(det) < kEps) return;
				const float idet = pm_rcp(det);
				const f3 tv = o - xyz(T.v0);
				const float u = dot(tv, pv) * idet;
				if (u < 0.0f || u > 1.0f) return;
				const f3 qv = cross(tv, e1);
				const float v = dot(d, qv) * idet;
				if (v < 0.0f || u + v > 1.0f) return;
				const float tt = dot(e2, qv) * idet;
				if (ANY_HIT) {
					if (tt > kEps && tt < best_t) occluded = true; // best_t is the ray's max distance here
				} else if (tt > kEps) {
					const uint32_t trank = (uint32_t)fbits(T.v0.w);
					bool take = tt < best_t;
					if (tt == best_t) { // exact tie: the reference keeps the first one tested (intersect.cl:281 strict <)
						const float4 bd = P.D[id_l];
						const uint32_t birank = (uint32_t)fbits(P.Bv[id_l].w), btrank = (uint32_t)fbits(bd.w);
						take = fbits(bd.z) >= 0 && (irank < birank || (irank == birank && trank < btrank));
					}
					if (take) {
						best_t = tt;
						P.A[id_l].w = tt;
						P.Bv[id_l].w = ibits((int)irank);
						P.D[id_l] = make_float4(u, v, T.e1.w, ibits((int)trank));
					}
				}
			};
			if (inline_leaf) {
				// ONE triangle per visit, the rest of the leaf is re-filed as a shorter leaf -- every lane of the batch runs
				// exactly one Moeller-Trumbore test
				test_tri(T0, xyz(a_l), xyz(c_l), (uint32_t)m_l.z);
				const uint32_t first = code_l >> 4, ntri = code_l & 15u;
				if (ntri > 1 && !occluded) cur_l = ~(int)(((first + 1u) << 4) | (ntri - 1u));
				else pop(id_l, cur_l, sp_l);
			} else if (cur_l == kExitMarker) { // leaving the instance: back to the world-space ray (intersect.cl:330-335)
				const float4 o4 = src_o[slot], d4 = src_d[slot];
				P.A[id_l] = make_float4(o4.x, o4.y, o4.z, a_l.w);
				float4 b = P.Bv[id_l];
				b.x = pm_rcp(d4.x); b.y = pm_rcp(d4.y); b.z = pm_rcp(d4.z);
				P.Bv[id_l] = b;
				P.C[id_l] = make_float4(d4.x, d4.y, d4.z, c_l.w);
				pop(id_l, cur_l, sp_l);
			} else {
				const int2 li = B.leaves[code_l >> 4];
				if (li.y == 0) { // top-level leaf: enter the mesh instance (intersect.cl:239-252)
					const InstRec I = B.insts[-li.x];
					const f3 o = xyz(a_l), d = xyz(c_l);
					// mul4x1 / mul3x1, util/transform.cl:9-26
					const f3 no = {I.r0.x * o.x + I.r0.y * o.y + I.r0.z * o.z + I.r0.w, I.r1.x * o.x + I.r1.y * o.y + I.r1.z * o.z + I.r1.w,
					               I.r2.x * o.x + I.r2.y * o.y + I.r2.z * o.z + I.r2.w};
					const f3 nd = {I.r0.x * d.x + I.r0.y * d.y + I.r0.z * d.z, I.r1.x * d.x + I.r1.y * d.y + I.r1.z * d.z,
					               I.r2.x * d.x + I.r2.y * d.y + I.r2.z * d.z};
					P.A[id_l] = make_float4(no.x, no.y, no.z, a_l.w);
					P.C[id_l] = make_float4(nd.x, nd.y, nd.z, c_l.w);
					float4 b = P.Bv[id_l];
					b.x = pm_rcp(nd.x); b.y = pm_rcp(nd.y); b.z = pm_rcp(nd.z);
					P.Bv[id_l] = b;
					P.M[id_l].z = I.meta.y;
					P.stk[sp_l++][id_l] = kExitMarker;
					cur_l = I.meta.x;
				} else { // a leaf of more than 15 triangles: walked in place
					const f3 o = xyz(a_l), d = xyz(c_l);
					for (int t = -li.x; t < -li.x + li.y && !occluded; t++) test_tri(B.tris[t], o, d, (uint32_t)m_l.z);
					pop(id_l, cur_l, sp_l);
				}
			}
			if (occluded) cur_l = kDoneMarker;
			if (cur_l == kDoneMarker) finish(id_l, slot, occluded);
			else { P.M[id_l].x = cur_l; P.M[id_l].y = sp_l; }
		} else cur_l = kDoneMarker;

		STAMP(sk_leaf)
		// ---- file every ray of the two batches under its next phase ------------------------------------
		push(P.q_in, in_head, in_cnt, act_i && cur_i >= 0, id_i);
		push(P.q_in, in_head, in_cnt, act_l && cur_l >= 0, id_l);
		push(P.q_lf, lf_head, lf_cnt, act_i && cur_i < 0 && cur_i != kDoneMarker, id_i);
		push(P.q_lf, lf_head, lf_cnt, act_l && cur_l < 0 && cur_l != kDoneMarker, id_l);
		push(P.q_free, fr_head, fr_cnt, act_i && cur_i == kDoneMarker, id_i);
		push(P.q_free, fr_head, fr_cnt, act_l && cur_l == kDoneMarker, id_l);
		STAMP(sk_push)
	}
#ifdef POLARIS_STAMPS
	if (lane == 0) {
		sk_total = __builtin_amdgcn_s_memtime() - sk_t0;
		const unsigned long long v[8] = {sk_refill, sk_pop, sk_state, sk_inner, sk_leaf, sk_push, sk_total, 1ull};
		for (int i = 0; i < 8; i++) atomicAdd(&stats[ST_DEBUG + (ANY_HIT ? 8 : 0) + i], v[i]);
	}
#endif
#ifdef POLARIS_TRACE_COUNTERS
	if (lane == 0) {
		const uint32_t v[8] = {c_node, c_leaf, 0, c_iter1, c_iter2, c_outer, c_refill, 0};
		for (int i = 0; i < 8; i++)
			if (v[i]) atomicAdd(&stats[ST_DEBUG + (ANY_HIT ? 8 : 0) + i], (unsigned long long)v[i]);
	}
#endif
	if (ANY_HIT) { // wave sum -> workgroup sum in LDS -> ONE global atomic per workgroup
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
