// kernels_wide.h -- k_trace_wide: the persistent traversal of kernels.h k_trace over the FOUR-WIDE tree of scene_layout.h
// build_wide, for the scenes whose tree is read from global memory (C4, C5, the terrain).
//
// Those scenes run at the CU's gather rate, and the gather unit charges per lane and LINE touched (profiles/r04_gather_rate.txt:
// a 16-, 32- or 64-byte record costs the same; so does a second load from the same line): what a ray pays for is the number of
// node records and triangle runs it touches.  A wide node keeps the pair node's 64 bytes -- one line, one gather per step -- but
// holds up to FOUR children, their boxes quantised to 8 bits per coordinate inside the node's own box: 0.57-0.65 x the steps.
//
// A quantised box is CONSERVATIVE (the upload rounds outward and checks the decode -- fma(q, scale, org), the instruction used
// here -- against the exact box), so a ray reaches every leaf the reference's traversal reaches, and some more.  The extra ones
// are taken out on arrival: a leaf record (and an instance record) starts with the leaf's EXACT reference box, tested with the
// reference's slab test.  That is the whole of the reference's answer: every child box lies inside its parent's (checked at
// upload), the slab test is monotone in the bounds, so a ray that passes the leaf's exact box passed every exact ancestor box
// -- except where 0 * inf turns up (a ray running exactly in the plane of a flat leaf box that is also a face of a non-flat
// ancestor): the upload marks those axes per leaf (`flat_unsafe`) and the arrival test refuses exactly the rays the ancestor
// would have refused.  tests/test_scene_layout.py walks this tree on the CPU against the oracle on random, axis-parallel and
// on-the-face rays; the GPU tests run the kernel against the oracle like every other traversal variant.
#pragma once

#include "kernels.h"

namespace pol {

struct WideDev {
	const float4 *nodes;   // 4 float4 per wide node: org.xyz | scale.x ; scale.y | scale.z | q[0..7] ; q[8..23] ; refs
	const float4 *leafrec; // per triangle leaf: exact lo.xyz | flat_unsafe bits ; exact hi.xyz | - ; then TriRec x count
	const float4 *winst;   // 6 float4 per instance: rows of the inverse matrix, meta (wide root, rank), exact lo | flat bits, exact hi
	int root_ref;
	int root_is_instance;  // single-instance scenes: the one instance is entered at ray set-up (kernels.h BvhDev)
	InstRec root_inst;     // meta.x = the mesh's wide root
};

constexpr int kWideEmpty = (int)0x80000003; // scene_layout.h kWideEmptyRef

__device__ __forceinline__ bool slab_hit3(f3 lo, f3 hi, f3 o, f3 inv, float maxDist, float &t) { // kernels.h slab_hit_hw on f3 bounds
	float t0x = (lo.x - o.x) * inv.x, t0y = (lo.y - o.y) * inv.y, t0z = (lo.z - o.z) * inv.z;
	float t1x = (hi.x - o.x) * inv.x, t1y = (hi.y - o.y) * inv.y, t1z = (hi.z - o.z) * inv.z;
	float minmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
	float maxmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
	t = maxmin;
	return !(minmax < 0) && !(maxmin > minmax) && !(maxmin >= maxDist) && maxmin < kFltMax;
}

// the exact reference box on arrival (+ the flat rule): does the reference's traversal reach this leaf / instance?
__device__ __forceinline__ bool arrives(float4 xlo, float4 xhi, f3 o, f3 inv, float maxDist) {
	float t;
	bool ok = slab_hit3(xyz(xlo), xyz(xhi), o, inv, maxDist, t);
	const uint32_t flat = (uint32_t)fbits(xlo.w);
	ok = ok && !((flat & 1u) && __builtin_isinf(inv.x) && o.x == xlo.x);
	ok = ok && !((flat & 2u) && __builtin_isinf(inv.y) && o.y == xlo.y);
	ok = ok && !((flat & 4u) && __builtin_isinf(inv.z) && o.z == xlo.z);
	return ok;
}

template <bool ANY_HIT, int STACK>
__global__ __launch_bounds__(WG) void k_trace_wide(Streams st, WideDev B, uint32_t num_chunks, float4 *acc, unsigned long long *stats) {
	constexpr int EXIT = kExitMarker;
	__shared__ int stk[STACK + 4][WG]; // row 0: the dummy below an empty stack; + 3: a step may write three entries before it knows how many it keeps
	constexpr uint32_t kRow = WG * sizeof(int);
	__shared__ uint32_t wg_cursor;
	if (threadIdx.x == 0) wg_cursor = 0;
	__syncthreads();
	char *const stk_bytes = reinterpret_cast<char *>(&stk[0][0]);
	const int tid = threadIdx.x;
	const uint32_t sp0 = (uint32_t)tid * (uint32_t)sizeof(int);
	auto write_ref = [&](uint32_t at, int ref) { *reinterpret_cast<int *>(stk_bytes + at) = ref; };
	auto read_ref = [&](uint32_t at) -> int { return *reinterpret_cast<const int *>(stk_bytes + at); };
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
	uint32_t sp = sp0;
	int cur = kIdle, cell = 0;
	uint32_t irank = 0, unocc = 0;
	float best_t = 0.0f, best_u = 0.0f, best_v = 0.0f;
	int best_tri = -1;
	uint32_t best_irank = 0, best_trank = 0;
	f3 wo = {0, 0, 0}, wd = {0, 0, 0};       // the world-space ray (kept across instances)
	f3 nee = {0, 0, 0}, acc_old = {0, 0, 0}; // any hit, exact mode: the NEE radiance and the accumulator cell, fetched at set-up

	auto pop = [&]() {
		const int popped = read_ref(sp);
		const bool empty = sp == sp0;
		const uint32_t spm = sp - kRow;
		cur = (empty || (popped == EXIT && spm == sp0)) ? kDone : popped;
		sp = empty ? sp : spm;
	};
	auto enter = [&](float4 r0, float4 r1, float4 r2, int4 meta) { // intersect.cl:239-252; mul4x1 / mul3x1, util/transform.cl:9-26
		const f3 no = {r0.x * o.x + r0.y * o.y + r0.z * o.z + r0.w, r1.x * o.x + r1.y * o.y + r1.z * o.z + r1.w, r2.x * o.x + r2.y * o.y + r2.z * o.z + r2.w};
		const f3 nd = {r0.x * d.x + r0.y * d.y + r0.z * d.z, r1.x * d.x + r1.y * d.y + r1.z * d.z, r2.x * d.x + r2.y * d.y + r2.z * d.z};
		o = no; d = nd;
		irank = (uint32_t)meta.y;
		write_ref(sp + kRow, EXIT);
		sp += kRow;
		cur = meta.x;
	};
	auto start_ray = [&](uint32_t ray_slot, float4 o4, float4 d4) {
		slot = ray_slot;
		o = xyz(o4); d = xyz(d4);
		wo = o; wd = d;
		maxDist = o4.w;
		cell = fbits(d4.w);
		if (ANY_HIT && acc) { // (exact mode only; batched mode writes the ray's visibility byte at the end: kernels.h nee_result)
			const float4 e4 = st.occ_e[slot], a4 = acc[cell];
			nee = xyz(e4); acc_old = xyz(a4);
		}
		sp = sp0;
		cur = B.root_ref;
		irank = 0;
		if (B.root_is_instance) enter(B.root_inst.r0, B.root_inst.r1, B.root_inst.r2, B.root_inst.meta); // (popping its marker ends the ray without a restore)
		inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)}; // native_recip(ray.dir), intersect.cl:302
		best_t = maxDist; best_tri = -1; best_u = best_v = 0.0f; best_irank = best_trank = 0;
	};
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
		{
			const unsigned long long freem = __ballot(cur == kIdle);
			if (!drained && (freem == ~0ull || __popcll(freem) >= (ANY_HIT ? kRefillMinAny : kRefillMin)))
				draw([&]() { return cur == kIdle; }, [&](uint32_t ray_slot) { start_ray(ray_slot, src_o[ray_slot], src_d[ray_slot]); });
			if (__ballot(cur != kIdle) == 0ull) {
				if (drained) break;
				continue;
			}
		}
		// ---- phase 1: wide inner nodes -------------------------------------------------------------------------------------
		if (__ballot(cur >= 0) != 0ull) do {
			if (cur >= 0) {
				const float4 *N = B.nodes + 4 * (size_t)cur;
				const float4 n0 = N[0], n1 = N[1], n2 = N[2], n3 = N[3];
				const int popped = read_ref(sp); // what a pop would deliver: read beside the node
				const bool empty = sp == sp0;
				const uint32_t spm = sp - kRow;
				const f3 org = xyz(n0), scl = {n0.w, n1.x, n1.y};
				const uint32_t D[6] = {(uint32_t)fbits(n1.z), (uint32_t)fbits(n1.w), (uint32_t)fbits(n2.x), (uint32_t)fbits(n2.y), (uint32_t)fbits(n2.z), (uint32_t)fbits(n2.w)};
				const int refs[4] = {fbits(n3.x), fbits(n3.y), fbits(n3.z), fbits(n3.w)};
				float t[4];
				int r[4];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					auto qf = [&](int c) { const int j = 6 * k + c; return (float)((D[j >> 2] >> (8 * (j & 3))) & 255u); }; // v_cvt_f32_ubyteN
					const f3 lo = {__builtin_fmaf(qf(0), scl.x, org.x), __builtin_fmaf(qf(1), scl.y, org.y), __builtin_fmaf(qf(2), scl.z, org.z)};
					const f3 hi = {__builtin_fmaf(qf(3), scl.x, org.x), __builtin_fmaf(qf(4), scl.y, org.y), __builtin_fmaf(qf(5), scl.z, org.z)};
					float tk;
					bool h = slab_hit3(lo, hi, o, inv, maxDist, tk) && refs[k] != kWideEmpty;
					h = h && (ANY_HIT || !(tk > best_t * kCullMargin)); // (every box of a scene with a wide tree bounds its subtree)
					t[k] = h ? tk : kFltMax;
					r[k] = refs[k];
				}
				// nearest first: a five-comparator network over (distance, reference); children that are not hit sink to the end
				auto cswap = [&](int a, int b) {
					const bool sw = t[b] < t[a];
					const float ta = sw ? t[b] : t[a], tb = sw ? t[a] : t[b];
					const int ra = sw ? r[b] : r[a], rb = sw ? r[a] : r[b];
					t[a] = ta; t[b] = tb; r[a] = ra; r[b] = rb;
				};
				cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
				const bool h0 = t[0] < kFltMax, h1 = t[1] < kFltMax, h2 = t[2] < kFltMax, h3 = t[3] < kFltMax;
				// pending children go on the stack farthest first: three unconditional stores at the places they would take, the
				// stack pointer advances by the number really pending (the rows above it are free)
				const uint32_t n_push = (h1 ? 1u : 0u) + (h2 ? 1u : 0u) + (h3 ? 1u : 0u);
				write_ref(sp + kRow, h3 ? r[3] : (h2 ? r[2] : r[1]));
				write_ref(sp + 2 * kRow, h3 ? r[2] : r[1]);
				write_ref(sp + 3 * kRow, r[1]);
				const int after_pop = (empty || (popped == EXIT && spm == sp0)) ? kDone : popped;
				cur = h0 ? r[0] : after_pop;
				sp = h0 ? sp + n_push * kRow : (empty ? sp : spm);
			}
		} while (__popcll(__ballot(cur >= 0)) >= (ANY_HIT ? kStragglersAny : kStragglers));
		// ---- phase 2: everything that is not an inner node -----------------------------------------------------------------
		if (cur == kDone) {
			if (ANY_HIT) {
				if (acc) {
					float *c = reinterpret_cast<float *>(acc + cell);
					c[0] = acc_old.x + nee.x; c[1] = acc_old.y + nee.y; c[2] = acc_old.z + nee.z;
				} else {
					st.vis[slot] = 1;
				}
				unocc++;
			} else {
				st.hit[slot] = make_float4(best_u, best_v, best_t, ibits(best_tri));
			}
			cur = kIdle;
		}
		if (cur == EXIT) { // leaving the instance: back to the world-space ray (intersect.cl:330-335)
			o = wo; d = wd;
			inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
			pop();
		}
		if (cur < 0 && cur >= kFirstLeafRef && (((uint32_t)~cur) & 15u) == 0u) { // a top-level leaf: an instance
			const float4 *I = B.winst + 6 * (size_t)(((uint32_t)~cur) >> 4);
			const float4 r0 = I[0], r1 = I[1], r2 = I[2], m4 = I[3], xlo = I[4], xhi = I[5];
			if (arrives(xlo, xhi, o, inv, maxDist)) {
				enter(r0, r1, r2, make_int4(fbits(m4.x), fbits(m4.y), 0, 0));
				inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
			} else {
				pop(); // the reference's traversal does not get here
			}
		}
		// ---- triangle leaves: the exact box first, then Moeller-Trumbore, intersect.cl:255-292, without early exits -------------
		{
			const bool tl = cur < 0 && cur >= kFirstLeafRef && (((uint32_t)~cur) & 15u) != 0u;
			if (__ballot(tl) != 0ull) {
				const uint32_t code = (uint32_t)~cur;
				const float4 *rec = B.leafrec + (tl ? (size_t)(code >> 4) : 0);
				const int popped = read_ref(sp);
				uint32_t ntri = 0;
				if (tl) {
					const float4 xlo = rec[0], xhi = rec[1];
					ntri = arrives(xlo, xhi, o, inv, maxDist) ? (code & 15u) : 0u;
				}
				bool occluded = false;
				uint32_t i = 0;
				auto test_tri = [&](float4 v0, float4 e1w, float4 e2w) {
					const f3 e1 = xyz(e1w), e2 = xyz(e2w);
					const f3 pv = cross(d, e2);
					const float det = dot(e1, pv);
					bool ok = !(pm_fabs(det) < kEps);
					const float idet = rcp_det(det);
					const f3 tv = o - xyz(v0);
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
						const uint32_t trank = (uint32_t)fbits(v0.w);
						const bool closer = tt < best_t;
						const bool tie = tt == best_t && best_tri >= 0 && (irank < best_irank || (irank == best_irank && trank < best_trank));
						const bool take = ok && (closer || tie);
						best_t = take ? tt : best_t; best_u = take ? u : best_u; best_v = take ? v : best_v;
						best_tri = take ? fbits(e1w.w) : best_tri;
						best_irank = take ? irank : best_irank; best_trank = take ? trank : best_trank;
					}
				};
				while (__ballot(i < ntri && !occluded) != 0ull) {
					if (i < ntri && !occluded) {
						const float4 *T = rec + 2 + 3 * i;
						test_tri(T[0], T[1], T[2]);
					}
					i++;
				}
				if (tl) {
					if (ANY_HIT && occluded) { cur = kIdle; if (!acc) st.vis[slot] = 0; }
					else {
						const bool empty = sp == sp0;
						const uint32_t spm = sp - kRow;
						cur = (empty || (popped == EXIT && spm == sp0)) ? kDone : popped;
						sp = empty ? sp : spm;
					}
				}
			}
		}
	}
	if (ANY_HIT) {
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
