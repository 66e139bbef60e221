// layout_check.cpp -- CPU harness over polaris_amd/csrc/scene_layout.h (test tool, not product).
//
// Builds the traversal layout the HIP backend uploads and walks it on the CPU with the same
// rules as polaris_amd/csrc/kernels.h traverse<false>(): same slab test, same Moeller-Trumbore
// order of operations, near child first, conservative pruning, DFS-rank tie break.  Used by
// tests/test_scene_layout.py to show that subdividing leaves at upload (max_leaf_tris) never
// changes a hit record, and to count box/triangle tests per ray.
//
//   g++ -O2 -ffp-contract=off -shared -fPIC -Iinclude -Ipolaris_amd/csrc layout_check.cpp
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "polaris_math.h"
#include "scene_layout.h"

using namespace pol;

namespace {
constexpr float kFltMax = 3.402823466e+38f;
constexpr float kEps = 0.00001f; // == shading.h kEps (constants.cl:23)
constexpr int kExit = (int)0x80000000;

struct V3 { float x, y, z; };
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

inline float slab(const float *lo, const float *hi, V3 o, V3 inv, float maxDist) {
	float t0x = (lo[0] - o.x) * inv.x, t0y = (lo[1] - o.y) * inv.y, t0z = (lo[2] - o.z) * inv.z;
	float t1x = (hi[0] - o.x) * inv.x, t1y = (hi[1] - o.y) * inv.y, t1z = (hi[2] - o.z) * inv.z;
	float minmax = pm_fmin(pm_fmin(pm_fmax(t0x, t1x), pm_fmax(t0y, t1y)), pm_fmax(t0z, t1z));
	float maxmin = pm_fmax(pm_fmax(pm_fmin(t0x, t1x), pm_fmin(t0y, t1y)), pm_fmin(t0z, t1z));
	return (minmax < 0 || maxmin > minmax) ? kFltMax : (maxmin >= maxDist ? kFltMax : maxmin);
}
} // namespace

extern "C" {

// rays: [n][8] = origin.xyz, maxDist, dir.xyz, unused.  hit: [n][6] = tri, inst, bits(t), bits(u), bits(v), found.
// counters: [0] pair steps, [1] triangle tests, [2] leaf visits, [3] pair records, [4] triangle slots, [5] stack need,
//           [6] inner nodes holding a box that does not bound its subtree.
// any_hit != 0: first-found semantics is order dependent, so only `found` is meaningful.
int layout_check_traverse(const PolarisSceneView *sc, int max_leaf_tris, const float *rays, uint32_t n, int any_hit,
                          int32_t *hit, uint64_t *counters, char *err, size_t err_len) {
	SceneLayout L;
	std::string e = build_layout(*sc, L, max_leaf_tris);
	if (e == "@retry-without-subdivision") { L = SceneLayout(); e = build_layout(*sc, L, 0); }
	if (!e.empty()) {
		if (err && err_len) { strncpy(err, e.c_str(), err_len - 1); err[err_len - 1] = 0; }
		return 1;
	}
	uint64_t steps = 0, tests = 0, visits = 0;
	std::vector<int> stk(4 * kTraversalStack + 8);
	for (uint32_t r = 0; r < n; r++) {
		const float *R = rays + 8 * (size_t)r;
		const V3 O = {R[0], R[1], R[2]}, D = {R[4], R[5], R[6]};
		const float maxDist = R[3];
		V3 o = O, d = D, inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
		int sp = 0, cur = L.root_ref, inst = 0, btri = -1, binst = 0;
		uint32_t irank = 0, birank = 0, btrank = 0;
		float bt = maxDist, bu = 0, bv = 0;
		bool found = false;
		for (;;) {
			if (cur >= 0) {
				const PairNodeH &P = L.pairs[cur];
				steps++;
				float t0 = slab(P.lo0, P.hi0, o, inv, maxDist), t1 = slab(P.lo1, P.hi1, o, inv, maxDist);
				if (!any_hit) {
					float f0, f1;
					memcpy(&f0, &P.pad0, 4); memcpy(&f1, &P.pad1, 4);
					if (t0 > bt * f0) t0 = kFltMax;
					if (t1 > bt * f1) t1 = kFltMax;
				}
				int c0 = P.ref0, c1 = P.ref1;
				const bool h0 = t0 < kFltMax, h1 = t1 < kFltMax;
				if (h0 && h1) {
					if (t1 < t0) { int t = c0; c0 = c1; c1 = t; }
					stk[sp++] = c1;
					cur = c0;
					continue;
				}
				if (h0 || h1) { cur = h0 ? c0 : c1; continue; }
			} else {
				const uint32_t code = (uint32_t)~cur; // kernels.h leaf_of()
				const LeafInfoH li = (code & 15u) ? LeafInfoH{-(int32_t)(code >> 4), (int32_t)(code & 15u)}
				                   : (!(code & kBigLeafFlag) ? LeafInfoH{-(int32_t)(code >> 4), 0} : L.leaves[(code & (kBigLeafFlag - 1u)) >> 4]);
				if (li.rdata == 0) {
					inst = -li.ldata;
					const InstH &I = L.insts[inst];
					irank = I.rank;
					stk[sp++] = kExit;
					V3 no = {I.r0[0] * o.x + I.r0[1] * o.y + I.r0[2] * o.z + I.r0[3], I.r1[0] * o.x + I.r1[1] * o.y + I.r1[2] * o.z + I.r1[3],
					         I.r2[0] * o.x + I.r2[1] * o.y + I.r2[2] * o.z + I.r2[3]};
					V3 nd = {I.r0[0] * d.x + I.r0[1] * d.y + I.r0[2] * d.z, I.r1[0] * d.x + I.r1[1] * d.y + I.r1[2] * d.z,
					         I.r2[0] * d.x + I.r2[1] * d.y + I.r2[2] * d.z};
					o = no; d = nd;
					inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
					cur = I.root_ref;
					continue;
				}
				visits++;
				const int first = -li.ldata;
				for (int t = first; t < first + li.rdata && !found; t++) {
					const TriH &T = L.tris[t];
					tests++;
					V3 e1 = {T.e1[0], T.e1[1], T.e1[2]}, e2 = {T.e2[0], T.e2[1], T.e2[2]};
					V3 pv = cross(d, e2);
					float det = dot(e1, pv);
					if (pm_fabs(det) < kEps) continue;
					float idet = pm_rcp(det);
					V3 tv = {o.x - T.v0[0], o.y - T.v0[1], o.z - T.v0[2]};
					float u = dot(tv, pv) * idet;
					if (u < 0.0f || u > 1.0f) continue;
					V3 qv = cross(tv, e1);
					float v = dot(d, qv) * idet;
					if (v < 0.0f || u + v > 1.0f) continue;
					float tt = dot(e2, qv) * idet;
					if (any_hit) {
						if (tt > kEps && tt < maxDist) found = true;
					} else if (tt > kEps) {
						const bool closer = tt < bt;
						const bool tie = tt == bt && btri >= 0 && (irank < birank || (irank == birank && T.rank < btrank));
						if (closer || tie) { bt = tt; bu = u; bv = v; btri = (int)(T.orig & ((1u << L.tri_bits) - 1u)); /* the shading class rides in the bits above (scene_layout.h) */ binst = inst; birank = irank; btrank = T.rank; }
					}
				}
				if (found) break;
			}
			bool done = false;
			for (;;) {
				if (sp == 0) { done = true; break; }
				cur = stk[--sp];
				if (cur != kExit) break;
				o = O; d = D;
				inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
			}
			if (done) break;
		}
		int32_t *H = hit + 6 * (size_t)r;
		if (any_hit) { H[0] = H[1] = H[2] = H[3] = H[4] = 0; H[5] = found; }
		else {
			H[0] = btri; H[1] = binst;
			memcpy(&H[2], &bt, 4); memcpy(&H[3], &bu, 4); memcpy(&H[4], &bv, 4);
			H[5] = btri >= 0;
		}
	}
	if (counters) {
		counters[0] = steps; counters[1] = tests; counters[2] = visits;
		counters[3] = L.pairs.size(); counters[4] = L.tris.size(); counters[5] = (uint64_t)L.max_stack; counters[6] = L.unbounded_boxes;
	}
	return 0;
}

// The same queries over the FOUR-WIDE tree (scene_layout.h build_wide) with the rules of kernels.h k_trace_wide: child boxes
// decoded as fma(q, scale, org), the slab test on the decoded (conservative) boxes, hit children visited nearest first (stored
// order for any-hit), the EXACT reference box -- and the flat rule -- tested on arrival at a triangle leaf or an instance.
// counters: [0] wide steps, [1] triangle tests, [2] leaves reached, [3] leaves rejected by the exact test, [4] wide nodes,
//           [5] stack need (0 = the scene has no wide tree: err says why), [6] deepest stack seen, [7] nodes left as pairs to stay
//           within the stack limit (layout_check_wide_stack_limit; default: the traversal stack's 32 entries)
static int g_wide_stack_limit = kTraversalStack;
void layout_check_wide_stack_limit(int entries) { g_wide_stack_limit = entries; }
int layout_check_traverse_wide(const PolarisSceneView *sc, int max_leaf_tris, const float *rays, uint32_t n, int any_hit,
                               int32_t *hit, uint64_t *counters, char *err, size_t err_len) {
	SceneLayout L;
	std::string e = build_layout(*sc, L, max_leaf_tris);
	if (e == "@retry-without-subdivision") { L = SceneLayout(); e = build_layout(*sc, L, 0); }
	if (e.empty()) { build_wide(L, g_wide_stack_limit); if (L.wide_stack == 0) e = "no wide tree: " + L.wide_why; }
	if (!e.empty()) {
		if (err && err_len) { strncpy(err, e.c_str(), err_len - 1); err[err_len - 1] = 0; }
		return 1;
	}
	uint64_t steps = 0, tests = 0, reached = 0, rejected = 0, deepest = 0;
	std::vector<int> stk(kTraversalStack + 8);
	auto exact_ok = [&](const float *xlo, const float *xhi, uint32_t flat, V3 o, V3 inv, float maxDist) {
		if (!(slab(xlo, xhi, o, inv, maxDist) < kFltMax)) return false;
		// the flat rule: the leaf's box is flat on this axis, lies in a face of a non-flat ancestor, and the ray runs in that plane
		const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
		for (int a = 0; a < 3; a++)
			if ((flat >> a & 1u) && std::isinf(ii[a]) && oo[a] == xlo[a]) return false;
		return true;
	};
	for (uint32_t r = 0; r < n; r++) {
		const float *R = rays + 8 * (size_t)r;
		const V3 O = {R[0], R[1], R[2]}, D = {R[4], R[5], R[6]};
		const float maxDist = R[3];
		V3 o = O, d = D, inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
		int sp = 0, cur = L.wide_root, inst = 0, btri = -1, binst = 0;
		uint32_t irank = 0, birank = 0, btrank = 0;
		float bt = maxDist, bu = 0, bv = 0;
		bool found = false, at_root = true;
		for (;;) {
			if (cur >= 0) {
				const SceneLayout::WideNodeH &Wn = L.wide[cur];
				at_root = false;
				steps++;
				const float scale[3] = {Wn.scale_x, Wn.scale_y, Wn.scale_z};
				float t[4];
				int idx[4], nh = 0;
				for (int k = 0; k < 4; k++) {
					if (Wn.ref[k] == kWideEmptyRef) continue;
					float lo[3], hi[3];
					for (int a = 0; a < 3; a++) { lo[a] = std::fmaf((float)Wn.q[6 * k + a], scale[a], Wn.org[a]); hi[a] = std::fmaf((float)Wn.q[6 * k + 3 + a], scale[a], Wn.org[a]); }
					t[k] = slab(lo, hi, o, inv, maxDist);
					bool h = t[k] < kFltMax;
					if (h && !any_hit && t[k] > bt * kCullMargin) h = false;
					if (h) idx[nh++] = k;
				}
				if (!any_hit)
					for (int a = 1; a < nh; a++)
						for (int b = a; b > 0 && t[idx[b]] < t[idx[b - 1]]; b--) { const int x = idx[b]; idx[b] = idx[b - 1]; idx[b - 1] = x; }
				if (nh > 0) {
					for (int a = nh - 1; a >= 1; a--) stk[sp++] = Wn.ref[idx[a]];
					if ((uint64_t)sp > deepest) deepest = (uint64_t)sp;
					if (sp > L.wide_stack) { if (err && err_len) strncpy(err, "stack need exceeded", err_len - 1); return 2; }
					cur = Wn.ref[idx[0]];
					continue;
				}
			} else {
				const uint32_t code = (uint32_t)~cur;
				if ((code & 15u) == 0) { // a top-level leaf: the instance (its exact world box first, unless it is the scene's root itself)
					const SceneLayout::WInstH &I = L.winst[code >> 4];
					reached++;
					if (!at_root && !exact_ok(I.xlo, I.xhi, I.flat_unsafe, o, inv, maxDist)) rejected++;
					else {
						inst = (int)(code >> 4);
						irank = I.rank;
						stk[sp++] = kExit;
						if ((uint64_t)sp > deepest) deepest = (uint64_t)sp;
						if (sp > L.wide_stack) { if (err && err_len) strncpy(err, "stack need exceeded", err_len - 1); return 2; }
						V3 no = {I.r0[0] * o.x + I.r0[1] * o.y + I.r0[2] * o.z + I.r0[3], I.r1[0] * o.x + I.r1[1] * o.y + I.r1[2] * o.z + I.r1[3],
						         I.r2[0] * o.x + I.r2[1] * o.y + I.r2[2] * o.z + I.r2[3]};
						V3 nd = {I.r0[0] * d.x + I.r0[1] * d.y + I.r0[2] * d.z, I.r1[0] * d.x + I.r1[1] * d.y + I.r1[2] * d.z,
						         I.r2[0] * d.x + I.r2[1] * d.y + I.r2[2] * d.z};
						o = no; d = nd;
						inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
						at_root = true; // the mesh tree's root, should it be a single leaf, is entered without a box test
						cur = I.root_ref;
						continue;
					}
				} else {
					const float *rec = &L.leafrec[4 * (size_t)(code >> 4)];
					const uint32_t cnt = code & 15u;
					uint32_t flat;
					memcpy(&flat, rec + 3, 4);
					reached++;
					if (!at_root && !exact_ok(rec, rec + 4, flat, o, inv, maxDist)) rejected++;
					else {
						for (uint32_t k = 0; k < cnt && !found; k++) {
							TriH T;
							memcpy(&T, rec + 8 + 12 * k, sizeof T);
							tests++;
							V3 e1 = {T.e1[0], T.e1[1], T.e1[2]}, e2 = {T.e2[0], T.e2[1], T.e2[2]};
							V3 pv = cross(d, e2);
							float det = dot(e1, pv);
							if (pm_fabs(det) < kEps) continue;
							float idet = pm_rcp(det);
							V3 tv = {o.x - T.v0[0], o.y - T.v0[1], o.z - T.v0[2]};
							float u = dot(tv, pv) * idet;
							if (u < 0.0f || u > 1.0f) continue;
							V3 qv = cross(tv, e1);
							float v = dot(d, qv) * idet;
							if (v < 0.0f || u + v > 1.0f) continue;
							float tt = dot(e2, qv) * idet;
							if (any_hit) {
								if (tt > kEps && tt < maxDist) found = true;
							} else if (tt > kEps) {
								const bool closer = tt < bt;
								const bool tie = tt == bt && btri >= 0 && (irank < birank || (irank == birank && T.rank < btrank));
								if (closer || tie) { bt = tt; bu = u; bv = v; btri = (int)(T.orig & ((1u << L.tri_bits) - 1u)); binst = inst; birank = irank; btrank = T.rank; }
							}
						}
						if (found) break;
					}
				}
				at_root = false;
			}
			bool done = false;
			for (;;) {
				if (sp == 0) { done = true; break; }
				cur = stk[--sp];
				if (cur != kExit) break;
				o = O; d = D;
				inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
			}
			if (done) break;
		}
		int32_t *H = hit + 6 * (size_t)r;
		if (any_hit) { H[0] = H[1] = H[2] = H[3] = H[4] = 0; H[5] = found; }
		else {
			H[0] = btri; H[1] = binst;
			memcpy(&H[2], &bt, 4); memcpy(&H[3], &bu, 4); memcpy(&H[4], &bv, 4);
			H[5] = btri >= 0;
		}
	}
	if (counters) {
		counters[0] = steps; counters[1] = tests; counters[2] = reached; counters[3] = rejected;
		counters[4] = L.wide.size(); counters[5] = (uint64_t)L.wide_stack; counters[6] = deepest; counters[7] = (uint64_t)L.wide_narrow;
	}
	return 0;
}
}
