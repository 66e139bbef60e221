// layout_check.cpp -- CPU harness over polaris_amd/csrc/scene_layout.h (test tool, not product).
//
// Builds the traversal layout the HIP backend uploads and walks it on the CPU with the same
// rules as polaris_amd/csrc/kernels.h traverse<false>(): same slab test, same Moeller-Trumbore
// order of operations, near child first, conservative pruning, DFS-rank tie break.  Used by
// tests/test_scene_layout.py to show that subdividing leaves at upload (max_leaf_tris) never
// changes a hit record, and to count box/triangle tests per ray.
//
//   g++ -O2 -ffp-contract=off -shared -fPIC -Iinclude -Ipolaris_amd/csrc layout_check.cpp
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

// Small scenes: the triangle slots as the tiny-scene traversal mode wants them (scene_layout.h): leaves in descending order of
// their box's surface area, every slot's packed word = rank << 19 | shading class << 11 | triangle.
// out_area[slot] = surface-area measure of the box of the leaf that holds the slot (-1: the tree is one leaf),
// out_word[slot] = TriH::pad2, out_rank / out_orig = the fields it packs.  Returns the number of slots, < 0 on error.
int layout_check_slots(const PolarisSceneView *sc, int max_leaf_tris, uint32_t cap, float *out_area, uint32_t *out_word, uint32_t *out_rank, uint32_t *out_orig,
                       char *err, size_t err_len) {
	SceneLayout L;
	std::string e = build_layout(*sc, L, max_leaf_tris);
	if (e == "@retry-without-subdivision") { L = SceneLayout(); e = build_layout(*sc, L, 0); }
	if (!e.empty()) { if (err && err_len) { strncpy(err, e.c_str(), err_len - 1); err[err_len - 1] = 0; } return -1; }
	if (L.tris.size() > cap) return -2;
	std::vector<float> area(L.tris.size(), -1.0f);
	auto note = [&](int32_t ref, const float *lo, const float *hi) {
		if (ref >= 0) return;
		const uint32_t code = (uint32_t)~ref;
		if ((code & 15u) == 0u || (code & kBigLeafFlag)) return;
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		for (uint32_t q = code >> 4; q < (code >> 4) + (code & 15u) && q < area.size(); q++) area[q] = dx * dy + dy * dz + dz * dx;
	};
	for (const PairNodeH &P : L.pairs) { note(P.ref0, P.lo0, P.hi0); note(P.ref1, P.lo1, P.hi1); }
	for (size_t q = 0; q < L.tris.size(); q++) { out_area[q] = area[q]; out_word[q] = L.tris[q].pad2; out_rank[q] = L.tris[q].rank; out_orig[q] = L.tris[q].orig; }
	return (int)L.tris.size();
}
}
