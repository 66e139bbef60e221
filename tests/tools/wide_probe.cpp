// wide_probe.cpp -- exploration tool (not product, not a test): what would a FOUR-WIDE tree with conservatively quantised child
// boxes (64 bytes per node: one gather per step, like today's pair node) and the exact reference box tested on arrival at a
// leaf buy on the gather-bound scenes?  Counts, for a set of rays: pair steps today, wide steps, leaves reached / rejected by
// the exact test, triangle tests, the deepest stack seen, and the static stack bound -- and checks that the closest hits are
// the pair tree's.  (Axis-parallel rays on box faces are not probed here: the flat-box rule of DESIGN.md is a separate matter.)
//   g++ -O2 -ffp-contract=off -shared -fPIC -Iinclude -Ipolaris_amd/csrc tests/tools/wide_probe.cpp -o tests/_build/libwide_probe.so
#include <algorithm>
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
constexpr float kEps = 0.00001f;
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

struct WideNode {
	int n = 0;
	int32_t ref[4];                 // >= 0: wide node index; < 0: leaf reference of the pair tree
	float lo[4][3], hi[4][3];       // decoded (conservative) boxes
	float xlo[4][3], xhi[4][3];     // exact boxes (what the leaf header would hold)
};
} // namespace

extern "C" {

// counters: [0] pair steps, [1] wide steps, [2] leaves reached (wide), [3] leaves rejected by the exact box, [4] triangle tests (wide),
// [5] triangle tests (pair), [6] deepest stack (wide, dynamic), [7] static stack bound (wide), [8] pair stack need, [9] mismatching rays,
// [10] wide nodes, [11] used child slots, [12] leaves reached (pair)
int wide_probe(const PolarisSceneView *sc, int max_leaf_tris, int bits, const float *rays, uint32_t n, uint64_t *counters, char *err, size_t err_len) {
	SceneLayout L;
	std::string e = build_layout(*sc, L, max_leaf_tris);
	if (e == "@retry-without-subdivision") { L = SceneLayout(); e = build_layout(*sc, L, 0); }
	if (!e.empty()) { if (err && err_len) { strncpy(err, e.c_str(), err_len - 1); err[err_len - 1] = 0; } return 1; }
	const float qmax = (float)((1 << bits) - 1);
	// ---- wide tree: every pair node that is a child of a wide node (or a root) becomes a wide node; children expanded greedily by area
	std::vector<int32_t> wide_of(L.pairs.size(), -1);
	std::vector<int32_t> order;
	auto want = [&](int32_t ref) -> int32_t { if (ref < 0) return ref; if (wide_of[ref] < 0) { wide_of[ref] = (int32_t)order.size(); order.push_back(ref); } return wide_of[ref]; };
	const int32_t wroot = want(L.root_ref);
	std::vector<int32_t> inst_wroot(L.insts.size());
	for (size_t i = 0; i < L.insts.size(); i++) inst_wroot[i] = want(L.insts[i].root_ref);
	std::vector<WideNode> W;
	uint64_t used = 0;
	for (size_t head = 0; head < order.size(); head++) {
		const PairNodeH &X = L.pairs[order[head]];
		struct C { float lo[3], hi[3]; int32_t ref; };
		C c[4];
		int k = 0;
		auto put = [&](int at, const float *lo, const float *hi, int32_t ref) { memcpy(c[at].lo, lo, 12); memcpy(c[at].hi, hi, 12); c[at].ref = ref; };
		put(k++, X.lo0, X.hi0, X.ref0);
		put(k++, X.lo1, X.hi1, X.ref1);
		while (k < 4) {
			int best = -1;
			float ba = -1;
			for (int i = 0; i < k; i++) {
				if (c[i].ref < 0) continue;
				const float dx = c[i].hi[0] - c[i].lo[0], dy = c[i].hi[1] - c[i].lo[1], dz = c[i].hi[2] - c[i].lo[2];
				const float a = dx * dy + dy * dz + dz * dx;
				if (a > ba) { ba = a; best = i; }
			}
			if (best < 0) break;
			const PairNodeH &G = L.pairs[c[best].ref];
			for (int i = k; i > best + 1; i--) c[i] = c[i - 1];
			k++;
			put(best, G.lo0, G.hi0, G.ref0);
			put(best + 1, G.lo1, G.hi1, G.ref1);
		}
		WideNode w;
		w.n = k;
		float blo[3] = {3e38f, 3e38f, 3e38f}, bhi[3] = {-3e38f, -3e38f, -3e38f};
		for (int i = 0; i < k; i++) for (int a = 0; a < 3; a++) { blo[a] = std::fmin(blo[a], c[i].lo[a]); bhi[a] = std::fmax(bhi[a], c[i].hi[a]); }
		for (int i = 0; i < k; i++) {
			w.ref[i] = c[i].ref; // fixed up below (want() may grow `order`)
			for (int a = 0; a < 3; a++) {
				w.xlo[i][a] = c[i].lo[a]; w.xhi[i][a] = c[i].hi[a];
				// scale = a power of two >= extent / qmax; decode = fma(q, scale, origin); q chosen outward
				const float ext = bhi[a] - blo[a];
				int ex;
				float sc2 = ext > 0 ? std::frexp(ext / qmax, &ex) : 0.0f;
				float scale = ext > 0 ? std::ldexp(1.0f, sc2 == 0.5f ? ex - 1 : ex) : 0.0f;
				float ql = scale > 0 ? std::floor((c[i].lo[a] - blo[a]) / scale) : 0.0f, qh = scale > 0 ? std::ceil((c[i].hi[a] - blo[a]) / scale) : 0.0f;
				ql = std::fmax(0.0f, std::fmin(ql, qmax)); qh = std::fmax(0.0f, std::fmin(qh, qmax));
				while (ql > 0 && std::fmaf(ql, scale, blo[a]) > c[i].lo[a]) ql -= 1;
				while (qh < qmax && std::fmaf(qh, scale, blo[a]) < c[i].hi[a]) qh += 1;
				w.lo[i][a] = std::fmaf(ql, scale, blo[a]);
				w.hi[i][a] = std::fmaf(qh, scale, blo[a]);
				if (w.lo[i][a] > c[i].lo[a]) w.lo[i][a] = c[i].lo[a]; // (scale 0 / clamped: fall back to the exact bound)
				if (w.hi[i][a] < c[i].hi[a]) w.hi[i][a] = c[i].hi[a];
			}
		}
		for (int i = 0; i < k; i++) w.ref[i] = want(c[i].ref);
		used += (uint64_t)k;
		W.push_back(w);
	}
	// ---- static stack bound of the wide tree
	std::vector<int> need(W.size(), -1);
	{
		std::vector<int32_t> st;
		auto need_of = [&](int32_t root) {
			if (root < 0) return;
			st.push_back(root);
			while (!st.empty()) {
				const int32_t q = st.back();
				if (need[q] >= 0) { st.pop_back(); continue; }
				bool ready = true;
				int deepest = 0;
				for (int i = 0; i < W[q].n; i++) {
					const int32_t r = W[q].ref[i];
					int d = 0;
					if (r >= 0) { if (need[r] < 0) { ready = false; st.push_back(r); } else d = need[r]; }
					else {
						const uint32_t code = (uint32_t)~r;
						if ((code & 15u) == 0 && !(code & kBigLeafFlag)) {
							const int32_t ir = inst_wroot[code >> 4];
							if (ir >= 0) { if (need[ir] < 0) { ready = false; st.push_back(ir); } else d = 1 + need[ir]; } else d = 1;
						}
					}
					deepest = std::max(deepest, d);
				}
				if (!ready) continue;
				need[q] = W[q].n - 1 + deepest;
				st.pop_back();
			}
		};
		for (size_t i = 0; i < L.insts.size(); i++) need_of(inst_wroot[i]);
		need_of(wroot);
	}
	uint64_t psteps = 0, wsteps = 0, wleaves = 0, wrej = 0, wtests = 0, ptests = 0, deepest = 0, mismatch = 0, pleaves = 0;
	std::vector<int> stk(4096);
	// exact box of a leaf reference = the box its parent slot holds; carried beside the reference on the stack in this prototype
	struct Ent { int32_t ref; const float *xlo, *xhi; };
	std::vector<Ent> wst(4096);
	for (uint32_t r = 0; r < n; r++) {
		const float *R = rays + 8 * (size_t)r;
		const V3 O = {R[0], R[1], R[2]}, D = {R[4], R[5], R[6]};
		const float maxDist = R[3];
		float res_t[2];
		int res_tri[2];
		for (int mode = 0; mode < 2; mode++) { // 0 = pair tree, 1 = wide tree
			V3 o = O, d = D, inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
			int sp = 0, btri = -1;
			uint32_t irank = 0, birank = 0, btrank = 0;
			float bt = maxDist;
			Ent cur = {mode ? wroot : L.root_ref, nullptr, nullptr};
			for (;;) {
				if (cur.ref >= 0) {
					if (mode == 0) {
						const PairNodeH &P = L.pairs[cur.ref];
						psteps++;
						float t0 = slab(P.lo0, P.hi0, o, inv, maxDist), t1 = slab(P.lo1, P.hi1, o, inv, maxDist);
						if (t0 > bt * kCullMargin) t0 = kFltMax;
						if (t1 > bt * kCullMargin) t1 = kFltMax;
						int c0 = P.ref0, c1 = P.ref1;
						const bool h0 = t0 < kFltMax, h1 = t1 < kFltMax;
						if (h0 && h1) { if (t1 < t0) std::swap(c0, c1); wst[sp++] = {c1, nullptr, nullptr}; cur = {c0, nullptr, nullptr}; continue; }
						if (h0 || h1) { cur = {h0 ? c0 : c1, nullptr, nullptr}; continue; }
					} else {
						const WideNode &w = W[cur.ref];
						wsteps++;
						float t[4];
						int idx[4], nh = 0;
						for (int i = 0; i < w.n; i++) {
							t[i] = slab(w.lo[i], w.hi[i], o, inv, maxDist);
							if (t[i] < kFltMax && !(t[i] > bt * kCullMargin)) idx[nh++] = i;
						}
						for (int a = 1; a < nh; a++) for (int b = a; b > 0 && t[idx[b]] < t[idx[b - 1]]; b--) std::swap(idx[b], idx[b - 1]);
						if (nh > 0) {
							for (int a = nh - 1; a >= 1; a--) wst[sp++] = {w.ref[idx[a]], w.xlo[idx[a]], w.xhi[idx[a]]};
							if ((uint64_t)sp > deepest) deepest = (uint64_t)sp;
							cur = {w.ref[idx[0]], w.xlo[idx[0]], w.xhi[idx[0]]};
							continue;
						}
					}
				} else {
					const uint32_t code = (uint32_t)~cur.ref;
					bool reached = true;
					if (mode == 1 && cur.xlo) { // exact reference box on arrival
						wleaves++;
						const float tx = slab(cur.xlo, cur.xhi, o, inv, maxDist);
						if (!(tx < kFltMax)) { reached = false; wrej++; }
					}
					if (reached) {
						const LeafInfoH li = (code & 15u) ? LeafInfoH{-(int32_t)(code >> 4), (int32_t)(code & 15u)}
						                   : (!(code & kBigLeafFlag) ? LeafInfoH{-(int32_t)(code >> 4), 0} : L.leaves[(code & (kBigLeafFlag - 1u)) >> 4]);
						if (li.rdata == 0) {
							const InstH &I = L.insts[-li.ldata];
							irank = I.rank;
							wst[sp++] = {kExit, nullptr, nullptr};
							if ((uint64_t)sp > deepest && mode) deepest = (uint64_t)sp;
							V3 no = {I.r0[0] * o.x + I.r0[1] * o.y + I.r0[2] * o.z + I.r0[3], I.r1[0] * o.x + I.r1[1] * o.y + I.r1[2] * o.z + I.r1[3],
							         I.r2[0] * o.x + I.r2[1] * o.y + I.r2[2] * o.z + I.r2[3]};
							V3 nd = {I.r0[0] * d.x + I.r0[1] * d.y + I.r0[2] * d.z, I.r1[0] * d.x + I.r1[1] * d.y + I.r1[2] * d.z,
							         I.r2[0] * d.x + I.r2[1] * d.y + I.r2[2] * d.z};
							o = no; d = nd;
							inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
							cur = {mode ? inst_wroot[-li.ldata] : I.root_ref, nullptr, nullptr};
							continue;
						}
						if (mode == 0) pleaves++;
						const int first = -li.ldata;
						for (int tI = first; tI < first + li.rdata; tI++) {
							const TriH &T = L.tris[tI];
							(mode ? wtests : ptests)++;
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
							if (tt > kEps) {
								const bool closer = tt < bt;
								const bool tie = tt == bt && btri >= 0 && (irank < birank || (irank == birank && T.rank < btrank));
								if (closer || tie) { bt = tt; btri = (int)(T.orig & ((1u << L.tri_bits) - 1u)); birank = irank; btrank = T.rank; }
							}
						}
					}
				}
				bool done = false;
				for (;;) {
					if (sp == 0) { done = true; break; }
					cur = wst[--sp];
					if (cur.ref != kExit) break;
					o = O; d = D;
					inv = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
				}
				if (done) break;
			}
			res_t[mode] = bt; res_tri[mode] = btri;
		}
		if (res_tri[0] != res_tri[1] || (res_tri[0] >= 0 && memcmp(&res_t[0], &res_t[1], 4) != 0)) mismatch++;
	}
	int wneed = 0;
	if (wroot >= 0) wneed = need[wroot];
	for (size_t i = 0; i < L.insts.size(); i++) if (inst_wroot[i] >= 0) wneed = std::max(wneed, need[inst_wroot[i]]);
	if (wroot >= 0) wneed = std::max(wneed, need[wroot]);
	counters[0] = psteps; counters[1] = wsteps; counters[2] = wleaves; counters[3] = wrej; counters[4] = wtests; counters[5] = ptests;
	counters[6] = deepest; counters[7] = (uint64_t)(wneed + 1); counters[8] = (uint64_t)L.max_stack; counters[9] = mismatch; counters[10] = W.size(); counters[11] = used;
	counters[12] = pleaves;
	return 0;
}
}
