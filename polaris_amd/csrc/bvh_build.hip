// bvh_build.hip -- polaris_hip_build_bvh: the two-level BVH built ON THE DEVICE (the stretch).
//
// The reference's builder (asset/compiler/bvh/bvh_builder.go:100-308) scores ~1024 / (depth + 1) candidate planes per axis with
// one goroutine each, every one an O(n) pass over the node's items: fine for a few thousand triangles, minutes for a million.
// polaris_amd/host/scene_compiler.cpp restates it for the CPU.  This file is an ALTERNATIVE producer for the same arrays
// (PolarisBvhNode in the reference's encoding, optimized_scene.go:14-64) shaped for the GPU, with two algorithms:
//
//   POLARIS_BVH_SAH (the default, round 5): a surface-area-heuristic tree built LEVEL BY LEVEL -- every node of a level is split at
//   once: centroid bounds and 16 bins per axis by atomics (aggregated per wave / per workgroup in LDS while a node still spans
//   whole workgroups), the cheapest of the 3 x 15 planes per node, a STABLE partition of the item array by one prefix sum over the
//   whole level.  The same criterion as polaris_amd/scenes.py's CPU producer (and, plane count aside, as bvh_builder.go:162-211),
//   so the trees trace like the CPU-built ones; no sort anywhere; leaves of <= max_leaf_tris items, so the upload-time leaf
//   subdivision (scene_layout.h) finds nothing left to do.  Depth: a node whose centroids no plane separates is halved by position, so a
//   mesh that packs thousands of triangles into one cell of a Morton grid -- which the linear builder turns into a chain deeper than the
//   traversal stack -- builds like any other; and since a binned split alone only bounds the depth by the float range of the centroid
//   extents (geometrically spaced outliers peel off one per level), every node carries a DEPTH BUDGET (sah_levels_needed): a split
//   that would leave a child more items than the remaining levels can finish is replaced by the halving.  The meshes' trees stay within
//   30 levels minus the top-level tree's share, whatever the geometry.
//
//   POLARIS_BVH_LBVH (round 4): a linear BVH (Morton order of the centroids on a grid whose cells stay near-cubic, one radix sort,
//   the hierarchy of Karras 2012 for all inner nodes at once, boxes fitted bottom-up, subtrees of up to max_leaf_tris items
//   collapsed into leaves).  2-4 x faster to build, 28-44 % slower to trace (profiles/r04_bvh_build.json): for trees that must
//   exist in a fraction of a frame.  (Round 4 also built PLOC -- Meister & Bittner 2018 -- on the same infrastructure: better on
//   the Cornell box, level elsewhere, too DEEP for the 32-entry traversal stack on the 58 K-triangle ball; removed again,
//   profiles/r04_bvh_build_lbvh_vs_ploc.json.)
//
// One tree per mesh over its triangles, one over the instances' world boxes (one instance per leaf, compiler.go:88-103).  The
// two library primitives used -- an exclusive prefix sum, and the radix sort of the linear builder -- are rocPRIM's (ROCm's own
// primitives, called directly: no CUB layer).
//
// Neither algorithm can reproduce the reference's tree (the reference breaks equal-score ties by goroutine arrival, SURVEY.md
// 5.2), and need not: the traversal is correct for ANY tree whose boxes contain their items, and parity is defined on the
// uploaded arrays.  What is checked instead (tests/test_gpu_bvh_build.py): the tree is valid under
// scene_layout.h's rules, every item sits in exactly one leaf, every box contains what is below it, two builds are byte-identical,
// and the HIP trace of a scene on the tree built here equals the CPU oracle's trace of the same arrays bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp> // (after <cstring>: its texture iterator calls memset unqualified)

#include "polaris_hip.h"

namespace {

constexpr int BT = 256;
constexpr uint32_t kLeafBit = 0x80000000u; // child word of the Karras tree: a sorted item, not an inner node

thread_local std::string g_build_error;

struct Box { float lo[3], hi[3]; };

// order-preserving float <-> uint (for atomicMin / atomicMax on bounds)
__device__ __forceinline__ uint32_t f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float o2f(uint32_t o) {
	const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
	float f;
	memcpy(&f, &u, 4);
	return f;
}

// boxes of triangles [first, first + n) of the vertex array (float4 per vertex)
__global__ __launch_bounds__(BT) void k_tri_boxes(const float4 *verts, uint32_t first, uint32_t n, Box *boxes) {
	const uint32_t i = blockIdx.x * BT + threadIdx.x;
	if (i >= n) return;
	const float4 a = verts[3 * (size_t)(first + i)], b = verts[3 * (size_t)(first + i) + 1], c = verts[3 * (size_t)(first + i) + 2];
	Box bx;
	bx.lo[0] = fminf(a.x, fminf(b.x, c.x)); bx.lo[1] = fminf(a.y, fminf(b.y, c.y)); bx.lo[2] = fminf(a.z, fminf(b.z, c.z));
	bx.hi[0] = fmaxf(a.x, fmaxf(b.x, c.x)); bx.hi[1] = fmaxf(a.y, fmaxf(b.y, c.y)); bx.hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
	boxes[i] = bx;
}

// bounds of the items' CENTROIDS (what the tree's Morton grid spans): wave reduction, one atomic per wave and word
__global__ __launch_bounds__(BT) void k_box_bounds(const Box *boxes, uint32_t n, uint32_t *bounds) {
	const uint32_t i = blockIdx.x * BT + threadIdx.x;
	for (int k = 0; k < 3; k++) {
		float c = i < n ? 0.5f * boxes[i].lo[k] + 0.5f * boxes[i].hi[k] : 3.0e38f, C = i < n ? c : -3.0e38f;
		for (int s = 32; s > 0; s >>= 1) { c = fminf(c, __shfl_xor(c, s)); C = fmaxf(C, __shfl_xor(C, s)); }
		if ((threadIdx.x & 63) == 0) { atomicMin(&bounds[k], f2o(c)); atomicMax(&bounds[3 + k], f2o(C)); }
	}
}

// key = 30-bit code of the centroid << 32 | item index (unique: the sort is total).  The code is a Morton code whose bits go to
// the axes in the order of the CELL'S SIZE -- every bit halves the axis along which the current cell is longest -- so that cells
// stay near-cubic at every level whatever the shape of the mesh: a heightfield spends its first bits on the two axes it
// extends along, not every third bit on the one it barely has (which would cut it into slabs by height).
__global__ __launch_bounds__(BT) void k_morton(const Box *boxes, uint32_t n, const uint32_t *bounds, uint64_t *keys) {
	const uint32_t i = blockIdx.x * BT + threadIdx.x;
	if (i >= n) return;
	float cell[3];
	uint32_t q[3], used[3] = {0, 0, 0};
	for (int k = 0; k < 3; k++) {
		const float lo = o2f(bounds[k]), ext = o2f(bounds[3 + k]) - lo;
		const float c = 0.5f * boxes[i].lo[k] + 0.5f * boxes[i].hi[k];
		const float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
		q[k] = (uint32_t)fminf(fmaxf(t * 65536.0f, 0.0f), 65535.0f);
		cell[k] = ext > 0.0f ? ext : 0.0f;
	}
	uint32_t code = 0;
	for (int b = 0; b < 30; b++) {
		int a = 0;
		if (cell[1] > cell[a]) a = 1;
		if (cell[2] > cell[a]) a = 2;
		uint32_t bit = 0;
		if (cell[a] > 0.0f) {
			bit = (q[a] >> (15u - used[a])) & 1u;
			used[a]++;
			cell[a] = used[a] < 16u ? 0.5f * cell[a] : 0.0f;
		}
		code = code << 1 | bit;
	}
	keys[i] = (uint64_t)code << 32 | i;
}

// The keys the HIERARCHY is built on: the sorted item's code, then its POSITION in the sorted order (not its original index: items
// that share a code then split by position, i.e. into runs of neighbours on the curve, and the depth below a full cell is
// log2 of its population).  `bits` < 30 would cut the code short and leave more to the positions -- a shallower tree; measured
// (cells of ~8 items): the bottom levels, split by position alone, cost more than the depth saves (terrain 28.7 -> 34.9 ms per
// frame) -- so the full code is used.
__global__ __launch_bounds__(BT) void k_hier_keys(const uint64_t *keys, uint32_t n, uint32_t bits, uint64_t *hkeys) {
	const uint32_t p = blockIdx.x * BT + threadIdx.x;
	if (p >= n) return;
	const uint64_t code = bits ? (keys[p] >> 32) >> (30u - bits) : 0ull;
	hkeys[p] = code << 32 | p;
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees": inner node i of n - 1, all at once.
// child word: index of an inner node, or kLeafBit | position in sorted order.  range[i] = sorted positions the node covers.
__device__ __forceinline__ int delta(const uint64_t *keys, int n, int i, int j) {
	if (j < 0 || j >= n) return -1;
	return __clzll((long long)(keys[i] ^ keys[j]));
}
__global__ __launch_bounds__(BT) void k_hierarchy(const uint64_t *keys, uint32_t n, uint32_t *left, uint32_t *right, uint32_t *parent_inner,
                                                  uint32_t *parent_leaf, uint2 *range) {
	const int i = (int)(blockIdx.x * BT + threadIdx.x);
	const int N = (int)n;
	if (i >= N - 1) return;
	const int d = (delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1)) >= 0 ? 1 : -1;
	const int dmin = delta(keys, N, i, i - d);
	int lmax = 2;
	while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
	int l = 0;
	for (int t = lmax / 2; t >= 1; t /= 2)
		if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
	const int j = i + l * d;
	const int dnode = delta(keys, N, i, j);
	int s = 0;
	for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
		if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
		if (t <= 1) break;
	}
	const int gamma = i + s * d + min(d, 0);
	const int lo = min(i, j), hi = max(i, j);
	const uint32_t L = lo == gamma ? (kLeafBit | (uint32_t)gamma) : (uint32_t)gamma;
	const uint32_t R = hi == gamma + 1 ? (kLeafBit | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
	left[i] = L; right[i] = R;
	range[i] = make_uint2((uint32_t)lo, (uint32_t)hi);
	if (L & kLeafBit) parent_leaf[gamma] = (uint32_t)i; else parent_inner[gamma] = (uint32_t)i;
	if (R & kLeafBit) parent_leaf[gamma + 1] = (uint32_t)i; else parent_inner[gamma + 1] = (uint32_t)i;
	if (i == 0) parent_inner[0] = 0xFFFFFFFFu;
}

// a box another CU has just written: read past this CU's L1 (agent-scope loads), one float at a time
__device__ __forceinline__ Box load_box_agent(const Box *p) {
	Box b;
	const float *f = reinterpret_cast<const float *>(p);
	for (int k = 0; k < 3; k++) {
		b.lo[k] = __hip_atomic_load(f + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		b.hi[k] = __hip_atomic_load(f + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	return b;
}

// boxes bottom-up: every sorted item climbs; the second thread to arrive at an inner node fits it and climbs on
__global__ __launch_bounds__(BT) void k_fit(const uint64_t *keys, const Box *boxes, uint32_t n, const uint32_t *left, const uint32_t *right,
                                            const uint32_t *parent_inner, const uint32_t *parent_leaf, Box *inner_box, uint32_t *arrived) {
	const uint32_t p = blockIdx.x * BT + threadIdx.x;
	if (p >= n || n < 2) return;
	uint32_t cur = parent_leaf[p];
	for (;;) {
		__threadfence();
		if (atomicAdd(&arrived[cur], 1u) == 0u) return; // first: the sibling will do it
		const uint32_t L = left[cur], R = right[cur];
		const Box a = (L & kLeafBit) ? boxes[(uint32_t)(keys[L & ~kLeafBit] & 0xFFFFFFFFu)] : load_box_agent(inner_box + L);
		const Box b = (R & kLeafBit) ? boxes[(uint32_t)(keys[R & ~kLeafBit] & 0xFFFFFFFFu)] : load_box_agent(inner_box + R);
		Box u;
		for (int k = 0; k < 3; k++) { u.lo[k] = fminf(a.lo[k], b.lo[k]); u.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
		inner_box[cur] = u;
		const uint32_t up = parent_inner[cur];
		if (up == 0xFFFFFFFFu) return;
		cur = up;
	}
}

// Output nodes.  Inner node i SURVIVES when it covers more than max_leaf items; a child that does not survive (an inner node
// of <= max_leaf items, or a single item) becomes a LEAF node covering its whole range.  need[i] = nodes a surviving inner node
// emits: itself + its leaf children; an exclusive scan gives their places.
__global__ __launch_bounds__(BT) void k_need(uint32_t n, uint32_t max_leaf, const uint32_t *left, const uint32_t *right, const uint2 *range,
                                             const uint32_t *parent_inner, uint32_t *need) {
	const uint32_t i = blockIdx.x * BT + threadIdx.x;
	if (i + 1 >= n) return;
	const uint2 r = range[i];
	uint32_t c = 0;
	if (r.y - r.x + 1 > max_leaf) {
		c = 1;
		const uint32_t kids[2] = {left[i], right[i]};
		for (int k = 0; k < 2; k++) {
			if (kids[k] & kLeafBit) c++;
			else { const uint2 q = range[kids[k]]; if (q.y - q.x + 1 <= max_leaf) c++; }
		}
	}
	need[i] = c;
}

__global__ __launch_bounds__(BT) void k_emit(uint32_t n, uint32_t max_leaf, const uint64_t *keys, const Box *boxes, const uint32_t *left,
                                             const uint32_t *right, const uint2 *range, const Box *inner_box, const uint32_t *place, uint32_t node_base,
                                             uint32_t item_base, int instances, PolarisBvhNode *out) {
	const uint32_t i = blockIdx.x * BT + threadIdx.x;
	if (i + 1 >= n) return;
	const uint2 r = range[i];
	if (r.y - r.x + 1 <= max_leaf) return;
	const uint32_t me = place[i];
	uint32_t next = me + 1;
	int32_t child[2];
	const uint32_t kids[2] = {left[i], right[i]};
	for (int k = 0; k < 2; k++) {
		bool leaf = (kids[k] & kLeafBit) != 0;
		uint32_t lo, hi;
		Box b;
		uint32_t inst = 0;
		if (leaf) {
			const uint32_t p = kids[k] & ~kLeafBit;
			inst = (uint32_t)(keys[p] & 0xFFFFFFFFu);
			b = boxes[inst];
			lo = hi = p;
		} else {
			const uint2 q = range[kids[k]];
			lo = q.x; hi = q.y;
			b = inner_box[kids[k]];
			leaf = hi - lo + 1 <= max_leaf;
		}
		if (!leaf) { child[k] = (int32_t)(node_base + place[kids[k]]); continue; }
		PolarisBvhNode nd;
		for (int a = 0; a < 3; a++) { nd.min[a] = b.lo[a]; nd.max[a] = b.hi[a]; }
		if (instances) { nd.ldata = -(int32_t)inst; nd.rdata = 0; } // top-level leaf: ONE instance (max_leaf = 1: never a collapsed subtree)
		else { nd.ldata = -(int32_t)(item_base + lo); nd.rdata = (int32_t)(hi - lo + 1); }        // triangles [first, first + count) of the NEW order
		out[node_base + next] = nd;
		child[k] = (int32_t)(node_base + next);
		next++;
	}
	PolarisBvhNode nd;
	const Box b = inner_box[i];
	for (int a = 0; a < 3; a++) { nd.min[a] = b.lo[a]; nd.max[a] = b.hi[a]; }
	nd.ldata = child[0]; nd.rdata = child[1];
	out[node_base + me] = nd;
}

// a tree of n <= max_leaf items: one leaf node
__global__ void k_single_leaf(const uint64_t *keys, const Box *boxes, uint32_t n, uint32_t node_base, uint32_t item_base, int instances, PolarisBvhNode *out) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	PolarisBvhNode nd;
	for (int a = 0; a < 3; a++) { nd.min[a] = 3.0e38f; nd.max[a] = -3.0e38f; }
	for (uint32_t p = 0; p < n; p++) {
		const Box b = boxes[(uint32_t)(keys[p] & 0xFFFFFFFFu)];
		for (int a = 0; a < 3; a++) { nd.min[a] = fminf(nd.min[a], b.lo[a]); nd.max[a] = fmaxf(nd.max[a], b.hi[a]); }
	}
	if (instances) { nd.ldata = -(int32_t)(uint32_t)(keys[0] & 0xFFFFFFFFu); nd.rdata = 0; }
	else { nd.ldata = -(int32_t)item_base; nd.rdata = (int32_t)n; }
	out[node_base] = nd;
}

__global__ __launch_bounds__(BT) void k_order(const uint64_t *keys, uint32_t n, uint32_t item_base, uint32_t *order) {
	const uint32_t p = blockIdx.x * BT + threadIdx.x;
	if (p < n) order[item_base + p] = item_base + (uint32_t)(keys[p] & 0xFFFFFFFFu); // (the tree keeps the Morton order: sorted item p is item p)
}

struct Scratch { // device buffers sized for the largest tree of the call
	uint64_t *keys = nullptr, *keys_alt = nullptr;
	Box *boxes = nullptr, *inner_box = nullptr;
	uint32_t *left = nullptr, *right = nullptr, *parent_inner = nullptr, *parent_leaf = nullptr, *arrived = nullptr, *need = nullptr, *place = nullptr, *bounds = nullptr;
	uint2 *range = nullptr;
	void *temp = nullptr;
	size_t temp_bytes = 0;
	std::vector<void *> all;
	template <typename T> hipError_t get(T **p, size_t count) {
		void *q = nullptr;
		const hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
		if (e == hipSuccess) { all.push_back(q); *p = (T *)q; }
		return e;
	}
	~Scratch() { for (void *q : all) (void)hipFree(q); }
};

#define BUILD_TRY(expr)                                                                                       \
	do {                                                                                                      \
		const hipError_t e_ = (expr);                                                                         \
		if (e_ != hipSuccess) { g_build_error = std::string(#expr) + ": " + hipGetErrorString(e_); return POLARIS_E_DEVICE; } \
	} while (0)

inline uint32_t grid(uint32_t n) { return (n + BT - 1) / BT; }

// One tree over n items whose boxes are in S.boxes: nodes to out[node_base ...], returns the number of nodes emitted.
int build_tree(Scratch &S, hipStream_t q, uint32_t n, uint32_t max_leaf, uint32_t node_base, uint32_t item_base, int instances,
               PolarisBvhNode *d_out, uint32_t *d_order, uint32_t *emitted) {
	BUILD_TRY(hipMemsetAsync(S.bounds, 0xFF, 3 * sizeof(uint32_t), q));
	BUILD_TRY(hipMemsetAsync(S.bounds + 3, 0x00, 3 * sizeof(uint32_t), q));
	hipLaunchKernelGGL(k_box_bounds, dim3(grid(n)), dim3(BT), 0, q, S.boxes, n, S.bounds);
	hipLaunchKernelGGL(k_morton, dim3(grid(n)), dim3(BT), 0, q, S.boxes, n, S.bounds, S.keys_alt);
	size_t tb = S.temp_bytes;
	BUILD_TRY(rocprim::radix_sort_keys(S.temp, tb, S.keys_alt, S.keys, (size_t)n, 0u, 62u, q));
	const bool single = n <= max_leaf || n < 2;
	if (d_order) hipLaunchKernelGGL(k_order, dim3(grid(n)), dim3(BT), 0, q, S.keys, n, item_base, d_order);
	if (single) {
		hipLaunchKernelGGL(k_single_leaf, dim3(1), dim3(64), 0, q, S.keys, S.boxes, n, node_base, item_base, instances, d_out);
		*emitted = 1;
		BUILD_TRY(hipGetLastError());
		return POLARIS_OK;
	}
	BUILD_TRY(hipMemsetAsync(S.arrived, 0, (size_t)(n - 1) * sizeof(uint32_t), q));
	hipLaunchKernelGGL(k_hier_keys, dim3(grid(n)), dim3(BT), 0, q, S.keys, n, 30u, S.keys_alt); // (the sort's input buffer is free again)
	hipLaunchKernelGGL(k_hierarchy, dim3(grid(n - 1)), dim3(BT), 0, q, S.keys_alt, n, S.left, S.right, S.parent_inner, S.parent_leaf, S.range);
	hipLaunchKernelGGL(k_fit, dim3(grid(n)), dim3(BT), 0, q, S.keys, S.boxes, n, S.left, S.right, S.parent_inner, S.parent_leaf, S.inner_box, S.arrived);
	hipLaunchKernelGGL(k_need, dim3(grid(n - 1)), dim3(BT), 0, q, n, max_leaf, S.left, S.right, S.range, S.parent_inner, S.need);
	tb = S.temp_bytes;
	BUILD_TRY(rocprim::exclusive_scan(S.temp, tb, S.need, S.place, 0u, (size_t)(n - 1), rocprim::plus<uint32_t>(), q));
	hipLaunchKernelGGL(k_emit, dim3(grid(n - 1)), dim3(BT), 0, q, n, max_leaf, S.keys, S.boxes, S.left, S.right, S.range, S.inner_box, S.place, node_base,
	                   item_base, instances, d_out);
	BUILD_TRY(hipGetLastError());
	uint32_t last_place = 0, last_need = 0;
	BUILD_TRY(hipMemcpyAsync(&last_place, S.place + (n - 2), 4, hipMemcpyDeviceToHost, q));
	BUILD_TRY(hipMemcpyAsync(&last_need, S.need + (n - 2), 4, hipMemcpyDeviceToHost, q));
	BUILD_TRY(hipStreamSynchronize(q));
	*emitted = last_place + last_need;
	return POLARIS_OK;
}


// =============================================================================================
// Binned SAH, level by level (POLARIS_BVH_SAH)
// =============================================================================================
// The items of a tree sit in ONE array of positions; every node owns a contiguous range of it.  A level:
//   nodes of > 512 items    k_sah_cbounds_big, k_sah_bin_big: centroid bounds, then kBins bins per axis (count + box), by atomics that a
//                           workgroup aggregates in LDS over its tile of 256 positions; k_sah_split_big: one thread per node
//   nodes of <= 512 items   k_sah_small: ONE WAVE per node does all of that in registers and LDS, no global atomic (the deep levels
//                           of a tree are hundreds of thousands of such nodes)
//   the split               the cheapest of 3 x (kBins - 1) planes (cost = n_l area_l + n_r area_r, as scenes.py / bvh_builder.go); the two
//                           children get their node ids, their boxes (unions of bins: exact) and, if they hold <= max_leaf items,
//                           become leaves; a node no plane separates is halved by position
//   prefix sum      over "this item goes left" flags of the whole level (rocPRIM) -- a STABLE partition, so the build is deterministic
//   k_sah_scatter   items move to their child's range; every position learns which active node of the next level owns it
// until no node is left to split.  Node ids are level by level, children adjacent.
constexpr int kBins = 16;
constexpr uint32_t kNone = 0xFFFFFFFFu;
struct SahBin { uint32_t cnt, lo[3], hi[3]; };          // boxes as order-preserving uints (atomicMin / atomicMax)
struct SahAct { uint32_t node, first, count, big; };     // an active node: it holds more than max_leaf items and will be split; big = its index among the
                                                         // level's nodes of more than kSmallMax items (their bins and centroid bounds live in arrays of THAT
                                                         // length: at most n / 513 such nodes exist per level), kNone for the others
struct SahSplit { uint32_t axis, k, nl; float clo, ext; }; // axis kNone = halved by position
struct SahCb { uint32_t lo[3], hi[3]; };

__device__ __forceinline__ float sah_centroid(const Box &b, int a) { return 0.5f * b.lo[a] + 0.5f * b.hi[a]; }
__device__ __forceinline__ uint32_t sah_bin_of(float c, float clo, float ext) { // the ONE expression both the binning and the partition use
	return (uint32_t)fminf(fmaxf((c - clo) / ext * (float)kBins, 0.0f), (float)(kBins - 1));
}

__global__ __launch_bounds__(BT) void k_sah_init(uint32_t n, uint32_t *item, uint32_t *owner) {
	const uint32_t p = blockIdx.x * BT + threadIdx.x;
	if (p < n) { item[p] = p; owner[p] = 0; }
}

// bounds of the item BOXES (the root's box): per thread over a tile, per wave by shuffles, per workgroup in LDS, six atomics per workgroup
__global__ __launch_bounds__(BT) void k_sah_root_bounds(const Box *boxes, uint32_t n, uint32_t *bounds) {
	__shared__ uint32_t l[6];
	if (threadIdx.x < 6) l[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
	__syncthreads();
	float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
	const uint32_t base = blockIdx.x * 2048u;
	for (uint32_t i = base + threadIdx.x; i < min(base + 2048u, n); i += BT)
		for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], boxes[i].lo[k]); hi[k] = fmaxf(hi[k], boxes[i].hi[k]); }
	for (int k = 0; k < 3; k++) {
		for (int s = 32; s > 0; s >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], s)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], s)); }
		if ((threadIdx.x & 63) == 0) { atomicMin(&l[k], f2o(lo[k])); atomicMax(&l[3 + k], f2o(hi[k])); }
	}
	__syncthreads();
	if (threadIdx.x < 3) { atomicMin(&bounds[threadIdx.x], l[threadIdx.x]); atomicMax(&bounds[3 + threadIdx.x], l[3 + threadIdx.x]); }
}
__global__ void k_sah_root_node(const uint32_t *bounds, uint32_t n, uint32_t max_leaf, uint32_t node, uint32_t item_base, int instances, PolarisBvhNode *out, SahAct *act, uint32_t *counts) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	PolarisBvhNode nd;
	for (int a = 0; a < 3; a++) { nd.min[a] = o2f(bounds[a]); nd.max[a] = o2f(bounds[3 + a]); }
	nd.ldata = 0; nd.rdata = 0;
	if (n <= max_leaf) { nd.ldata = -(int32_t)item_base; nd.rdata = instances ? 0 : (int32_t)n; counts[0] = 0; } // the whole tree is one leaf (instances: position 0, resolved by k_sah_finish)
	else { act[0] = SahAct{node, 0u, n, n > 512u ? 0u : 0xFFFFFFFFu}; counts[0] = 1; } // (512 = kSmallMax, kNone: declared below)
	counts[1] = 0;
	counts[2] = 0;
	out[node] = nd;
}

__global__ __launch_bounds__(BT) void k_sah_clear_cb(uint32_t count, SahCb *cb) {
	const uint32_t i = blockIdx.x * BT + threadIdx.x;
	if (i < count) { SahCb c; for (int a = 0; a < 3; a++) { c.lo[a] = 0xFFFFFFFFu; c.hi[a] = 0u; } cb[i] = c; }
}
__global__ __launch_bounds__(BT) void k_sah_clear_bins(uint32_t count, SahBin *bins) {
	const uint32_t i = blockIdx.x * BT + threadIdx.x;
	if (i < count) { SahBin b; b.cnt = 0; for (int a = 0; a < 3; a++) { b.lo[a] = 0xFFFFFFFFu; b.hi[a] = 0u; } bins[i] = b; }
}

// ---- nodes of more than kSmallMax items: two passes over their positions, tiles of kTile positions per workgroup.  A tile's atomics go
// to LDS while its positions belong to the tile's FIRST active node (nodes are contiguous ranges: at the top of the tree that is every
// position of the tile) and are flushed once per workgroup; positions of another node of the tile use global atomics directly.
constexpr uint32_t kSmallMax = 512; // a node of at most this many items is split by ONE wave (k_sah_small)
constexpr uint32_t kTile = 256;      // positions per workgroup of the two big-node passes
constexpr int kBigBlock = 256;        // ... and its threads: one position each (measured: 2048 positions per 256 threads 13.9 ms for 1 M triangles -- too few
                                      // workgroups --, 1024 per 1024 threads 9.8 ms -- sixteen waves at every barrier --, 256 per 256: 6.9 ms)

__global__ __launch_bounds__(kBigBlock) void k_sah_cbounds_big(uint32_t n, const uint32_t *item, const uint32_t *owner, const SahAct *act, const Box *boxes, SahCb *cb) {
	__shared__ SahCb l;
	__shared__ uint32_t s_owner;
	const uint32_t base = blockIdx.x * kTile;
	if (threadIdx.x == 0) {
		s_owner = kNone;
		for (int a = 0; a < 3; a++) { l.lo[a] = 0xFFFFFFFFu; l.hi[a] = 0u; }
	}
	__syncthreads();
	for (uint32_t p = base + threadIdx.x; p < min(base + kTile, n); p += kBigBlock) { // the first big node of the tile (lowest position wins)
		const uint32_t o = owner[p];
		if (o != kNone && act[o].count > kSmallMax) { atomicMin(&s_owner, o); break; } // (active indices ascend with position)
	}
	__syncthreads();
	const uint32_t o0 = s_owner;
	if (o0 == kNone) return;
	float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
	bool any = false;
	for (uint32_t p = base + threadIdx.x; p < min(base + kTile, n); p += kBigBlock) {
		const uint32_t o = owner[p];
		if (o == kNone || act[o].count <= kSmallMax) continue;
		const Box b = boxes[item[p]];
		if (o == o0) { any = true; for (int a = 0; a < 3; a++) { const float c = sah_centroid(b, a); lo[a] = fminf(lo[a], c); hi[a] = fmaxf(hi[a], c); } }
		else for (int a = 0; a < 3; a++) { const float c = sah_centroid(b, a); atomicMin(&cb[act[o].big].lo[a], f2o(c)); atomicMax(&cb[act[o].big].hi[a], f2o(c)); }
	}
	if (any) for (int a = 0; a < 3; a++) { atomicMin(&l.lo[a], f2o(lo[a])); atomicMax(&l.hi[a], f2o(hi[a])); }
	__syncthreads();
	if (threadIdx.x < 3) { const uint32_t b0 = act[o0].big; atomicMin(&cb[b0].lo[threadIdx.x], l.lo[threadIdx.x]); atomicMax(&cb[b0].hi[threadIdx.x], l.hi[threadIdx.x]); }
}

__global__ __launch_bounds__(kBigBlock) void k_sah_bin_big(uint32_t n, const uint32_t *item, const uint32_t *owner, const SahAct *act, const Box *boxes, const SahCb *cb, SahBin *bins) {
	__shared__ SahBin lb[3 * kBins];
	__shared__ uint32_t s_owner;
	const uint32_t base = blockIdx.x * kTile;
	if (threadIdx.x == 0) s_owner = kNone;
	if (threadIdx.x < 3 * kBins) { SahBin b; b.cnt = 0; for (int a = 0; a < 3; a++) { b.lo[a] = 0xFFFFFFFFu; b.hi[a] = 0u; } lb[threadIdx.x] = b; }
	__syncthreads();
	for (uint32_t p = base + threadIdx.x; p < min(base + kTile, n); p += kBigBlock) {
		const uint32_t o = owner[p];
		if (o != kNone && act[o].count > kSmallMax) { atomicMin(&s_owner, o); break; }
	}
	__syncthreads();
	const uint32_t o0 = s_owner;
	if (o0 == kNone) return;
	for (uint32_t p = base + threadIdx.x; p < min(base + kTile, n); p += kBigBlock) {
		const uint32_t o = owner[p];
		if (o == kNone || act[o].count <= kSmallMax) continue;
		const Box b = boxes[item[p]];
		const uint32_t ob = act[o].big; // (bins and centroid bounds are indexed by the node's rank among the level's BIG nodes)
		const SahCb c = cb[ob];
		for (int a = 0; a < 3; a++) {
			const float clo = o2f(c.lo[a]), ext = o2f(c.hi[a]) - clo;
			if (!(ext > 1e-12f)) continue; // (scenes.py: an axis along which the centroids coincide offers no plane)
			const uint32_t k = sah_bin_of(sah_centroid(b, a), clo, ext);
			SahBin *t = o == o0 ? &lb[a * kBins + k] : &bins[((size_t)ob * 3 + a) * kBins + k];
			atomicAdd(&t->cnt, 1u);
			for (int d = 0; d < 3; d++) { atomicMin(&t->lo[d], f2o(b.lo[d])); atomicMax(&t->hi[d], f2o(b.hi[d])); }
		}
	}
	__syncthreads();
	if (threadIdx.x < 3 * kBins && lb[threadIdx.x].cnt) {
		SahBin *t = &bins[(size_t)act[o0].big * 3 * kBins + threadIdx.x];
		const SahBin v = lb[threadIdx.x];
		atomicAdd(&t->cnt, v.cnt);
		for (int d = 0; d < 3; d++) { atomicMin(&t->lo[d], v.lo[d]); atomicMax(&t->hi[d], v.hi[d]); }
	}
}

__device__ __forceinline__ float sah_half_area(const float lo[3], const float hi[3]) {
	const float dx = fmaxf(hi[0] - lo[0], 0.0f), dy = fmaxf(hi[1] - lo[1], 0.0f), dz = fmaxf(hi[2] - lo[2], 0.0f);
	return dx * dy + dy * dz + dx * dz;
}

// The split of ONE node from its bins (B: [3][kBins], global or LDS) and centroid bounds: the cheapest of the 3 x (kBins - 1) planes,
// the children's ids, boxes and leaf records, flags[2 o + side] = 1 when that child is active in the next level.
// The depth budget.  A binned split only shrinks a node's centroid EXTENT: items spaced geometrically (outliers at 16^-k) peel off one
// per level, a chain as long as the float range allows -- deeper than the 32-entry traversal stack, and upload_scene then refuses the
// scene.  Every node therefore carries a budget: the levels that may still follow below it.  levels_needed(m) is what the shallowest
// possible subtree over m items takes (halving all the way); a SAH split is kept only if BOTH children can still be finished within
// the budget, otherwise the node is halved by position -- which always can (ceil(m / 2) items need one level less than m).  A tree
// whose natural SAH depth fits the budget never sees the rule.
__host__ __device__ __forceinline__ uint32_t sah_levels_needed(uint32_t m, uint32_t max_leaf) {
	const uint32_t q = (m + max_leaf - 1) / max_leaf; // leaves, at best
	uint32_t l = 0;
	while ((1u << l) < q) l++;
	return l;
}

__device__ __forceinline__ void sah_split_node(uint32_t o, const SahAct me, const SahCb c, const SahBin *B3, uint32_t max_leaf, uint32_t next_base, uint32_t item_base,
                                               int instances, uint32_t levels_below, PolarisBvhNode *out, SahSplit *split, uint32_t *flags, uint32_t *counts) {
	const uint32_t m = me.count;
	float best = 3.0e38f;
	uint32_t best_axis = kNone, best_k = 0, best_nl = 0;
	for (int a = 0; a < 3; a++) {
		const SahBin *B = B3 + a * kBins;
		// suffix: area and count of bins [k, kBins)
		float ra[kBins];
		uint32_t rc[kBins];
		{
			float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
			uint32_t cc = 0;
#pragma unroll
			for (int k = kBins - 1; k >= 0; k--) {
				if (B[k].cnt) { for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], o2f(B[k].lo[d])); hi[d] = fmaxf(hi[d], o2f(B[k].hi[d])); } cc += B[k].cnt; }
				ra[k] = cc ? sah_half_area(lo, hi) : 0.0f;
				rc[k] = cc;
			}
		}
		if (rc[0] != m) continue; // this axis was not binned (no extent)
		float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
		uint32_t nl = 0;
#pragma unroll
		for (int k = 0; k < kBins - 1; k++) { // left = bins [0, k], right = bins [k + 1, kBins)
			if (B[k].cnt) { for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], o2f(B[k].lo[d])); hi[d] = fmaxf(hi[d], o2f(B[k].hi[d])); } nl += B[k].cnt; }
			if (nl == 0 || rc[k + 1] == 0) continue;
			const float cost = (float)nl * sah_half_area(lo, hi) + (float)rc[k + 1] * ra[k + 1];
			if (cost < best) { best = cost; best_axis = (uint32_t)a; best_k = (uint32_t)k; best_nl = nl; }
		}
	}
	// (levels_below: what may follow below this node's CHILDREN)
	if (best_axis != kNone && sah_levels_needed(max(best_nl, m - best_nl), max_leaf) > levels_below) best_axis = kNone;
	SahSplit sp;
	sp.axis = best_axis; sp.k = best_k; sp.clo = 0.0f; sp.ext = 1.0f;
	PolarisBvhNode kid[2];
	if (best_axis != kNone) {
		sp.nl = best_nl;
		sp.clo = o2f(c.lo[best_axis]);
		sp.ext = o2f(c.hi[best_axis]) - sp.clo;
		const SahBin *B = B3 + best_axis * kBins;
		for (int side = 0; side < 2; side++) {
			float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
			for (uint32_t k = side ? best_k + 1 : 0u; k < (side ? (uint32_t)kBins : best_k + 1); k++)
				if (B[k].cnt) for (int d = 0; d < 3; d++) { lo[d] = fminf(lo[d], o2f(B[k].lo[d])); hi[d] = fmaxf(hi[d], o2f(B[k].hi[d])); }
			for (int d = 0; d < 3; d++) { kid[side].min[d] = lo[d]; kid[side].max[d] = hi[d]; }
		}
	} else { // no plane separates the centroids: halve the range by position; the halves' boxes are fitted by k_sah_fit_halves
		sp.nl = m / 2;
		for (int side = 0; side < 2; side++) for (int d = 0; d < 3; d++) { kid[side].min[d] = out[me.node].min[d]; kid[side].max[d] = out[me.node].max[d]; }
		atomicAdd(&counts[1], 1u);
	}
	split[o] = sp;
	const uint32_t ids[2] = {next_base + 2 * o, next_base + 2 * o + 1};
	const uint32_t firsts[2] = {me.first, me.first + sp.nl}, cnts[2] = {sp.nl, m - sp.nl};
	for (int side = 0; side < 2; side++) {
		const bool leaf = cnts[side] <= max_leaf;
		kid[side].ldata = leaf ? -(int32_t)(item_base + firsts[side]) : 0;
		kid[side].rdata = leaf ? (instances ? 0 : (int32_t)cnts[side]) : 0;
		out[ids[side]] = kid[side];
		flags[2 * o + side] = leaf ? 0u : 1u;
	}
	out[me.node].ldata = (int32_t)ids[0];
	out[me.node].rdata = (int32_t)ids[1];
}

// big nodes: one thread per active node, bins in global memory
__global__ __launch_bounds__(BT) void k_sah_split_big(uint32_t A, const SahAct *act, const SahCb *cb, const SahBin *bins, uint32_t max_leaf, uint32_t next_base,
                                                      uint32_t item_base, int instances, uint32_t levels_below, PolarisBvhNode *out, SahSplit *split, uint32_t *flags, uint32_t *counts) {
	const uint32_t o = blockIdx.x * BT + threadIdx.x;
	if (o >= A || act[o].count <= kSmallMax) return;
	sah_split_node(o, act[o], cb[act[o].big], bins + (size_t)act[o].big * 3 * kBins, max_leaf, next_base, item_base, instances, levels_below, out, split, flags, counts);
}

// small nodes (<= kSmallMax items): ONE WAVE per node does everything -- centroid bounds by a wave reduction, the bins in LDS, the
// split by its first lane -- without a global atomic (the deep levels of a tree are hundreds of thousands of such nodes).
__global__ __launch_bounds__(BT) void k_sah_small(uint32_t A, const SahAct *act, const uint32_t *item, const Box *boxes, uint32_t max_leaf, uint32_t next_base,
                                                  uint32_t item_base, int instances, uint32_t levels_below, PolarisBvhNode *out, SahSplit *split, uint32_t *flags, uint32_t *counts) {
	__shared__ SahBin lb[BT / 64][3 * kBins];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t o = blockIdx.x * (BT / 64) + wave;
	if (o >= A) return;
	const SahAct me = act[o];
	if (me.count > kSmallMax) return;
	float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
	for (uint32_t i = lane; i < me.count; i += 64) {
		const Box b = boxes[item[me.first + i]];
		for (int a = 0; a < 3; a++) { const float c = sah_centroid(b, a); lo[a] = fminf(lo[a], c); hi[a] = fmaxf(hi[a], c); }
	}
	SahCb c;
	for (int a = 0; a < 3; a++) {
		for (int sft = 32; sft > 0; sft >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], sft)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], sft)); }
		c.lo[a] = f2o(lo[a]); c.hi[a] = f2o(hi[a]); // (the same encoding round trip as the big path: sah_bin_of sees identical operands)
	}
	if (lane < 3 * kBins) { SahBin b; b.cnt = 0; for (int a = 0; a < 3; a++) { b.lo[a] = 0xFFFFFFFFu; b.hi[a] = 0u; } lb[wave][lane] = b; }
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	for (uint32_t i = lane; i < me.count; i += 64) {
		const Box b = boxes[item[me.first + i]];
		for (int a = 0; a < 3; a++) {
			const float clo = o2f(c.lo[a]), ext = o2f(c.hi[a]) - clo;
			if (!(ext > 1e-12f)) continue;
			SahBin *t = &lb[wave][a * kBins + sah_bin_of(sah_centroid(b, a), clo, ext)];
			atomicAdd(&t->cnt, 1u);
			for (int d = 0; d < 3; d++) { atomicMin(&t->lo[d], f2o(b.lo[d])); atomicMax(&t->hi[d], f2o(b.hi[d])); }
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	if (lane == 0) sah_split_node(o, me, c, lb[wave], max_leaf, next_base, item_base, instances, levels_below, out, split, flags, counts);
}

__global__ __launch_bounds__(BT) void k_sah_next(uint32_t A, const SahAct *act, const SahSplit *split, const uint32_t *flags, const uint32_t *nidx, uint32_t next_base,
                                                 SahAct *act_next, uint32_t *counts) {
	const uint32_t o = blockIdx.x * BT + threadIdx.x;
	if (o >= A) return;
	const SahAct me = act[o];
	const SahSplit sp = split[o];
	// the children that take the three-pass path in the next level get their rank among that level's big nodes here: counts[2] ends up as
	// their number (0: those kernels are not launched).  Which rank a node gets depends on the order of the atomics -- it only names the
	// node's scratch bins, never anything the tree is made of: two builds stay byte-identical.
	const uint32_t bigl = flags[2 * o] && sp.nl > kSmallMax ? 1u : 0u, bigr = flags[2 * o + 1] && me.count - sp.nl > kSmallMax ? 1u : 0u;
	const uint32_t rank = (bigl + bigr) ? atomicAdd(&counts[2], bigl + bigr) : 0u;
	if (flags[2 * o]) act_next[nidx[2 * o]] = SahAct{next_base + 2 * o, me.first, sp.nl, bigl ? rank : kNone};
	if (flags[2 * o + 1]) act_next[nidx[2 * o + 1]] = SahAct{next_base + 2 * o + 1, me.first + sp.nl, me.count - sp.nl, bigr ? rank + bigl : kNone};
	if (o == A - 1) counts[0] = nidx[2 * o + 1] + flags[2 * o + 1];
}

__global__ __launch_bounds__(BT) void k_sah_flag(uint32_t n, const uint32_t *item, const uint32_t *owner, const Box *boxes, const SahAct *act, const SahSplit *split, uint32_t *left) {
	const uint32_t p = blockIdx.x * BT + threadIdx.x;
	if (p >= n) return;
	const uint32_t o = owner[p];
	uint32_t l = 0;
	if (o != kNone) {
		const SahSplit sp = split[o];
		if (sp.axis == kNone) l = (p - act[o].first) < sp.nl ? 1u : 0u;
		else l = sah_bin_of(sah_centroid(boxes[item[p]], (int)sp.axis), sp.clo, sp.ext) <= sp.k ? 1u : 0u;
	}
	left[p] = l;
}

__global__ __launch_bounds__(BT) void k_sah_scatter(uint32_t n, const uint32_t *item, const uint32_t *owner, const uint32_t *left, const uint32_t *lscan, const SahAct *act,
                                                    const SahSplit *split, const uint32_t *flags, const uint32_t *nidx, uint32_t *item_next, uint32_t *owner_next) {
	const uint32_t p = blockIdx.x * BT + threadIdx.x;
	if (p >= n) return;
	const uint32_t o = owner[p];
	if (o == kNone) { item_next[p] = item[p]; owner_next[p] = kNone; return; }
	const SahAct me = act[o];
	const uint32_t nl = split[o].nl;
	const uint32_t rank_l = lscan[p] - lscan[me.first];
	const uint32_t side = left[p] ? 0u : 1u;
	const uint32_t q = side == 0 ? me.first + rank_l : me.first + nl + (p - me.first - rank_l);
	item_next[q] = item[p];
	owner_next[q] = flags[2 * o + side] ? nidx[2 * o + side] : kNone;
}

// the boxes of the two halves of a node that was halved by position (rare: coincident centroids): per-item atomics into a scratch box
__global__ __launch_bounds__(BT) void k_sah_fit_halves(uint32_t n, const uint32_t *item_next, const Box *boxes, const uint32_t *owner, const uint32_t *left, const uint32_t *lscan,
                                                       const SahAct *act, const SahSplit *split, SahCb *half) {
	const uint32_t p = blockIdx.x * BT + threadIdx.x;
	if (p >= n) return;
	const uint32_t o = owner[p];
	if (o == kNone || split[o].axis != kNone) return;
	const uint32_t side = left[p] ? 0u : 1u;
	const SahAct me = act[o];
	const uint32_t rank_l = lscan[p] - lscan[me.first];
	const uint32_t q = side == 0 ? me.first + rank_l : me.first + split[o].nl + (p - me.first - rank_l);
	const Box b = boxes[item_next[q]];
	SahCb *t = &half[2 * o + side];
	for (int d = 0; d < 3; d++) { atomicMin(&t->lo[d], f2o(b.lo[d])); atomicMax(&t->hi[d], f2o(b.hi[d])); }
}
__global__ __launch_bounds__(BT) void k_sah_write_halves(uint32_t A, const SahSplit *split, const SahCb *half, uint32_t next_base, PolarisBvhNode *out) {
	const uint32_t o = blockIdx.x * BT + threadIdx.x;
	if (o >= A || split[o].axis != kNone) return;
	for (int side = 0; side < 2; side++)
		for (int d = 0; d < 3; d++) { out[next_base + 2 * o + side].min[d] = o2f(half[2 * o + side].lo[d]); out[next_base + 2 * o + side].max[d] = o2f(half[2 * o + side].hi[d]); }
}

// the item order the leaves name; top-level leaves: the instance itself (leaves were written with the POSITION of their one item)
__global__ __launch_bounds__(BT) void k_sah_finish(uint32_t n, const uint32_t *item, uint32_t item_base, uint32_t *order, int instances, uint32_t node_base, uint32_t num_nodes, PolarisBvhNode *out) {
	const uint32_t p = blockIdx.x * BT + threadIdx.x;
	if (order && p < n) order[item_base + p] = item_base + item[p];
	if (instances && p < num_nodes) {
		PolarisBvhNode &nd = out[node_base + p];
		if (nd.ldata <= 0 && nd.rdata == 0) nd.ldata = -(int32_t)item[(uint32_t)(-nd.ldata)];
	}
}

struct SahScratch {
	uint32_t *item[2] = {nullptr, nullptr}, *owner[2] = {nullptr, nullptr}, *left = nullptr, *lscan = nullptr, *flags = nullptr, *nidx = nullptr, *counts = nullptr, *bounds = nullptr;
	SahBin *bins = nullptr;
	SahCb *cb = nullptr, *half = nullptr;
	SahAct *act[2] = {nullptr, nullptr};
	SahSplit *split = nullptr;
	void *temp = nullptr;
	size_t temp_bytes = 0;
	Box *boxes = nullptr;
};

// One tree over n items whose boxes are in S.boxes: nodes to out[node_base ...]; *emitted = the number of nodes written.
// max_depth: the levels the tree may have below its root (the depth budget above); raised to what n items need at the least.
int build_tree_sah(SahScratch &S, hipStream_t q, uint32_t n, uint32_t max_leaf, uint32_t node_base, uint32_t item_base, int instances, uint32_t max_depth,
                   PolarisBvhNode *d_out, uint32_t *d_order, uint32_t *emitted) {
	max_depth = std::max(max_depth, sah_levels_needed(n, max_leaf));
	BUILD_TRY(hipMemsetAsync(S.bounds, 0xFF, 3 * sizeof(uint32_t), q));
	BUILD_TRY(hipMemsetAsync(S.bounds + 3, 0x00, 3 * sizeof(uint32_t), q));
	hipLaunchKernelGGL(k_sah_init, dim3(grid(n)), dim3(BT), 0, q, n, S.item[0], S.owner[0]);
	hipLaunchKernelGGL(k_sah_root_bounds, dim3((n + 2047u) / 2048u), dim3(BT), 0, q, S.boxes, n, S.bounds);
	hipLaunchKernelGGL(k_sah_root_node, dim3(1), dim3(64), 0, q, S.bounds, n, max_leaf, node_base, instances ? 0u : item_base, instances, d_out, S.act[0], S.counts);
	uint32_t A = n <= max_leaf ? 0u : 1u, total = 1, n_big = n > kSmallMax ? 1u : 0u;
	int cur = 0;
	for (int level = 0; A > 0; level++) {
		if (level > 96) { g_build_error = "build_bvh: the SAH builder did not terminate (more than 96 levels)"; return POLARIS_E_DEVICE; }
		const uint32_t next_base = node_base + total;
		const uint32_t levels_below = max_depth > (uint32_t)level + 1u ? max_depth - (uint32_t)level - 1u : 0u; // below the children of this level's nodes
		if (n_big) { // nodes of more than kSmallMax items: centroid bounds, bins, split as three passes (atomics aggregated per tile in LDS)
			// (n_big <= n / 513: at most 2^26 / 513 x 48 = 6.3 M bin records, far inside uint32)
			hipLaunchKernelGGL(k_sah_clear_cb, dim3(grid(n_big)), dim3(BT), 0, q, n_big, S.cb);
			hipLaunchKernelGGL(k_sah_clear_bins, dim3(grid(n_big * 3u * kBins)), dim3(BT), 0, q, n_big * 3u * kBins, S.bins);
			hipLaunchKernelGGL(k_sah_cbounds_big, dim3((n + kTile - 1) / kTile), dim3(kBigBlock), 0, q, n, S.item[cur], S.owner[cur], S.act[cur], S.boxes, S.cb);
			hipLaunchKernelGGL(k_sah_bin_big, dim3((n + kTile - 1) / kTile), dim3(kBigBlock), 0, q, n, S.item[cur], S.owner[cur], S.act[cur], S.boxes, S.cb, S.bins);
			hipLaunchKernelGGL(k_sah_split_big, dim3(grid(A)), dim3(BT), 0, q, A, S.act[cur], S.cb, S.bins, max_leaf, next_base, instances ? 0u : item_base, instances, levels_below, d_out, S.split, S.flags, S.counts);
		}
		hipLaunchKernelGGL(k_sah_small, dim3((A + BT / 64 - 1) / (BT / 64)), dim3(BT), 0, q, A, S.act[cur], S.item[cur], S.boxes, max_leaf, next_base, instances ? 0u : item_base, instances, levels_below, d_out, S.split, S.flags, S.counts);
		size_t tb = S.temp_bytes;
		BUILD_TRY(rocprim::exclusive_scan(S.temp, tb, S.flags, S.nidx, 0u, (size_t)2 * A, rocprim::plus<uint32_t>(), q));
		hipLaunchKernelGGL(k_sah_next, dim3(grid(A)), dim3(BT), 0, q, A, S.act[cur], S.split, S.flags, S.nidx, next_base, S.act[cur ^ 1], S.counts);
		hipLaunchKernelGGL(k_sah_flag, dim3(grid(n)), dim3(BT), 0, q, n, S.item[cur], S.owner[cur], S.boxes, S.act[cur], S.split, S.left);
		tb = S.temp_bytes;
		BUILD_TRY(rocprim::exclusive_scan(S.temp, tb, S.left, S.lscan, 0u, (size_t)n, rocprim::plus<uint32_t>(), q));
		hipLaunchKernelGGL(k_sah_scatter, dim3(grid(n)), dim3(BT), 0, q, n, S.item[cur], S.owner[cur], S.left, S.lscan, S.act[cur], S.split, S.flags, S.nidx, S.item[cur ^ 1], S.owner[cur ^ 1]);
		uint32_t counts[3] = {0, 0, 0};
		BUILD_TRY(hipMemcpyAsync(counts, S.counts, sizeof counts, hipMemcpyDeviceToHost, q));
		BUILD_TRY(hipStreamSynchronize(q));
		BUILD_TRY(hipMemsetAsync(S.counts + 2, 0, sizeof(uint32_t), q));
		n_big = counts[2];
		if (counts[1]) { // some node was halved by position: fit its halves' boxes
			hipLaunchKernelGGL(k_sah_clear_cb, dim3(grid(2 * A)), dim3(BT), 0, q, 2 * A, S.half);
			hipLaunchKernelGGL(k_sah_fit_halves, dim3(grid(n)), dim3(BT), 0, q, n, S.item[cur ^ 1], S.boxes, S.owner[cur], S.left, S.lscan, S.act[cur], S.split, S.half);
			hipLaunchKernelGGL(k_sah_write_halves, dim3(grid(A)), dim3(BT), 0, q, A, S.split, S.half, next_base, d_out);
			BUILD_TRY(hipMemsetAsync(S.counts + 1, 0, sizeof(uint32_t), q));
		}
		total += 2 * A;
		A = counts[0];
		cur ^= 1;
	}
	hipLaunchKernelGGL(k_sah_finish, dim3(grid(std::max(n, total))), dim3(BT), 0, q, n, S.item[cur], item_base, d_order, instances, node_base, total, d_out);
	BUILD_TRY(hipGetLastError());
	*emitted = total;
	return POLARIS_OK;
}

} // namespace

extern "C" {

const char *polaris_hip_build_bvh_error(void) { return g_build_error.c_str(); }

int polaris_hip_build_bvh(int device, const PolarisBvhBuildInput *in, PolarisBvhNode *nodes, uint32_t nodes_capacity, uint32_t *num_nodes,
                          uint32_t *tri_order, uint32_t *mesh_root, double *device_ms) {
	g_build_error.clear();
	auto bad = [&](const char *m) { g_build_error = m; return POLARIS_E_BAD_ARGUMENT; };
	if (!in || !nodes || !num_nodes || !tri_order || !mesh_root) return bad("build_bvh: null argument");
	if (in->struct_size != sizeof(PolarisBvhBuildInput)) return bad("build_bvh: in->struct_size is not sizeof(PolarisBvhBuildInput): the caller was built against another ABI (polaris_hip.h, ABI history)");
	if (!in->vertices || in->num_triangles == 0 || in->num_triangles > (1u << 26)) return bad("build_bvh: no triangles (or more than 2^26)");
	if (!in->mesh_first_tri || !in->mesh_num_tris || in->num_meshes == 0) return bad("build_bvh: no meshes");
	if (!in->instance_boxes || !in->instance_mesh || in->num_instances == 0 || in->num_instances > (1u << 24)) return bad("build_bvh: no instances (or more than 2^24)");
	if (in->max_leaf_tris < 1 || in->max_leaf_tris > 15) return bad("build_bvh: max_leaf_tris must be 1..15");
	if (in->algorithm != POLARIS_BVH_SAH && in->algorithm != POLARIS_BVH_LBVH) return bad("build_bvh: algorithm must be POLARIS_BVH_SAH or POLARIS_BVH_LBVH");
	uint32_t biggest = in->num_instances;
	uint64_t covered = 0;
	for (uint32_t m = 0; m < in->num_meshes; m++) {
		const uint64_t f = in->mesh_first_tri[m], c = in->mesh_num_tris[m];
		if (c == 0 || f + c > in->num_triangles) return bad("build_bvh: a mesh's triangle range is empty or outside the vertex array");
		if (f != covered) return bad("build_bvh: the meshes' triangle ranges must tile [0, num_triangles) in order");
		covered += c;
		biggest = std::max<uint32_t>(biggest, (uint32_t)c);
	}
	if (covered != in->num_triangles) return bad("build_bvh: the meshes' triangle ranges must tile [0, num_triangles) in order");
	for (uint32_t i = 0; i < in->num_instances; i++)
		if (in->instance_mesh[i] >= in->num_meshes) return bad("build_bvh: an instance names a missing mesh");
	if ((uint64_t)nodes_capacity < 2ull * in->num_instances + 2ull * in->num_triangles) return bad("build_bvh: nodes_capacity must be at least 2 * (instances + triangles)");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_build_error = "build_bvh: no such HIP device"; return POLARIS_E_NO_DEVICE; }
	BUILD_TRY(hipSetDevice(device));
	hipStream_t q = nullptr;
	BUILD_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
	struct StreamGuard { hipStream_t q; ~StreamGuard() { (void)hipStreamDestroy(q); } } guard{q};
	const bool sah = in->algorithm != POLARIS_BVH_LBVH;
	Scratch S;
	SahScratch H;
	float4 *d_verts = nullptr;
	PolarisBvhNode *d_nodes = nullptr;
	uint32_t *d_order = nullptr;
	BUILD_TRY(S.get(&d_verts, (size_t)in->num_triangles * 3));
	BUILD_TRY(S.get(&d_nodes, nodes_capacity));
	BUILD_TRY(S.get(&d_order, in->num_triangles));
	BUILD_TRY(S.get(&S.boxes, biggest));
	if (sah) {
		H.boxes = S.boxes;
		const size_t amax = (size_t)biggest / 2 + 1; // active nodes of a level hold >= 2 items each and are disjoint
		for (int i = 0; i < 2; i++) { BUILD_TRY(S.get(&H.item[i], biggest)); BUILD_TRY(S.get(&H.owner[i], biggest)); BUILD_TRY(S.get(&H.act[i], amax)); }
		BUILD_TRY(S.get(&H.left, biggest)); BUILD_TRY(S.get(&H.lscan, biggest));
		BUILD_TRY(S.get(&H.flags, 2 * amax)); BUILD_TRY(S.get(&H.nidx, 2 * amax)); BUILD_TRY(S.get(&H.counts, 4)); BUILD_TRY(S.get(&H.bounds, 8));
		// bins and centroid bounds exist for the nodes of more than kSmallMax items only: disjoint ranges of > 512 items, so at most
		// biggest / 513 of them per level (round 5 sized them for EVERY active node: 672 bytes per triangle, 6.7 GB for a 10 M-triangle mesh)
		const size_t bmax = (size_t)biggest / (kSmallMax + 1) + 1;
		BUILD_TRY(S.get(&H.bins, bmax * 3 * kBins)); BUILD_TRY(S.get(&H.cb, bmax)); BUILD_TRY(S.get(&H.half, 2 * amax)); BUILD_TRY(S.get(&H.split, amax));
		size_t a = 0;
		BUILD_TRY(rocprim::exclusive_scan(nullptr, a, H.left, H.lscan, 0u, std::max<size_t>(biggest, 2 * amax), rocprim::plus<uint32_t>(), q));
		H.temp_bytes = a;
		uint8_t *t = nullptr;
		BUILD_TRY(S.get(&t, a));
		H.temp = t;
	} else {
		BUILD_TRY(S.get(&S.keys, biggest)); BUILD_TRY(S.get(&S.keys_alt, biggest));
		BUILD_TRY(S.get(&S.inner_box, biggest));
		BUILD_TRY(S.get(&S.left, biggest)); BUILD_TRY(S.get(&S.right, biggest));
		BUILD_TRY(S.get(&S.parent_inner, biggest)); BUILD_TRY(S.get(&S.parent_leaf, biggest));
		BUILD_TRY(S.get(&S.arrived, biggest)); BUILD_TRY(S.get(&S.need, biggest)); BUILD_TRY(S.get(&S.place, biggest));
		BUILD_TRY(S.get(&S.range, biggest)); BUILD_TRY(S.get(&S.bounds, 8));
		size_t a = 0, b = 0;
		BUILD_TRY(rocprim::radix_sort_keys(nullptr, a, S.keys_alt, S.keys, (size_t)biggest, 0u, 62u, q));
		BUILD_TRY(rocprim::exclusive_scan(nullptr, b, S.need, S.place, 0u, (size_t)std::max<uint32_t>(biggest, 2u), rocprim::plus<uint32_t>(), q));
		S.temp_bytes = std::max(a, b);
		uint8_t *t = nullptr;
		BUILD_TRY(S.get(&t, S.temp_bytes));
		S.temp = t;
	}
	BUILD_TRY(hipMemcpyAsync(d_verts, in->vertices, (size_t)in->num_triangles * 3 * sizeof(float4), hipMemcpyHostToDevice, q));
	BUILD_TRY(hipStreamSynchronize(q)); // (the upload is not part of the build time: a scene's vertices are on the device anyway)
	hipEvent_t e0, e1;
	BUILD_TRY(hipEventCreate(&e0));
	BUILD_TRY(hipEventCreate(&e1));
	struct EventGuard { hipEvent_t a, b; ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } eguard{e0, e1};
	BUILD_TRY(hipEventRecord(e0, q));
	// top-level tree first (node 0 is the scene's root): one instance per leaf (compiler.go:88-103)
	static_assert(sizeof(Box) == 24, "instance boxes arrive as 6 floats");
	BUILD_TRY(hipMemcpyAsync(S.boxes, in->instance_boxes, (size_t)in->num_instances * sizeof(Box), hipMemcpyHostToDevice, q));
	uint32_t total = 0, emitted = 0;
	// the SAH builder's depth budget: a ray's stack holds at most one entry per level of the top-level tree, the instance's exit marker and
	// one per level of the mesh's tree -- 32 entries (kernels.h; intersect.cl:4).  The top-level tree gets what a balanced tree over the
	// instances needs plus four levels of slack for the heuristic, the meshes' trees the rest of 30.
	const uint32_t top_depth = in->num_instances > 1 ? sah_levels_needed(in->num_instances, 1) + 4u : 0u;
	const uint32_t mesh_depth = top_depth < 30u ? 30u - top_depth : 0u;
	if (int rc = sah ? build_tree_sah(H, q, in->num_instances, 1, 0, 0, 1, top_depth, d_nodes, nullptr, &emitted) : build_tree(S, q, in->num_instances, 1, 0, 0, 1, d_nodes, nullptr, &emitted)) return rc;
	total += emitted;
	for (uint32_t m = 0; m < in->num_meshes; m++) {
		const uint32_t first = in->mesh_first_tri[m], n = in->mesh_num_tris[m];
		hipLaunchKernelGGL(k_tri_boxes, dim3(grid(n)), dim3(BT), 0, q, d_verts, first, n, S.boxes);
		mesh_root[m] = total;
		if (int rc = sah ? build_tree_sah(H, q, n, in->max_leaf_tris, total, first, 0, mesh_depth, d_nodes, d_order, &emitted) : build_tree(S, q, n, in->max_leaf_tris, total, first, 0, d_nodes, d_order, &emitted)) return rc;
		total += emitted;
	}
	BUILD_TRY(hipEventRecord(e1, q));
	BUILD_TRY(hipMemcpyAsync(nodes, d_nodes, (size_t)total * sizeof(PolarisBvhNode), hipMemcpyDeviceToHost, q));
	BUILD_TRY(hipMemcpyAsync(tri_order, d_order, (size_t)in->num_triangles * sizeof(uint32_t), hipMemcpyDeviceToHost, q));
	BUILD_TRY(hipStreamSynchronize(q));
	*num_nodes = total;
	if (device_ms) { float ms = 0.0f; (void)hipEventElapsedTime(&ms, e0, e1); *device_ms = ms; }
	return POLARIS_OK;
}

} // extern "C"
