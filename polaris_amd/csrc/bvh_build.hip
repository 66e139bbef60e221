// bvh_build.hip -- polaris_hip_build_bvh: the two-level BVH built ON THE DEVICE (the stretch).
//
// The reference's builder (asset/compiler/bvh/bvh_builder.go:100-308) scores ~1024 / (depth + 1) candidate planes per axis with
// one goroutine each, every one an O(n) pass over the node's items: fine for a few thousand triangles, minutes for a million.
// polaris_amd/host/scene_compiler.cpp restates it for the CPU.  This file is an ALTERNATIVE producer for the same arrays
// (PolarisBvhNode in the reference's encoding, optimized_scene.go:14-64) shaped for the GPU: a linear BVH (Morton order of the
// centroids on a grid whose cells stay near-cubic, one radix sort, the hierarchy of Karras 2012 built for all inner nodes at
// once, boxes fitted bottom-up), with subtrees of up to max_leaf_tris items collapsed into the reference's kind of leaf (first
// item, count).  (Round 4 also built PLOC -- bottom-up clustering of mutual nearest neighbours, Meister & Bittner 2018 -- on the
// same infrastructure: better on the Cornell box (14.5 vs 16.7 ms per frame), level elsewhere, and too DEEP for the 32-entry
// traversal stack on the 58 K-triangle ball; removed again, profiles/r04_bvh_build_lbvh_vs_ploc.json.)  One tree per mesh
// over its triangles, one over the instances' world boxes (one instance per leaf, compiler.go:88-103).
//
// It cannot reproduce the reference's tree (a different algorithm, and the reference breaks equal-score ties by goroutine
// arrival), and need not: the traversal is correct for ANY tree whose boxes contain their items, and parity is
// defined on the uploaded arrays.  What is checked instead (tests/test_gpu_bvh_build.py): the tree is valid under
// scene_layout.h's rules, every item sits in exactly one leaf, every box contains what is below it, and the HIP trace of a scene
// on the tree built here equals the CPU oracle's trace of the same arrays bit for bit.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

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
	BUILD_TRY(hipcub::DeviceRadixSort::SortKeys(S.temp, tb, S.keys_alt, S.keys, (int)n, 0, 62, q));
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
	BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(S.temp, tb, S.need, S.place, (int)(n - 1), q));
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

} // namespace

extern "C" {

const char *polaris_hip_build_bvh_error(void) { return g_build_error.c_str(); }

int polaris_hip_build_bvh(int device, const PolarisBvhBuildInput *in, PolarisBvhNode *nodes, uint32_t nodes_capacity, uint32_t *num_nodes,
                          uint32_t *tri_order, uint32_t *mesh_root, double *device_ms) {
	g_build_error.clear();
	auto bad = [&](const char *m) { g_build_error = m; return POLARIS_E_BAD_ARGUMENT; };
	if (!in || !nodes || !num_nodes || !tri_order || !mesh_root) return bad("build_bvh: null argument");
	if (!in->vertices || in->num_triangles == 0 || in->num_triangles > (1u << 26)) return bad("build_bvh: no triangles (or more than 2^26)");
	if (!in->mesh_first_tri || !in->mesh_num_tris || in->num_meshes == 0) return bad("build_bvh: no meshes");
	if (!in->instance_boxes || !in->instance_mesh || in->num_instances == 0 || in->num_instances > (1u << 24)) return bad("build_bvh: no instances (or more than 2^24)");
	if (in->max_leaf_tris < 1 || in->max_leaf_tris > 15) return bad("build_bvh: max_leaf_tris must be 1..15");
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
	Scratch S;
	float4 *d_verts = nullptr;
	PolarisBvhNode *d_nodes = nullptr;
	uint32_t *d_order = nullptr;
	BUILD_TRY(S.get(&d_verts, (size_t)in->num_triangles * 3));
	BUILD_TRY(S.get(&d_nodes, nodes_capacity));
	BUILD_TRY(S.get(&d_order, in->num_triangles));
	BUILD_TRY(S.get(&S.keys, biggest)); BUILD_TRY(S.get(&S.keys_alt, biggest));
	BUILD_TRY(S.get(&S.boxes, biggest)); BUILD_TRY(S.get(&S.inner_box, biggest));
	BUILD_TRY(S.get(&S.left, biggest)); BUILD_TRY(S.get(&S.right, biggest));
	BUILD_TRY(S.get(&S.parent_inner, biggest)); BUILD_TRY(S.get(&S.parent_leaf, biggest));
	BUILD_TRY(S.get(&S.arrived, biggest)); BUILD_TRY(S.get(&S.need, biggest)); BUILD_TRY(S.get(&S.place, biggest));
	BUILD_TRY(S.get(&S.range, biggest)); BUILD_TRY(S.get(&S.bounds, 8));
	{
		size_t a = 0, b = 0;
		BUILD_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, a, S.keys_alt, S.keys, (int)biggest, 0, 62, q));
		BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b, S.need, S.place, (int)biggest, q));
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
	if (int rc = build_tree(S, q, in->num_instances, 1, 0, 0, 1, d_nodes, nullptr, &emitted)) return rc;
	total += emitted;
	for (uint32_t m = 0; m < in->num_meshes; m++) {
		const uint32_t first = in->mesh_first_tri[m], n = in->mesh_num_tris[m];
		hipLaunchKernelGGL(k_tri_boxes, dim3(grid(n)), dim3(BT), 0, q, d_verts, first, n, S.boxes);
		mesh_root[m] = total;
		if (int rc = build_tree(S, q, n, in->max_leaf_tris, total, first, 0, d_nodes, d_order, &emitted)) return rc;
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
