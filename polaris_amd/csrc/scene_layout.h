// scene_layout.h -- host-side validation and traversal re-layout of the uploaded scene.
//
// Input is the reference's compiled scene (asset/scene/optimized_scene.go:167-190), taken
// as-is from the caller.  Nothing the GPU will dereference is trusted: every index is range
// checked here (an out-of-bounds fetch on the device can reset every GPU of the host), and the
// traversal stack depth the scene needs is computed exactly and checked against the kernel's
// LDS stack.
//
// Re-layout (HBM data layout):
//   * PairNode[node]: for every INNER node, the boxes of BOTH children side by side (64 B, one
//     fetch per traversal step instead of the reference's two dependent 32 B fetches,
//     kernels/intersect.cl:296-298) plus a tagged reference per child:
//         ref >= 0  -> inner node index          ref < 0 -> ~code of a leaf:
//         code = first triangle slot << 4 | triangle count (1..15): no fetch needed to start testing;
//         code = mesh instance << 4 | 0: a top-level leaf -- the instance id is in the reference itself (until round 4 it was
//                looked up: one dependent fetch per instance a ray enters);
//         code = 1 << 30 | node index << 4 | 0: a leaf of more than 15 triangles, LeafInfo[node] has first / count
//     and a cull factor per child (1.001, or +inf when the box does not bound its subtree).
//   * LeafInfo[node] = (ldata, rdata) of a leaf: 8 B fetch when a leaf is popped.
//   * Tri[slot] = {v0 | rank, e01 = v1 - v0 | scene triangle index, e02 = v2 - v0}: the two edge subtractions of
//     Moeller-Trumbore are hoisted to upload (one IEEE subtraction each, so bit-identical to
//     intersect.cl:253-254).  rank = position of the triangle in the reference's left-first
//     depth-first traversal order of its mesh BVH; the closest-hit kernel uses it to break exact
//     ties the way "first tested wins" does in the reference (intersect.cl:281 strict <).
//   * Inst[i] = rows of the inverse 3x4 matrix (so mul4x1 / mul3x1 of util/transform.cl:9-26
//     read 3 float4), tagged root reference, and the instance's rank in top-level DFS order.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <utility>
#include <vector>

#include "polaris_types.h"

namespace pol {

struct PairNodeH { float lo0[3]; int32_t ref0; float hi0[3]; int32_t pad0; float lo1[3]; int32_t ref1; float hi1[3]; int32_t pad1; };
struct LeafInfoH { int32_t ldata, rdata; };
struct TriH { float v0[3]; uint32_t rank; float e1[3]; uint32_t orig; float e2[3]; uint32_t pad2; };
struct InstH { float r0[4], r1[4], r2[4]; int32_t root_ref; uint32_t rank; uint32_t pad[2]; };
static_assert(sizeof(PairNodeH) == 64 && sizeof(TriH) == 48 && sizeof(InstH) == 64, "layout");
// rank << 19 | shading class << 11 | scene triangle: one word per triangle slot for scenes of < 2 047 triangles (kernels.h tiny_meta_tri)
inline uint32_t tiny_meta_word(uint32_t rank, uint32_t orig) { return rank << 19 | (orig >> 24 & 0xFFu) << 11 | (orig & 0x7FFu); }

constexpr float kCullMargin = 1.001f; // a subtree is skipped when its box starts beyond kCullMargin x the best hit distance
constexpr float kSplitCost = 1.0f;  // cost of one added pair-of-boxes step, in triangle tests (leaf subdivision)
constexpr float kMaxCoordinate = 1099511627776.0f; // 2^40: largest vertex coordinate a scene may hold (build_layout, magnitudes)
constexpr float kMaxMatrixEntry = 1073741824.0f;   // 2^30: largest entry of an instance matrix
constexpr int kTraversalStack = 32; // entries per ray, == BVH_MAX_STACK_SIZE (intersect.cl:4)

constexpr uint32_t kBigLeafFlag = 1u << 30; // leaf code: count nibble 0 and this bit = "look the triangle range up in LeafInfo[node]"

struct SceneLayout {
	std::vector<PairNodeH> pairs;   // inner nodes only, in breadth-first order (the first kLdsTopNodes are staged in LDS)
	std::vector<LeafInfoH> leaves;  // indexed by node id (only leaves meaningful; the kernels read it for leaves of > 15 triangles only)
	uint32_t big_leaves = 0;        // reachable leaves of more than 15 triangles (the tiny-scene traversal mode excludes them)
	std::vector<TriH> tris;
	std::vector<InstH> insts;
	int32_t root_ref = 0;
	int max_stack = 0;
	uint32_t unbounded_boxes = 0;   // inner nodes with a child whose box does not contain its subtree (never culled by distance)
	// A triangle record's `orig` word = scene triangle index | shading class << tri_bits (24 when the scene has at most 2^24
	// triangles, else 31 = no class).  The traversal kernels hand the word through to the hit record untouched; k_shade
	// sorts a workgroup's rays by the class and masks it off.
	uint32_t tri_bits = 31;
};

// Shading class of every material node = which BxDF leaves and texture operators the tree rooted at it can reach.  k_shade
// groups the rays of a workgroup by the class of the triangle's material root before shading them (rays of one class run
// the same code, so a wave is no longer the union of five BxDF paths).  A sort key, nothing else: results do not depend on
// it.  Classes 1..15 (0 is "the ray missed"); scenes with more distinct reach sets share class 15.
inline void shading_classes(const PolarisSceneView &sc, std::vector<uint8_t> &cls) {
	const uint32_t n = sc.num_material_nodes;
	std::vector<uint32_t> reach(n, 0);
	std::vector<uint8_t> state(n, 0); // 0 new, 1 on the walk, 2 done
	auto leaf_bit = [](uint32_t type) -> uint32_t {
		switch (type) {
		case POLARIS_BXDF_EMISSIVE: return 1u;
		case POLARIS_BXDF_DIFFUSE: return 2u;
		case POLARIS_BXDF_CONDUCTOR: return 4u;
		case POLARIS_BXDF_ROUGH_CONDUCTOR: return 8u;
		case POLARIS_BXDF_DIELECTRIC: return 16u;
		case POLARIS_BXDF_ROUGH_DIELECTRIC: return 32u;
		default: return 64u; // invalid: the path ends
		}
	};
	for (uint32_t root = 0; root < n; root++) { // iterative post-order walk (trees may be deep, and a malformed one may loop)
		if (state[root] == 2) continue;
		std::vector<std::pair<uint32_t, int>> stack{{root, 0}};
		state[root] = 1;
		while (!stack.empty()) {
			const uint32_t i = stack.back().first;
			const int step = stack.back().second++;
			const PolarisMaterialNode &m = sc.material_nodes[i];
			if (m.type < POLARIS_MAT_OP_MIX) { reach[i] = leaf_bit(m.type); state[i] = 2; stack.pop_back(); continue; }
			const bool two = m.type == POLARIS_MAT_OP_MIX || m.type == POLARIS_MAT_OP_MIX_MAP;
			const uint32_t kids[2] = {m.left_child, two ? (uint32_t)m.right_child : m.left_child};
			if (step < (two ? 2 : 1)) {
				const uint32_t c = kids[step];
				if (c < n && state[c] == 0) { state[c] = 1; stack.push_back({c, 0}); }
				continue;
			}
			uint32_t r = m.type == POLARIS_MAT_OP_MIX ? 0u : (m.type == POLARIS_MAT_OP_DISPERSE ? 256u : 128u); // textured operators / dispersion
			for (int k = 0; k < (two ? 2 : 1); k++) r |= kids[k] < n && state[kids[k]] == 2 ? reach[kids[k]] : 64u; // a child still on the walk = a cycle
			reach[i] = r;
			state[i] = 2;
			stack.pop_back();
		}
	}
	std::vector<uint32_t> distinct(reach.begin(), reach.end());
	std::sort(distinct.begin(), distinct.end());
	distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
	cls.resize(n);
	for (uint32_t i = 0; i < n; i++) {
		const size_t k = (size_t)(std::lower_bound(distinct.begin(), distinct.end(), reach[i]) - distinct.begin());
		cls[i] = (uint8_t)std::min<size_t>(1 + k, 15);
	}
}


inline bool is_leaf(const PolarisBvhNode &n) { return n.ldata <= 0; }

// Returns "" on success, otherwise a description of the first inconsistency.
//
// max_leaf_tris > 0: triangle leaves holding more triangles than that are subdivided at upload
// (surface-area-heuristic splits of the leaf's own triangles, where a split pays).  The reference's boxes, down to and including its
// leaves, are kept bit for bit, so a triangle is still only ever tested when the reference's
// traversal would have reached its leaf; the ADDED boxes only cull inside such a leaf and are
// inflated by 2^-13 of the scene's extent S in the mesh's object space.  A slab distance t carries a
// rounding error of about |t| * 2^-23, the padding moves the slab planes by >= 2^-13 * S in t, so a
// hit at distance t < 2^9 * S (a ray that starts within ~500 scene extents) is never lost -- they can
// never hide a triangle the reference would have hit.  Results are unchanged (parity tests run
// with and without); only the number of Moeller-Trumbore tests per ray drops.
inline std::string build_layout(const PolarisSceneView &sc, SceneLayout &out, int max_leaf_tris = 0) {
	const uint32_t NN = sc.num_bvh_nodes, NT = sc.num_triangles, NI = sc.num_mesh_instances;
	if (!sc.bvh_nodes || NN == 0) return "scene has no BVH nodes";
	if (!sc.mesh_instances || NI == 0) return "scene has no mesh instances";
	if (NT == 0 || !sc.vertices || !sc.normals || !sc.uvs || !sc.material_index) return "scene has no triangles";
	if (NT > (1u << 30)) return "too many triangles";
	if (NN >= (1u << 26)) return "too many BVH nodes";
	if (NI >= (1u << 26)) return "too many mesh instances";
	if (sc.num_material_nodes == 0 || !sc.material_nodes) return "scene has no material nodes";
	if (sc.num_emissives && !sc.emissives) return "emissive list pointer is null";
	if (sc.num_textures && (!sc.texture_meta || !sc.texture_data)) return "texture pointers are null";

	// ---- magnitudes ------------------------------------------------------------------------
	// The triangle tests take 1 / det from v_rcp_f32 + one Newton step, which equals the correctly rounded quotient for
	// 2^-126 <= |det| < 2^126 (kernels.h, rcp_det).  det = e1 . (d x e2) with edges of at most 2 kMaxCoordinate and an instance
	// space direction of at most 3 sqrt(3) kMaxMatrixEntry: < 2^118 with the bounds below, which no meaningful scene approaches.
	for (size_t i = 0; i < (size_t)NT * 3; i++) {
		const float *v = sc.vertices + 4 * i;
		for (int k = 0; k < 3; k++)
			if (!(std::fabs(v[k]) <= kMaxCoordinate)) // (also rejects NaN)
				return "triangle " + std::to_string(i / 3) + ": vertex coordinate not finite or beyond 2^40";
	}
	for (uint32_t i = 0; i < NI; i++)
		for (int k = 0; k < 16; k++)
			if (!(std::fabs(sc.mesh_instances[i].inv_transform[k]) <= kMaxMatrixEntry))
				return "mesh instance " + std::to_string(i) + ": matrix entry not finite or beyond 2^30";

	// ---- materials / textures / emissives ------------------------------------------------
	for (uint32_t t = 0; t < sc.num_textures; t++) {
		const PolarisTextureMetadata &m = sc.texture_meta[t];
		if (m.format > POLARIS_TEX_RGBA32F) return "texture " + std::to_string(t) + ": unknown format";
		if (m.width == 0 || m.height == 0) return "texture " + std::to_string(t) + ": empty";
		const uint64_t bpp = m.format == POLARIS_TEX_L8 ? 1 : (m.format == POLARIS_TEX_RGBA32F ? 16 : 4);
		if ((uint64_t)m.data_offset + bpp * m.width * m.height > sc.texture_data_bytes)
			return "texture " + std::to_string(t) + ": data outside the texture blob";
		// texels are fetched as whole words (the reference reads them through uchar4 / float / float4 pointers, texture_sampler.cl:42-91)
		if (m.format != POLARIS_TEX_L8 && (m.data_offset & 3u))
			return "texture " + std::to_string(t) + ": texel data not dword aligned";
	}
	auto tex_ok = [&](int32_t t) { return t == -1 || (t >= 0 && (uint32_t)t < sc.num_textures); };
	for (uint32_t i = 0; i < sc.num_material_nodes; i++) {
		const PolarisMaterialNode &m = sc.material_nodes[i];
		const std::string who = "material node " + std::to_string(i);
		if (m.type >= POLARIS_MAT_OP_MIX) {
			if (m.type > POLARIS_MAT_OP_DISPERSE) return who + ": unknown operator";
			if (m.left_child >= sc.num_material_nodes) return who + ": left child out of range";
			if ((m.type == POLARIS_MAT_OP_MIX || m.type == POLARIS_MAT_OP_MIX_MAP) &&
			    (m.right_child < 0 || (uint32_t)m.right_child >= sc.num_material_nodes))
				return who + ": right child out of range";
			if (m.type == POLARIS_MAT_OP_MIX_MAP || m.type == POLARIS_MAT_OP_BUMP_MAP || m.type == POLARIS_MAT_OP_NORMAL_MAP) {
				if (m.tex < 0 || (uint32_t)m.tex >= sc.num_textures) return who + ": operator texture out of range";
			} else if (!tex_ok(m.tex)) // mix / disperse: unused by the operator, but a background or light reference may land on any node
				return who + ": texture out of range";
		} else {
			if (!tex_ok(m.tex) || !tex_ok(m.roughness_tex)) return who + ": texture out of range";
			if ((m.type == POLARIS_BXDF_DIELECTRIC || m.type == POLARIS_BXDF_ROUGH_DIELECTRIC) && !tex_ok(m.right_child))
				return who + ": transmittance texture out of range";
		}
	}
	for (uint32_t t = 0; t < NT; t++)
		if (sc.material_index[t] >= sc.num_material_nodes) return "triangle " + std::to_string(t) + ": material root out of range";
	for (uint32_t e = 0; e < sc.num_emissives; e++) {
		const PolarisEmissive &em = sc.emissives[e];
		if (em.type > POLARIS_EMISSIVE_ENVIRONMENT) return "emissive " + std::to_string(e) + ": unknown type";
		if (em.mat_node_index >= sc.num_material_nodes) return "emissive " + std::to_string(e) + ": material node out of range";
		if (em.type == POLARIS_EMISSIVE_AREA && em.tri_index >= NT) return "emissive " + std::to_string(e) + ": triangle out of range";
	}
	if (sc.scene_diffuse_mat_index != -1 && (sc.scene_diffuse_mat_index < 0 || (uint32_t)sc.scene_diffuse_mat_index >= sc.num_material_nodes))
		return "scene diffuse material index out of range";

	// ---- BVH: structure, ranks, stack depth ----------------------------------------------
	// Pass 1 walks the caller's tree: validation, DFS ranks, which leaves are reachable.
	// Pass 2 walks the tree the kernels use (the same one, or the one with subdivided leaves)
	// and emits pair/leaf records and the exact stack depth.
	out.insts.assign(NI, InstH{});
	std::vector<PolarisBvhNode> nodes(sc.bvh_nodes, sc.bvh_nodes + NN);
	uint32_t n_nodes = NN, n_slots = NT;
	std::vector<uint32_t> tri_rank(NT, 0xFFFFFFFFu);
	std::vector<uint32_t> slot_src;          // triangle slot of the kernels -> scene triangle (pass 2)
	std::vector<int32_t> leaf_root(NN, -1);  // reachable triangle leaf -> root of its mesh BVH
	std::vector<uint32_t> seen;              // generation stamps: seen[n] == seen_gen <=> reached in the current walk
	uint32_t seen_gen = 0;
	// tagged child reference: inner node -> its index; leaf -> ~code, see the header comment
	auto ref_of = [&](int32_t idx) -> int32_t {
		const PolarisBvhNode &n = nodes[idx];
		if (!is_leaf(n)) return idx;
		const uint32_t first = (uint32_t)(-(int64_t)n.ldata);
		if (n.rdata == 0) return ~(int32_t)(first << 4);                     // top-level leaf: first = the mesh instance (< 2^26: checked below)
		if (n.rdata >= 1 && n.rdata <= 15 && first < (1u << 26) - 1u) return ~(int32_t)((first << 4) | (uint32_t)n.rdata);
		return ~(int32_t)(kBigLeafFlag | ((uint32_t)idx << 4));             // (node ids < 2^26: checked below)
	};

	// iterative left-first DFS from `root`; level = 0 top tree, 1 bottom tree.  `need` tracks the
	// number of stack entries a traversal can hold at a node (one pending sibling per level).
	struct Item { int32_t node; int depth; };
	uint32_t next_inst_rank = 0;
	int top_max = 0, pass = 1;
	std::vector<int> inst_entry_depth(NI, 0);
	std::string err;
	auto walk = [&](int32_t root, int level, uint32_t &rank_counter, int &max_depth) -> bool {
		std::vector<Item> st;
		st.push_back({root, 0});
		while (!st.empty()) {
			Item it = st.back();
			st.pop_back();
			if (it.node < 0 || (uint32_t)it.node >= n_nodes) { err = "BVH child index out of range"; return false; }
			// a node reached twice within one tree (a DAG) makes every walk -- this one and the GPU's -- exponential in the depth
			if (seen[it.node] == seen_gen) { err = "BVH node " + std::to_string(it.node) + " reachable twice"; return false; }
			seen[it.node] = seen_gen;
			if (it.depth > max_depth) max_depth = it.depth;
			if (it.depth > kTraversalStack) { err = pass == 2 ? "@retry-without-subdivision" : "BVH too deep for the traversal stack"; return false; }
			const PolarisBvhNode &n = nodes[it.node];
			if (is_leaf(n)) {
				if (pass == 2) out.leaves[it.node] = {n.ldata, n.rdata};
				if (n.rdata == 0) {
					if (level != 0) { err = "instance leaf inside a mesh BVH (node " + std::to_string(it.node) + ")"; return false; }
					const uint32_t inst = (uint32_t)(-(int64_t)n.ldata);
					if (inst >= NI) { err = "top-level leaf points at a missing mesh instance"; return false; }
					out.insts[inst].rank = next_inst_rank++;
					inst_entry_depth[inst] = it.depth;
				} else {
					if (level != 1) { err = "triangle leaf in the top-level BVH (node " + std::to_string(it.node) + ")"; return false; }
					if (n.rdata < 0) { err = "negative triangle count"; return false; }
					const uint64_t first = (uint64_t)(-(int64_t)n.ldata);
					if (first + (uint64_t)n.rdata > n_slots) { err = "leaf triangle range out of bounds"; return false; }
					if (pass == 2 && n.rdata > 15) out.big_leaves++;
					if (pass == 1) {
						if (leaf_root[it.node] < 0) leaf_root[it.node] = root;
						for (uint64_t t = first; t < first + (uint64_t)n.rdata; t++)
							if (tri_rank[t] == 0xFFFFFFFFu) tri_rank[t] = rank_counter++;
					}
				}
			} else {
				if (n.rdata <= 0) { err = "inner node " + std::to_string(it.node) + " has a non-positive right child"; return false; }
				if ((uint32_t)n.ldata >= n_nodes || (uint32_t)n.rdata >= n_nodes) { err = "BVH child index out of range"; return false; }
				if (pass == 2) {
					PairNodeH &p = out.pairs[it.node];
					const PolarisBvhNode &l = nodes[n.ldata], &r = nodes[n.rdata];
					memcpy(p.lo0, l.min, 12); memcpy(p.hi0, l.max, 12); p.ref0 = ref_of(n.ldata);
					memcpy(p.lo1, r.min, 12); memcpy(p.hi1, r.max, 12); p.ref1 = ref_of(n.rdata);
				}
				// right pushed first so the left subtree is visited first (reference order)
				st.push_back({n.rdata, it.depth + 1});
				st.push_back({n.ldata, it.depth + 1});
			}
		}
		return true;
	};

	std::vector<int> root_depth;
	int need = 0;
	auto walk_scene = [&]() -> std::string {
		seen.assign(n_nodes, 0);
		seen_gen = 1;
		next_inst_rank = 0;
		top_max = 0;
		uint32_t dummy = 0;
		if (!walk(0, 0, dummy, top_max)) return err;
		// bottom trees: one walk per distinct root (meshes are shared by instances)
		root_depth.assign(n_nodes, -1);
		need = top_max;
		for (uint32_t i = 0; i < NI; i++) {
			const PolarisMeshInstance &mi = sc.mesh_instances[i];
			if (mi.bvh_root >= NN) return "mesh instance " + std::to_string(i) + ": bvh root out of range";
			if (mi.bvh_root == 0) return "mesh instance " + std::to_string(i) + ": bvh root is the scene root";
			if (root_depth[mi.bvh_root] < 0) {
				int md = 0;
				uint32_t rank = 0;
				seen_gen++; // distinct mesh roots may share subtrees; one walk may not reach a node twice
				if (!walk((int32_t)mi.bvh_root, 1, rank, md)) return err;
				root_depth[mi.bvh_root] = md;
			}
			// stack use below an instance: pending top-level siblings + the exit marker + bottom depth
			const int use = inst_entry_depth[i] + 1 + root_depth[mi.bvh_root];
			if (use > need) need = use;
		}
		return "";
	};
	{
		const std::string e = walk_scene();
		if (!e.empty()) return e;
	}
	const int need_reference = need;

	// ---- optional subdivision of big leaves ---------------------------------------------------
	bool subdivided = false;
	if (max_leaf_tris > 0) {
		// extent of the scene in each mesh's object space -> padding of the added boxes
		std::vector<float> root_pad(NN, 0.0f);
		const PolarisBvhNode &world = sc.bvh_nodes[0];
		for (uint32_t i = 0; i < NI; i++) {
			const PolarisMeshInstance &mi = sc.mesh_instances[i];
			const float *m = mi.inv_transform; // column major: m[4*c + r]
			float ext = 0.0f;
			for (int corner = 0; corner < 8; corner++) {
				const float w[3] = {corner & 1 ? world.max[0] : world.min[0], corner & 2 ? world.max[1] : world.min[1],
				                    corner & 4 ? world.max[2] : world.min[2]};
				for (int r = 0; r < 3; r++) {
					const float v = std::fabs(m[r] * w[0] + m[4 + r] * w[1] + m[8 + r] * w[2] + m[12 + r]);
					if (!(v <= ext)) ext = v; // also takes NaN
				}
			}
			const PolarisBvhNode &rb = sc.bvh_nodes[mi.bvh_root];
			for (int k = 0; k < 3; k++) { ext = std::fmax(ext, std::fabs(rb.min[k])); ext = std::fmax(ext, std::fabs(rb.max[k])); }
			float pad = ext * (1.0f / 8192.0f);
			if (!(pad <= 1e30f)) pad = 1e30f;
			if (pad > root_pad[mi.bvh_root]) root_pad[mi.bvh_root] = pad;
		}
		struct Range { int32_t node; uint32_t lo, hi; };
		std::vector<Range> work;
		for (uint32_t idx = 0; idx < NN; idx++) {
			if (leaf_root[idx] < 0) continue;
			const PolarisBvhNode src = sc.bvh_nodes[idx];
			const uint32_t first = (uint32_t)(-(int64_t)src.ldata), count = (uint32_t)src.rdata;
			const uint32_t lo = (uint32_t)slot_src.size();
			if (slot_src.size() + (size_t)count > 4ull * NT + 64) return "triangle leaves overlap";
			for (uint32_t t = first; t < first + count; t++) slot_src.push_back(t);
			nodes[idx].ldata = -(int32_t)lo;
			if (count <= (uint32_t)max_leaf_tris) continue;
			subdivided = true;
			const float pad = root_pad[leaf_root[idx]];
			auto centroid = [&](uint32_t tri, int axis) {
				const float *v = sc.vertices + 4 * (size_t)(3 * tri);
				return v[axis] + v[4 + axis] + v[8 + axis];
			};
			work.clear();
			work.push_back({(int32_t)idx, lo, lo + count});
			while (!work.empty()) {
				const Range rg = work.back();
				work.pop_back();
				const uint32_t cnt = rg.hi - rg.lo;
				float bmin[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bmax[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
				for (uint32_t s = rg.lo; s < rg.hi; s++) {
					const float *v = sc.vertices + 4 * (size_t)(3 * slot_src[s]);
					for (int k = 0; k < 3; k++) {
						bmin[k] = std::fmin(bmin[k], std::fmin(v[k], std::fmin(v[4 + k], v[8 + k])));
						bmax[k] = std::fmax(bmax[k], std::fmax(v[k], std::fmax(v[4 + k], v[8 + k])));
					}
				}
				PolarisBvhNode &nd = nodes[rg.node];
				if (rg.node != (int32_t)idx) // the reference's own leaf keeps the reference's box
					for (int k = 0; k < 3; k++) { nd.min[k] = bmin[k] - pad; nd.max[k] = bmax[k] + pad; }
				// surface-area sweep over the three axes (triangles sorted by centroid): split where
				// area(L)*|L| + area(R)*|R| is smallest, and only if that beats testing all of them
				int best_axis = -1;
				uint32_t best_at = 0;
				float best_cost = 0.0f;
				if (cnt > (uint32_t)max_leaf_tris) {
					auto area = [](const float *lo, const float *hi) {
						const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
						return dx * dy + dy * dz + dz * dx;
					};
					best_cost = (float)cnt * area(bmin, bmax) - kSplitCost * area(bmin, bmax);
					std::vector<uint32_t> order(slot_src.begin() + rg.lo, slot_src.begin() + rg.hi);
					std::vector<float> right_area(cnt + 1, 0.0f);
					for (int axis = 0; axis < 3; axis++) {
						std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return centroid(a, axis) < centroid(b, axis); });
						float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
						auto grow = [&](uint32_t tri) {
							const float *v = sc.vertices + 4 * (size_t)(3 * tri);
							for (int k = 0; k < 3; k++) {
								lo[k] = std::fmin(lo[k], std::fmin(v[k], std::fmin(v[4 + k], v[8 + k])));
								hi[k] = std::fmax(hi[k], std::fmax(v[k], std::fmax(v[4 + k], v[8 + k])));
							}
						};
						for (uint32_t i = cnt; i-- > 1;) { grow(order[i]); right_area[i] = area(lo, hi); }
						for (int k = 0; k < 3; k++) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; }
						for (uint32_t i = 1; i < cnt; i++) {
							grow(order[i - 1]);
							const float c = area(lo, hi) * (float)i + right_area[i] * (float)(cnt - i);
							if (c < best_cost) { best_cost = c; best_axis = axis; best_at = i; }
						}
					}
				}
				if (best_axis < 0) {
					nd.ldata = -(int32_t)rg.lo;
					nd.rdata = (int32_t)cnt;
					continue;
				}
				std::stable_sort(slot_src.begin() + rg.lo, slot_src.begin() + rg.hi,
				                 [&](uint32_t a, uint32_t b) { return centroid(a, best_axis) < centroid(b, best_axis); });
				const uint32_t mid = rg.lo + best_at;
				const int32_t l = (int32_t)nodes.size();
				nodes.push_back(PolarisBvhNode{});
				nodes.push_back(PolarisBvhNode{});
				nodes[rg.node].ldata = l;
				nodes[rg.node].rdata = l + 1;
				work.push_back({l, rg.lo, mid});
				work.push_back({l + 1, mid, rg.hi});
			}
		}
		if (nodes.size() >= (1ull << 26)) return "too many BVH nodes";
	}
	if (!subdivided) { // nothing to do: keep the caller's tree and triangle order
		nodes.assign(sc.bvh_nodes, sc.bvh_nodes + NN);
		slot_src.resize(NT);
		for (uint32_t t = 0; t < NT; t++) slot_src[t] = t;
	}
	n_nodes = (uint32_t)nodes.size();
	n_slots = (uint32_t)slot_src.size();
	out.pairs.assign(n_nodes, PairNodeH{});
	out.leaves.assign(n_nodes, LeafInfoH{0, 0});
	pass = 2;
	{
		const std::string e = walk_scene();
		if (!e.empty()) return e;
	}
	out.root_ref = ref_of(0);
	for (uint32_t i = 0; i < NI; i++) {
		const PolarisMeshInstance &mi = sc.mesh_instances[i];
		InstH &d = out.insts[i];
		const float *m = mi.inv_transform; // column major: m[4*c + r]
		for (int c = 0; c < 4; c++) { d.r0[c] = m[4 * c + 0]; d.r1[c] = m[4 * c + 1]; d.r2[c] = m[4 * c + 2]; }
		d.root_ref = ref_of((int32_t)mi.bvh_root);
	}
	out.max_stack = need + 1;
	if (need_reference + 1 > kTraversalStack)
		return "BVH needs a traversal stack of " + std::to_string(need_reference + 1) + " entries; the kernel (like the reference, "
		       "intersect.cl:4) has " + std::to_string(kTraversalStack);
	if (out.max_stack > kTraversalStack) return "@retry-without-subdivision";

	// ---- which boxes really bound their contents? ----------------------------------------------
	// The closest-hit kernels skip a child whose box starts beyond the best hit so far.  That is
	// only the reference's answer if the box bounds everything below it -- and the reference's own
	// scene reader breaks this: an `instance` with a rotation or a scale gets the mesh box moved by
	// the translation alone (asset/scene/reader/wavefront.go:514-519), and the top-level boxes are
	// unions of those.  The reference still finds hits inside such an instance whenever the ray
	// touches the (wrong) box, however far away the box starts, because it never culls by distance
	// (intersect.cl:309 compares with the ray's max distance only).  So: the real extent of every
	// subtree is computed here, and a child whose box does not contain it gets an infinite cull
	// limit (PairNode.hi?.w is the factor applied to the best distance).
	{
		struct Box { float lo[3], hi[3]; };
		const Box empty = {{3.0e38f, 3.0e38f, 3.0e38f}, {-3.0e38f, -3.0e38f, -3.0e38f}};
		std::vector<Box> content(n_nodes, empty);
		std::vector<uint8_t> done(n_nodes, 0);
		auto grow = [](Box &b, const float *p) { for (int k = 0; k < 3; k++) { b.lo[k] = std::fmin(b.lo[k], p[k]); b.hi[k] = std::fmax(b.hi[k], p[k]); } };
		auto merge = [&](Box &b, const Box &c) { grow(b, c.lo); grow(b, c.hi); };
		// forward (object -> world) matrices, in double: the stored ones are the inverses
		auto invert = [](const float *a, double *o) {
			double m[4][8];
			for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { m[r][c] = a[4 * c + r]; m[r][4 + c] = r == c; }
			for (int c = 0; c < 4; c++) {
				int piv = c;
				for (int r = c + 1; r < 4; r++) if (std::fabs(m[r][c]) > std::fabs(m[piv][c])) piv = r;
				if (!(std::fabs(m[piv][c]) > 0.0)) return false;
				for (int k = 0; k < 8; k++) std::swap(m[c][k], m[piv][k]);
				const double d = m[c][c];
				for (int k = 0; k < 8; k++) m[c][k] /= d;
				for (int r = 0; r < 4; r++) if (r != c) { const double f = m[r][c]; for (int k = 0; k < 8; k++) m[r][k] -= f * m[c][k]; }
			}
			for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) o[4 * r + c] = m[r][4 + c]; // row major
			return true;
		};
		// scene triangles below a mesh root (cached); empty when transforming them for every instance of the
		// mesh would be too much work (more than ~32 M vertex transforms per upload)
		std::vector<uint32_t> inst_count(n_nodes, 0);
		for (uint32_t i = 0; i < NI; i++) inst_count[sc.mesh_instances[i].bvh_root]++;
		std::vector<std::vector<uint32_t>> mesh_tri_cache(n_nodes);
		std::vector<uint8_t> mesh_tri_done(n_nodes, 0);
		auto mesh_tris = [&](uint32_t root) -> const std::vector<uint32_t> & {
			if (!mesh_tri_done[root]) {
				mesh_tri_done[root] = 1;
				std::vector<uint32_t> &out_t = mesh_tri_cache[root];
				std::vector<int32_t> stack{(int32_t)root};
				while (!stack.empty()) {
					const PolarisBvhNode &m = nodes[stack.back()];
					stack.pop_back();
					if (is_leaf(m)) {
						const uint32_t f0 = (uint32_t)(-(int64_t)m.ldata);
						for (uint32_t q = f0; q < f0 + (uint32_t)m.rdata; q++) out_t.push_back(slot_src[q]);
					} else { stack.push_back(m.rdata); stack.push_back(m.ldata); }
					if ((uint64_t)out_t.size() * inst_count[root] > (32ull << 20) / 3) { out_t.clear(); break; }
				}
			}
			return mesh_tri_cache[root];
		};
		// post-order over a tree; instance leaves pull in the (already computed) box of their mesh
		std::vector<std::pair<int32_t, int>> st;
		auto bound_tree = [&](int32_t root) {
			st.clear();
			st.push_back({root, 0});
			while (!st.empty()) {
				const int32_t idx = st.back().first;
				const int phase = st.back().second;
				const PolarisBvhNode &n = nodes[idx];
				if (done[idx]) { st.pop_back(); continue; }
				if (is_leaf(n)) {
					Box b = empty;
					if (n.rdata == 0) {
						const uint32_t inst = (uint32_t)(-(int64_t)n.ldata);
						const uint32_t root = sc.mesh_instances[inst].bvh_root;
						const Box &mb = content[root];
						double fwd[16];
						if (invert(sc.mesh_instances[inst].inv_transform, fwd) && mb.lo[0] <= mb.hi[0]) {
							auto world = [&](const double *p, float *w) {
								for (int r = 0; r < 3; r++) w[r] = (float)(fwd[4 * r] * p[0] + fwd[4 * r + 1] * p[1] + fwd[4 * r + 2] * p[2] + fwd[4 * r + 3]);
							};
							const std::vector<uint32_t> &tris_of_mesh = mesh_tris(root);
							if (!tris_of_mesh.empty()) { // exact: every vertex of the mesh through the instance's matrix
								for (uint32_t t : tris_of_mesh)
									for (int k = 0; k < 3; k++) {
										const float *v = sc.vertices + 4 * (size_t)(3 * t + k);
										const double p[3] = {v[0], v[1], v[2]};
										float w[3];
										world(p, w);
										grow(b, w);
									}
							} else { // big mesh x many instances: the 8 corners of the mesh's box (a superset: may flag a tight host box)
								for (int corner = 0; corner < 8; corner++) {
									const double p[3] = {corner & 1 ? mb.hi[0] : mb.lo[0], corner & 2 ? mb.hi[1] : mb.lo[1], corner & 4 ? mb.hi[2] : mb.lo[2]};
									float w[3];
									world(p, w);
									grow(b, w);
								}
							}
						} else { // singular matrix: nothing can be promised about this instance
							for (int k = 0; k < 3; k++) { b.lo[k] = -3.0e38f; b.hi[k] = 3.0e38f; }
						}
					} else {
						const uint32_t first = (uint32_t)(-(int64_t)n.ldata);
						for (uint32_t s = first; s < first + (uint32_t)n.rdata; s++) {
							const float *v = sc.vertices + 4 * (size_t)(3 * slot_src[s]);
							grow(b, v); grow(b, v + 4); grow(b, v + 8);
						}
					}
					content[idx] = b;
					done[idx] = 1;
					st.pop_back();
				} else if (phase == 0) {
					st.back().second = 1;
					st.push_back({n.ldata, 0});
					st.push_back({n.rdata, 0});
				} else {
					Box b = content[n.ldata];
					merge(b, content[n.rdata]);
					content[idx] = b;
					done[idx] = 1;
					st.pop_back();
				}
			}
		};
		for (uint32_t i = 0; i < NI; i++) bound_tree((int32_t)sc.mesh_instances[i].bvh_root);
		bound_tree(0);
		auto cull_factor = [&](int32_t child) {
			const PolarisBvhNode &n = nodes[child];
			const Box &c = content[child];
			bool inside = true;
			for (int k = 0; k < 3; k++) {
				// a few ulps: the extent of an instance is recomputed here through the forward matrix in
				// double, the host rounded its own way (far inside the 1.001 cull margin either way)
				const float tol = 9.6e-7f * std::fmax(std::fmax(std::fabs(c.lo[k]), std::fabs(c.hi[k])), c.hi[k] - c.lo[k]) + 1e-30f;
				if (!(n.min[k] - tol <= c.lo[k] && c.hi[k] <= n.max[k] + tol)) inside = false;
			}
			return inside ? kCullMargin : std::numeric_limits<float>::infinity();
		};
		for (uint32_t idx = 0; idx < n_nodes; idx++) {
			if (!done[idx] || is_leaf(nodes[idx])) continue;
			PairNodeH &p = out.pairs[idx];
			const float f0 = cull_factor(nodes[idx].ldata), f1 = cull_factor(nodes[idx].rdata);
			memcpy(&p.pad0, &f0, 4);
			memcpy(&p.pad1, &f1, 4);
			if (!(f0 == kCullMargin && f1 == kCullMargin)) out.unbounded_boxes++;
		}
	}

	// ---- renumber inner nodes breadth-first ----------------------------------------------------
	// The traversal kernels keep the first kLdsTopNodes pair records in LDS: every ray walks the top
	// of the tree, so those fetches come from LDS (broadcast when lanes agree) instead of 64
	// per-lane L1 gathers.  BFS continues through an instance leaf into the mesh BVH while the scene
	// has few instances (a single-mesh scene's hot nodes are the top of that mesh's tree).
	{
		std::vector<int32_t> new_id(n_nodes, -1);
		std::vector<int32_t> order;
		order.reserve(n_nodes);
		auto visit = [&](int32_t root) {
			size_t head = order.size();
			if (is_leaf(nodes[root]) || new_id[root] >= 0) return;
			new_id[root] = (int32_t)order.size();
			order.push_back(root);
			while (head < order.size()) {
				const PolarisBvhNode &n = nodes[order[head++]];
				const int32_t kids[2] = {n.ldata, n.rdata};
				for (int32_t c : kids) {
					const PolarisBvhNode &cn = nodes[c];
					int32_t next = c;
					if (is_leaf(cn)) {
						if (cn.rdata != 0 || NI > 16) continue;
						next = (int32_t)sc.mesh_instances[(uint32_t)(-(int64_t)cn.ldata)].bvh_root; // into the instance
						if (is_leaf(nodes[next])) continue;
					}
					if (new_id[next] < 0) {
						new_id[next] = (int32_t)order.size();
						order.push_back(next);
					}
				}
			}
		};
		visit(0);
		for (uint32_t i = 0; i < NI; i++) visit((int32_t)sc.mesh_instances[i].bvh_root);
		auto remap = [&](int32_t ref) { return ref >= 0 ? new_id[ref] : ref; };
		std::vector<PairNodeH> compact(order.size());
		for (size_t k = 0; k < order.size(); k++) {
			PairNodeH p = out.pairs[order[k]];
			p.ref0 = remap(p.ref0);
			p.ref1 = remap(p.ref1);
			compact[k] = p;
		}
		out.pairs.swap(compact);
		if (out.pairs.empty()) out.pairs.push_back(PairNodeH{}); // never an empty device array
		out.root_ref = remap(out.root_ref);
		for (uint32_t i = 0; i < NI; i++) out.insts[i].root_ref = remap(out.insts[i].root_ref);
	}

	// ---- small scenes: triangle slots in the order of how often a ray reaches their leaf ------------------------------------------
	// The tiny-scene traversal mode (kernels.h, kNodesLdsAll) keeps as many triangle records as fit beside the tree in LDS: the
	// slots [0, lds_tris).  A leaf is reached about as often as its box is large (surface area), so the leaves take their slots in
	// descending order of that -- the walls of a room before the facets of a small sphere.  (Bigger scenes keep the depth-first
	// order: there the neighbours of a record in memory are its neighbours in space, which is what the caches want.)
	if (n_slots <= 2046u && out.big_leaves == 0) { // (2046: kernels.h kTinyMaxIndex)
		struct HotLeaf { float area; uint32_t first, count; };
		std::vector<HotLeaf> hot;
		auto note_leaf = [&](int32_t ref, const float *lo, const float *hi) {
			if (ref >= 0) return;
			const uint32_t code = (uint32_t)~ref;
			if ((code & 15u) == 0u) return; // an instance
			float area = std::numeric_limits<float>::infinity(); // (a tree that is one leaf: no box, always reached)
			if (lo) { const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; area = dx * dy + dy * dz + dz * dx; if (!(area >= 0.0f)) area = 0.0f; }
			hot.push_back({area, code >> 4, code & 15u});
		};
		for (const PairNodeH &P : out.pairs) { note_leaf(P.ref0, P.lo0, P.hi0); note_leaf(P.ref1, P.lo1, P.hi1); }
		note_leaf(out.root_ref, nullptr, nullptr);
		for (uint32_t i = 0; i < NI; i++) note_leaf(out.insts[i].root_ref, nullptr, nullptr);
		std::stable_sort(hot.begin(), hot.end(), [](const HotLeaf &a, const HotLeaf &b) { return a.area > b.area; });
		std::vector<uint32_t> new_first(n_slots, 0xFFFFFFFFu), new_src;
		new_src.reserve(n_slots);
		bool clean = true; // every slot in exactly one leaf (leaves shared by two parents keep one place)
		for (const HotLeaf &h : hot) {
			if (h.first + h.count > n_slots) { clean = false; break; }
			if (new_first[h.first] != 0xFFFFFFFFu) continue; // (the same leaf under a second parent: a mesh shared by instances)
			new_first[h.first] = (uint32_t)new_src.size();
			for (uint32_t q = h.first; q < h.first + h.count; q++) new_src.push_back(slot_src[q]);
		}
		if (clean && new_src.size() == n_slots) {
			auto moved = [&](int32_t ref) -> int32_t {
				if (ref >= 0) return ref;
				const uint32_t code = (uint32_t)~ref;
				if ((code & 15u) == 0u) return ref;
				return ~(int32_t)((new_first[code >> 4] << 4) | (code & 15u));
			};
			for (PairNodeH &P : out.pairs) { P.ref0 = moved(P.ref0); P.ref1 = moved(P.ref1); }
			out.root_ref = moved(out.root_ref);
			for (uint32_t i = 0; i < NI; i++) out.insts[i].root_ref = moved(out.insts[i].root_ref);
			slot_src.swap(new_src);
		}
	}

	std::vector<uint8_t> node_class;
	shading_classes(sc, node_class);
	out.tri_bits = NT <= (1u << 24) ? 24 : 31;
	out.tris.assign(n_slots, TriH{});
	for (uint32_t s = 0; s < n_slots; s++) {
		const uint32_t t = slot_src[s];
		const float *v0 = sc.vertices + 4 * (size_t)(3 * t), *v1 = v0 + 4, *v2 = v0 + 8;
		TriH &d = out.tris[s];
		for (int k = 0; k < 3; k++) { d.v0[k] = v0[k]; d.e1[k] = v1[k] - v0[k]; d.e2[k] = v2[k] - v0[k]; }
		d.rank = tri_rank[t] == 0xFFFFFFFFu ? t : tri_rank[t];
		d.orig = out.tri_bits < 31 ? (t | (uint32_t)node_class[sc.material_index[t]] << out.tri_bits) : t;
		d.pad2 = (n_slots <= 2046u && NT <= 2046u) ? tiny_meta_word(d.rank, d.orig) : 0u; // (what the tiny-scene traversal mode keeps per slot)
	}
	return "";
}

} // namespace pol
