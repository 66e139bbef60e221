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
//         ref >= 0  -> inner node index          ref < 0 -> ~leaf node index
//   * LeafInfo[node] = (ldata, rdata) of a leaf: 8 B fetch when a leaf is popped.
//   * Tri[t] = {v0 | rank, e01 = v1 - v0, e02 = v2 - v0}: the two edge subtractions of
//     Moeller-Trumbore are hoisted to upload (one IEEE subtraction each, so bit-identical to
//     intersect.cl:253-254).  rank = position of the triangle in the reference's left-first
//     depth-first traversal order of its mesh BVH; the closest-hit kernel uses it to break exact
//     ties the way "first tested wins" does in the reference (intersect.cl:281 strict <).
//   * Inst[i] = rows of the inverse 3x4 matrix (so mul4x1 / mul3x1 of util/transform.cl:9-26
//     read 3 float4), tagged root reference, and the instance's rank in top-level DFS order.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "polaris_types.h"

namespace pol {

struct PairNodeH { float lo0[3]; int32_t ref0; float hi0[3]; int32_t pad0; float lo1[3]; int32_t ref1; float hi1[3]; int32_t pad1; };
struct LeafInfoH { int32_t ldata, rdata; };
struct TriH { float v0[3]; uint32_t rank; float e1[3]; uint32_t pad1; float e2[3]; uint32_t pad2; };
struct InstH { float r0[4], r1[4], r2[4]; int32_t root_ref; uint32_t rank; uint32_t pad[2]; };
static_assert(sizeof(PairNodeH) == 64 && sizeof(TriH) == 48 && sizeof(InstH) == 64, "layout");

constexpr int kTraversalStack = 32; // entries per ray, == BVH_MAX_STACK_SIZE (intersect.cl:4)

struct SceneLayout {
	std::vector<PairNodeH> pairs;   // inner nodes only, in breadth-first order (the first kLdsTopNodes are staged in LDS)
	std::vector<LeafInfoH> leaves;  // indexed by node id (only leaves meaningful)
	std::vector<TriH> tris;
	std::vector<InstH> insts;
	int32_t root_ref = 0;
	int max_stack = 0;
};

inline bool is_leaf(const PolarisBvhNode &n) { return n.ldata <= 0; }

// Returns "" on success, otherwise a description of the first inconsistency.
inline std::string build_layout(const PolarisSceneView &sc, SceneLayout &out) {
	const uint32_t NN = sc.num_bvh_nodes, NT = sc.num_triangles, NI = sc.num_mesh_instances;
	if (!sc.bvh_nodes || NN == 0) return "scene has no BVH nodes";
	if (!sc.mesh_instances || NI == 0) return "scene has no mesh instances";
	if (NT == 0 || !sc.vertices || !sc.normals || !sc.uvs || !sc.material_index) return "scene has no triangles";
	if (NT > (1u << 30)) return "too many triangles";
	if (sc.num_material_nodes == 0 || !sc.material_nodes) return "scene has no material nodes";
	if (sc.num_emissives && !sc.emissives) return "emissive list pointer is null";
	if (sc.num_textures && (!sc.texture_meta || !sc.texture_data)) return "texture pointers are null";

	// ---- materials / textures / emissives ------------------------------------------------
	for (uint32_t t = 0; t < sc.num_textures; t++) {
		const PolarisTextureMetadata &m = sc.texture_meta[t];
		if (m.format > POLARIS_TEX_RGBA32F) return "texture " + std::to_string(t) + ": unknown format";
		if (m.width == 0 || m.height == 0) return "texture " + std::to_string(t) + ": empty";
		const uint64_t bpp = m.format == POLARIS_TEX_L8 ? 1 : (m.format == POLARIS_TEX_RGBA32F ? 16 : 4);
		if ((uint64_t)m.data_offset + bpp * m.width * m.height > sc.texture_data_bytes)
			return "texture " + std::to_string(t) + ": data outside the texture blob";
		if ((m.format == POLARIS_TEX_L32F || m.format == POLARIS_TEX_RGBA32F) && (m.data_offset & 3u))
			return "texture " + std::to_string(t) + ": float data not dword aligned";
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
			if (m.type == POLARIS_MAT_OP_MIX_MAP || m.type == POLARIS_MAT_OP_BUMP_MAP || m.type == POLARIS_MAT_OP_NORMAL_MAP)
				if (m.tex < 0 || (uint32_t)m.tex >= sc.num_textures) return who + ": operator texture out of range";
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
	out.pairs.assign(NN, PairNodeH{});
	out.leaves.assign(NN, LeafInfoH{0, 0});
	out.insts.assign(NI, InstH{});
	out.tris.assign(NT, TriH{});
	std::vector<uint8_t> seen(NN, 0);
	auto ref_of = [&](int32_t idx) { return is_leaf(sc.bvh_nodes[idx]) ? ~idx : idx; };

	// iterative left-first DFS from `root`; level = 0 top tree, 1 bottom tree.  `need` tracks the
	// number of stack entries a traversal can hold at a node (one pending sibling per level).
	struct Item { int32_t node; int depth; };
	uint32_t next_inst_rank = 0;
	std::vector<uint32_t> tri_rank(NT, 0xFFFFFFFFu);
	int top_max = 0;
	std::vector<int> inst_entry_depth(NI, 0);
	std::string err;
	auto walk = [&](int32_t root, int level, uint32_t &rank_counter, int &max_depth) -> bool {
		std::vector<Item> st;
		st.push_back({root, 0});
		while (!st.empty()) {
			Item it = st.back();
			st.pop_back();
			if (it.node < 0 || (uint32_t)it.node >= NN) { err = "BVH child index out of range"; return false; }
			if (seen[it.node] && level == 0) { err = "BVH node " + std::to_string(it.node) + " reachable twice"; return false; }
			seen[it.node] = 1;
			if (it.depth > max_depth) max_depth = it.depth;
			if (it.depth > 4 * kTraversalStack) { err = "BVH too deep"; return false; }
			const PolarisBvhNode &n = sc.bvh_nodes[it.node];
			if (is_leaf(n)) {
				out.leaves[it.node] = {n.ldata, n.rdata};
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
					if (first + (uint64_t)n.rdata > NT) { err = "leaf triangle range out of bounds"; return false; }
					for (uint64_t t = first; t < first + (uint64_t)n.rdata; t++)
						if (tri_rank[t] == 0xFFFFFFFFu) tri_rank[t] = rank_counter++;
				}
			} else {
				if (n.rdata <= 0) { err = "inner node " + std::to_string(it.node) + " has a non-positive right child"; return false; }
				if ((uint32_t)n.ldata >= NN || (uint32_t)n.rdata >= NN) { err = "BVH child index out of range"; return false; }
				PairNodeH &p = out.pairs[it.node];
				const PolarisBvhNode &l = sc.bvh_nodes[n.ldata], &r = sc.bvh_nodes[n.rdata];
				memcpy(p.lo0, l.min, 12); memcpy(p.hi0, l.max, 12); p.ref0 = ref_of(n.ldata);
				memcpy(p.lo1, r.min, 12); memcpy(p.hi1, r.max, 12); p.ref1 = ref_of(n.rdata);
				// right pushed first so the left subtree is visited first (reference order)
				st.push_back({n.rdata, it.depth + 1});
				st.push_back({n.ldata, it.depth + 1});
			}
		}
		return true;
	};

	uint32_t dummy = 0;
	if (!walk(0, 0, dummy, top_max)) return err;
	out.root_ref = ref_of(0);
	// bottom trees: one walk per distinct root (meshes are shared by instances)
	std::vector<int> root_depth(NN, -1);
	int need = top_max;
	for (uint32_t i = 0; i < NI; i++) {
		const PolarisMeshInstance &mi = sc.mesh_instances[i];
		if (mi.bvh_root >= NN) return "mesh instance " + std::to_string(i) + ": bvh root out of range";
		if (mi.bvh_root == 0) return "mesh instance " + std::to_string(i) + ": bvh root is the scene root";
		if (root_depth[mi.bvh_root] < 0) {
			int md = 0;
			uint32_t rank = 0;
			std::fill(seen.begin(), seen.end(), 0);
			if (!walk((int32_t)mi.bvh_root, 1, rank, md)) return err;
			root_depth[mi.bvh_root] = md;
		}
		InstH &d = out.insts[i];
		const float *m = mi.inv_transform; // column major: m[4*c + r]
		for (int c = 0; c < 4; c++) { d.r0[c] = m[4 * c + 0]; d.r1[c] = m[4 * c + 1]; d.r2[c] = m[4 * c + 2]; }
		d.root_ref = ref_of((int32_t)mi.bvh_root);
		// stack use below an instance: pending top-level siblings + the exit marker + bottom depth
		const int use = inst_entry_depth[i] + 1 + root_depth[mi.bvh_root];
		if (use > need) need = use;
	}
	out.max_stack = need + 1;
	if (out.max_stack > kTraversalStack)
		return "BVH needs a traversal stack of " + std::to_string(out.max_stack) + " entries; the kernel (like the reference, "
		       "intersect.cl:4) has " + std::to_string(kTraversalStack);

	// ---- renumber inner nodes breadth-first ----------------------------------------------------
	// The traversal kernels keep the first kLdsTopNodes pair records in LDS: every ray walks the top
	// of the tree, so those fetches come from LDS (broadcast when lanes agree) instead of 64
	// per-lane L1 gathers.  BFS continues through an instance leaf into the mesh BVH while the scene
	// has few instances (a single-mesh scene's hot nodes are the top of that mesh's tree).
	{
		std::vector<int32_t> new_id(NN, -1);
		std::vector<int32_t> order;
		order.reserve(NN);
		auto visit = [&](int32_t root) {
			size_t head = order.size();
			if (is_leaf(sc.bvh_nodes[root]) || new_id[root] >= 0) return;
			new_id[root] = (int32_t)order.size();
			order.push_back(root);
			while (head < order.size()) {
				const PolarisBvhNode &n = sc.bvh_nodes[order[head++]];
				const int32_t kids[2] = {n.ldata, n.rdata};
				for (int32_t c : kids) {
					const PolarisBvhNode &cn = sc.bvh_nodes[c];
					int32_t next = c;
					if (is_leaf(cn)) {
						if (cn.rdata != 0 || NI > 16) continue;
						next = (int32_t)sc.mesh_instances[(uint32_t)(-(int64_t)cn.ldata)].bvh_root; // into the instance
						if (is_leaf(sc.bvh_nodes[next])) continue;
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

	for (uint32_t t = 0; t < NT; t++) {
		const float *v0 = sc.vertices + 4 * (size_t)(3 * t), *v1 = v0 + 4, *v2 = v0 + 8;
		TriH &d = out.tris[t];
		for (int k = 0; k < 3; k++) { d.v0[k] = v0[k]; d.e1[k] = v1[k] - v0[k]; d.e2[k] = v2[k] - v0[k]; }
		d.rank = tri_rank[t] == 0xFFFFFFFFu ? t : tri_rank[t];
	}
	return "";
}

} // namespace pol
