#include "scene_compiler.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>

#include "material_expr.hpp"
#include "texture.hpp"

namespace polaris {
namespace compiler {

static inline Vec3 vmin(Vec3 a, Vec3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
static inline Vec3 vmax(Vec3 a, Vec3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
static inline float axis_of(const Vec3 &v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

namespace bvh {

namespace {
constexpr float minSideLength = 1e-3f; // bvh_builder.go:19-21
constexpr float minSplitStep = 1e-5f;  // bvh_builder.go:23-26

struct Builder {
	const std::vector<BoundedVolume> &items;
	int minLeafItems;
	const LeafCallback &leafCb;
	std::vector<PolarisBvhNode> nodes;

	// surfaceAreaHeuristic.ScorePartition (bvh_builder.go:288-308)
	float scorePartition(const std::vector<uint32_t> &w) const {
		if (w.empty()) return FLT_MAX;
		Vec3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
		for (uint32_t i : w) { lo = vmin(lo, items[i].bbox[0]); hi = vmax(hi, items[i].bbox[1]); }
		const float sx = hi.x - lo.x, sy = hi.y - lo.y, sz = hi.z - lo.z;
		return float(w.size()) * (sx * sy + sy * sz + sx * sz);
	}
	// surfaceAreaHeuristic.ScoreSplit (bvh_builder.go:246-283)
	float scoreSplit(const std::vector<uint32_t> &w, int axis, float splitPoint, int *lc, int *rc) const {
		Vec3 lmin{FLT_MAX, FLT_MAX, FLT_MAX}, rmin = lmin, lmax{-FLT_MAX, -FLT_MAX, -FLT_MAX}, rmax = lmax;
		int l = 0, r = 0;
		for (uint32_t i : w) {
			const BoundedVolume &it = items[i];
			if (axis_of(it.center, axis) < splitPoint) { l++; lmin = vmin(lmin, it.bbox[0]); lmax = vmax(lmax, it.bbox[1]); }
			else { r++; rmin = vmin(rmin, it.bbox[0]); rmax = vmax(rmax, it.bbox[1]); }
		}
		*lc = l; *rc = r;
		if (l == 0 || r == 0) return FLT_MAX;
		const float lx = lmax.x - lmin.x, ly = lmax.y - lmin.y, lz = lmax.z - lmin.z;
		const float rx = rmax.x - rmin.x, ry = rmax.y - rmin.y, rz = rmax.z - rmin.z;
		return float(l) * (lx * ly + ly * lz + lx * lz) + float(r) * (rx * ry + ry * rz + rx * rz);
	}

	uint32_t createLeaf(PolarisBvhNode node, const std::vector<uint32_t> &w) { // bvh_builder.go:215-229
		leafCb(&node, w);
		nodes.push_back(node);
		return uint32_t(nodes.size() - 1);
	}

	uint32_t partition(const std::vector<uint32_t> &w, int depth) { // bvh_builder.go:127-211
		PolarisBvhNode node{};
		Vec3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
		for (uint32_t i : w) { lo = vmin(lo, items[i].bbox[0]); hi = vmax(hi, items[i].bbox[1]); }
		node.min[0] = lo.x; node.min[1] = lo.y; node.min[2] = lo.z;
		node.max[0] = hi.x; node.max[1] = hi.y; node.max[2] = hi.z;
		if ((int)w.size() <= minLeafItems) return createLeaf(node, w);

		float bestScore = scorePartition(w);
		bool haveSplit = false;
		int bestAxis = 0;
		float bestPoint = 0.0f;
		const float side[3] = {hi.x - lo.x, hi.y - lo.y, hi.z - lo.z};
		const float mn[3] = {lo.x, lo.y, lo.z}, mx[3] = {hi.x, hi.y, hi.z};
		for (int axis = 0; axis < 3; axis++) {
			if (side[axis] < minSideLength) continue;
			const float splitStep = side[axis] / (1024.0f / float(depth + 1));
			if (splitStep < minSplitStep) continue;
			// candidates are scored concurrently by the reference (one goroutine each) and the
			// best is kept with a strict "<": ties go to whichever goroutine reports first.
			// Here: ascending (axis, splitPoint) order, deterministic.
			std::vector<float> points;
			for (float p = mn[axis]; p < mx[axis]; p += splitStep) {
				points.push_back(p);
				if (p + splitStep == p) break; // step below half an ulp of p (coordinates far from the origin): the reference spins here
			}
			std::vector<float> scores(points.size());
#pragma omp parallel for schedule(static) if (points.size() * w.size() > (1u << 16))
			for (long k = 0; k < (long)points.size(); k++) {
				int l, r;
				scores[k] = scoreSplit(w, axis, points[k], &l, &r);
			}
			for (size_t k = 0; k < points.size(); k++)
				if (scores[k] < bestScore) { bestScore = scores[k]; haveSplit = true; bestAxis = axis; bestPoint = points[k]; }
		}
		if (!haveSplit) return createLeaf(node, w);

		std::vector<uint32_t> left, right;
		for (uint32_t i : w) (axis_of(items[i].center, bestAxis) < bestPoint ? left : right).push_back(i);
		const uint32_t nodeIndex = uint32_t(nodes.size());
		nodes.push_back(node);
		const uint32_t l = partition(left, depth + 1);
		const uint32_t r = partition(right, depth + 1);
		nodes[nodeIndex].ldata = int32_t(l); // SetChildNodes, optimized_scene.go:39-42
		nodes[nodeIndex].rdata = int32_t(r);
		return nodeIndex;
	}
};
} // namespace

std::vector<PolarisBvhNode> Build(const std::vector<BoundedVolume> &workList, int minLeafItems, const LeafCallback &leafCb) {
	Builder b{workList, minLeafItems, leafCb, {}};
	std::vector<uint32_t> all(workList.size());
	for (size_t i = 0; i < all.size(); i++) all[i] = uint32_t(i);
	b.partition(all, 0);
	return std::move(b.nodes);
}

} // namespace bvh

PolarisSceneView Output::View() const {
	PolarisSceneView v{};
	v.bvh_nodes = bvhNodes.data(); v.num_bvh_nodes = uint32_t(bvhNodes.size());
	v.mesh_instances = meshInstances.data(); v.num_mesh_instances = uint32_t(meshInstances.size());
	v.material_nodes = materialNodes.data(); v.num_material_nodes = uint32_t(materialNodes.size());
	v.emissives = emissives.data(); v.num_emissives = uint32_t(emissives.size());
	v.texture_data = textureData.data(); v.texture_data_bytes = uint32_t(textureData.size());
	v.texture_meta = textureMeta.data(); v.num_textures = uint32_t(textureMeta.size());
	v.vertices = vertices.data(); v.normals = normals.data(); v.uvs = uvs.data();
	v.material_index = materialIndex.data(); v.num_triangles = uint32_t(materialIndex.size());
	v.scene_diffuse_mat_index = sceneDiffuseMatIndex;
	v.scene_emissive_mat_index = sceneEmissiveMatIndex;
	return v;
}

int32_t FindMaterialNodeByBxdf(const std::vector<PolarisMaterialNode> &nodes, uint32_t nodeIndex, uint32_t bxdf) {
	if (nodeIndex >= nodes.size()) return -1;
	const PolarisMaterialNode &n = nodes[nodeIndex];
	if (n.type < POLARIS_MAT_OP_MIX) return n.type == bxdf ? int32_t(nodeIndex) : -1;
	int32_t out = FindMaterialNodeByBxdf(nodes, n.left_child, bxdf);
	if (out != -1) return out;
	if (n.type == POLARIS_MAT_OP_MIX) out = FindMaterialNodeByBxdf(nodes, uint32_t(n.right_child), bxdf);
	return out;
}

// 4x4 inverse (general), column major
static bool invert4(const float a[16], float out[16]) { // Mat4.Inv, types/matrix.go:108-138 (float32, same order)
	types::Mat4 m;
	memcpy(m.m, a, sizeof m.m);
	bool singular = false;
	const types::Mat4 inv = m.Inv(&singular);
	memcpy(out, inv.m, sizeof inv.m);
	return !singular;
}

static Vec3 xform(const float m[16], Vec3 v) {
	return {m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12], m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13], m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14]};
}

Error Compile(const Input &in, Output *out) { // compiler.go:81-231
	if (!out) return Error{POLARIS_E_BAD_ARGUMENT, "output is null"};
	if (in.meshes.empty() || in.instances.empty()) return Error{POLARIS_E_BAD_SCENE, "scene has no meshes or no mesh instances"};
	*out = Output{};
	out->materialNodes = in.materialNodes;
	out->textureMeta = in.textureMeta;
	out->textureData = in.textureData;
	out->sceneDiffuseMatIndex = in.sceneDiffuseMatIndex;
	out->sceneEmissiveMatIndex = in.sceneEmissiveMatIndex;

	// mesh bounding boxes (raw_scene.go:98-114) and instance volumes (world-space box of the 8 corners)
	std::vector<Vec3> meshLo(in.meshes.size()), meshHi(in.meshes.size());
	for (size_t m = 0; m < in.meshes.size(); m++) {
		Vec3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
		for (const Primitive &p : in.meshes[m].primitives)
			for (int k = 0; k < 3; k++) { lo = vmin(lo, p.vertices[k]); hi = vmax(hi, p.vertices[k]); }
		meshLo[m] = lo; meshHi[m] = hi;
	}
	std::vector<bvh::BoundedVolume> instVols(in.instances.size());
	for (size_t i = 0; i < in.instances.size(); i++) {
		const MeshInstance &mi = in.instances[i];
		if (mi.meshIndex >= in.meshes.size()) return Error{POLARIS_E_BAD_SCENE, "mesh instance references a missing mesh"};
		if (mi.hasBounds) { instVols[i] = {{mi.bbox[0], mi.bbox[1]}, mi.center}; continue; }
		Vec3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
		const Vec3 a = meshLo[mi.meshIndex], b = meshHi[mi.meshIndex];
		for (int c = 0; c < 8; c++) {
			Vec3 p = xform(mi.transform, {c & 1 ? b.x : a.x, c & 2 ? b.y : a.y, c & 4 ? b.z : a.z});
			lo = vmin(lo, p); hi = vmax(hi, p);
		}
		instVols[i] = {{lo, hi}, {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)}};
	}
	// top-level tree: one mesh instance per leaf (compiler.go:88-103)
	out->bvhNodes = bvh::Build(instVols, 1, [](PolarisBvhNode *leaf, const std::vector<uint32_t> &items) {
		leaf->ldata = -int32_t(items[0]); // SetMeshIndex
		leaf->rdata = 0;
	});

	size_t totalPrims = 0;
	for (const Mesh &m : in.meshes) totalPrims += m.primitives.size();
	out->vertices.assign(totalPrims * 3 * 4, 0.0f);
	out->normals.assign(totalPrims * 3 * 4, 0.0f);
	out->uvs.assign(totalPrims * 3 * 2, 0.0f);
	out->materialIndex.assign(totalPrims, 0);

	struct MeshEmissive { uint32_t mesh; PolarisEmissive e; };
	std::vector<MeshEmissive> meshEmissives;
	std::vector<uint32_t> meshBvhRoots(in.meshes.size());
	uint32_t vertexOffset = 0, primOffset = 0;
	Error err;
	for (size_t mIndex = 0; mIndex < in.meshes.size(); mIndex++) { // compiler.go:120-180
		const Mesh &pm = in.meshes[mIndex];
		std::vector<bvh::BoundedVolume> vols(pm.primitives.size());
		for (size_t i = 0; i < vols.size(); i++) {
			const Primitive &p = pm.primitives[i];
			if (p.hasBounds) { vols[i] = {{p.bbox[0], p.bbox[1]}, p.center}; continue; }
			Vec3 lo = vmin(vmin(p.vertices[0], p.vertices[1]), p.vertices[2]), hi = vmax(vmax(p.vertices[0], p.vertices[1]), p.vertices[2]);
			vols[i] = {{lo, hi}, {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)}};
		}
		std::vector<PolarisBvhNode> nodes = bvh::Build(vols, in.minPrimitivesPerLeaf, [&](PolarisBvhNode *leaf, const std::vector<uint32_t> &items) {
			leaf->ldata = -int32_t(primOffset); // SetPrimitives
			leaf->rdata = int32_t(items.size());
			for (uint32_t it : items) {
				const Primitive &prim = pm.primitives[it];
				for (int k = 0; k < 3; k++) {
					float *v = &out->vertices[(size_t)(vertexOffset + k) * 4], *n = &out->normals[(size_t)(vertexOffset + k) * 4];
					v[0] = prim.vertices[k].x; v[1] = prim.vertices[k].y; v[2] = prim.vertices[k].z;
					n[0] = prim.normals[k].x; n[1] = prim.normals[k].y; n[2] = prim.normals[k].z;
					out->uvs[(size_t)(vertexOffset + k) * 2] = prim.uvs[k][0];
					out->uvs[(size_t)(vertexOffset + k) * 2 + 1] = prim.uvs[k][1];
				}
				if (prim.materialIndex < 0 || (size_t)prim.materialIndex >= in.materialRoots.size()) {
					err = Error{POLARIS_E_BAD_SCENE, "primitive references a missing material"};
				} else {
					const int32_t root = in.materialRoots[prim.materialIndex];
					out->materialIndex[primOffset] = uint32_t(root);
					const int32_t emissiveNode = FindMaterialNodeByBxdf(in.materialNodes, uint32_t(root), POLARIS_BXDF_EMISSIVE);
					if (emissiveNode != -1) { // compiler.go:147-160
						PolarisEmissive e{};
						const Vec3 &a = prim.vertices[0], &b = prim.vertices[1], &c = prim.vertices[2];
						const Vec3 u{c.x - a.x, c.y - a.y, c.z - a.z}, w{c.x - b.x, c.y - b.y, c.z - b.z};
						const Vec3 cr{u.y * w.z - u.z * w.y, u.z * w.x - u.x * w.z, u.x * w.y - u.y * w.x};
						e.area = 0.5f * std::sqrt(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
						e.tri_index = primOffset;
						e.mat_node_index = uint32_t(emissiveNode);
						e.type = POLARIS_EMISSIVE_AREA;
						meshEmissives.push_back({uint32_t(mIndex), e});
					}
				}
				vertexOffset += 3;
				primOffset++;
			}
		});
		if (err) return err;
		const int32_t offset = int32_t(out->bvhNodes.size()); // compiler.go:172-179
		meshBvhRoots[mIndex] = uint32_t(offset);
		for (PolarisBvhNode &n : nodes)
			if (n.ldata > 0) { n.ldata += offset; n.rdata += offset; } // OffsetChildNodes
		out->bvhNodes.insert(out->bvhNodes.end(), nodes.begin(), nodes.end());
	}

	out->meshInstances.resize(in.instances.size()); // compiler.go:185-192
	for (size_t i = 0; i < in.instances.size(); i++) {
		PolarisMeshInstance &mi = out->meshInstances[i];
		memset(&mi, 0, sizeof mi);
		mi.mesh_index = in.instances[i].meshIndex;
		mi.bvh_root = meshBvhRoots[mi.mesh_index];
		if (!invert4(in.instances[i].transform, mi.inv_transform)) return Error{POLARIS_E_BAD_SCENE, "singular mesh instance transform"};
	}
	for (const PolarisMeshInstance &mi : out->meshInstances) // compiler.go:200-211 (instance order, then mesh emissive order)
		for (const MeshEmissive &me : meshEmissives) {
			if (me.mesh != mi.mesh_index) continue;
			PolarisEmissive e = me.e;
			memcpy(e.transform, mi.inv_transform, sizeof e.transform); // the reference stores the instance's (inverse) matrix
			out->emissives.push_back(e);
		}
	if (in.sceneEmissiveMatIndex != -1) { // compiler.go:214-220
		const int32_t en = FindMaterialNodeByBxdf(in.materialNodes, uint32_t(in.sceneEmissiveMatIndex), POLARIS_BXDF_EMISSIVE);
		if (en != -1) {
			PolarisEmissive e{};
			e.mat_node_index = uint32_t(en);
			e.type = POLARIS_EMISSIVE_ENVIRONMENT;
			out->emissives.push_back(e);
		}
	}
	return Error::Nil();
}

// ---- material expressions -> layered material trees (compiler.go:271-552) -----------------------
namespace {

struct MaterialCompiler {
	const ParsedScene &in;
	Output &out;
	std::map<int, int32_t> matIndexToMatRoot;        // compiler.go:29
	std::map<std::string, int32_t> texIndexCache;    // compiler.go:33
	std::vector<std::string> matRefList;             // compiler.go:39

	static std::string dirOf(const std::string &path) {
		const size_t slash = path.find_last_of('/');
		return slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : path.substr(0, slash));
	}

	// bakeTexture, compiler.go:496-552
	Error bakeTexture(const Material &mat, const std::string &texName, int32_t *index) {
		std::string rel = texName;
		for (char &c : rel) if (c == '\\') c = '/'; // asset.NewResource, resource.go:48
		*index = -1;
		if (rel.find("://") != std::string::npos) {
			out.warnings.push_back("\"" + mat.name + "\": skipping remote texture \"" + texName + "\" (no network in this build)");
			return Error::Nil();
		}
		const std::string path = mat.assetRelPath.empty() ? rel : dirOf(mat.assetRelPath) + "/" + rel;
		const auto cached = texIndexCache.find(path);
		if (cached != texIndexCache.end()) { *index = cached->second; return Error::Nil(); }
		FILE *probe = fopen(path.c_str(), "rb");
		if (!probe) { // a texture that cannot be opened is skipped with a warning, compiler.go:499-502
			out.warnings.push_back("\"" + mat.name + "\": skipping missing texture \"" + texName + "\"");
			return Error::Nil();
		}
		fclose(probe);
		texture::Texture tex;
		if (Error e = texture::Load(path, &tex)) return Error{e.code, "\"" + mat.name + "\": " + e.msg};
		PolarisTextureMetadata md{};
		md.format = tex.format; md.width = tex.width; md.height = tex.height;
		md.data_offset = uint32_t(out.textureData.size());
		out.textureData.insert(out.textureData.end(), tex.data.begin(), tex.data.end());
		while (out.textureData.size() % 4) out.textureData.push_back(0); // align4, compiler.go:555-562
		out.textureMeta.push_back(md);
		*index = int32_t(out.textureMeta.size() - 1);
		texIndexCache[path] = *index;
		return Error::Nil();
	}

	// setMaterialNodeParameter, compiler.go:441-492
	Error setParameter(const Material &mat, PolarisMaterialNode *node, const material::Param &p) {
		using material::Param;
		if (p.name == material::ParamReflectance || p.name == material::ParamSpecularity || p.name == material::ParamRadiance) {
			if (p.kind == Param::Vec3) { node->k[0] = p.v[0]; node->k[1] = p.v[1]; node->k[2] = p.v[2]; node->k[3] = 0.0f; }
			else if (p.kind == Param::Texture) return bakeTexture(mat, p.s, &node->tex);
		} else if (p.name == material::ParamTransmittance) {
			if (p.kind == Param::Vec3) { node->t[0] = p.v[0]; node->t[1] = p.v[1]; node->t[2] = p.v[2]; node->t[3] = 0.0f; }
			else if (p.kind == Param::Texture) return bakeTexture(mat, p.s, &node->right_child);
		} else if (p.name == material::ParamIntIOR || p.name == material::ParamExtIOR) {
			float *dst = p.name == material::ParamExtIOR ? &node->ext_ior : &node->int_ior;
			if (p.kind == Param::Float) *dst = p.f;
			else if (p.kind == Param::MaterialName) return material::IOR(p.s, dst);
		} else if (p.name == material::ParamScale) {
			node->scale = p.f;
		} else if (p.name == material::ParamRoughness) {
			if (p.kind == Param::Float) node->scale = p.f;
			else if (p.kind == Param::Texture) return bakeTexture(mat, p.s, &node->roughness_tex);
		}
		return Error::Nil();
	}

	// generateMaterial, compiler.go:313-327
	Error generateMaterial(const Material &mat, int32_t *root) {
		std::unique_ptr<material::Expr> expr;
		if (Error e = material::ParseExpression(mat.expression, &expr)) return Error{e.code, "material \"" + mat.name + "\": " + e.msg};
		if (Error e = expr->Validate()) return Error{e.code, "material \"" + mat.name + "\": " + e.msg};
		matRefList.push_back(mat.name);
		return generateTree(mat, *expr, root);
	}

	// generateMaterialTree, compiler.go:331-438: children first, then the node itself
	Error generateTree(const Material &mat, const material::Expr &e, int32_t *index) {
		using material::Expr;
		PolarisMaterialNode node{};
		node.left_child = uint32_t(-1); node.right_child = -1; node.tex = -1; node.roughness_tex = -1; // Union1 {0,-1,-1,-1}, Union5 {-1}
		node.int_ior = material::DefaultIntIOR;
		node.ext_ior = material::DefaultExtIOR;
		auto child = [&](const std::unique_ptr<Expr> &c, int32_t *dst) -> Error { return generateTree(mat, *c, dst); };
		auto set4 = [](float *dst, const float *src) { memcpy(dst, src, 16); };
		int32_t tmp = -1;
		switch (e.kind) {
		case Expr::MaterialRef: {
			for (const std::string &seen : matRefList)
				if (seen == e.ref) {
					std::string chain;
					for (size_t i = 0; i < matRefList.size(); i++) chain += (i ? " -> " : "") + matRefList[i];
					return Error{POLARIS_E_BAD_SCENE, "detected circular dependency loop while processing \"" + matRefList[0] + "\"; " + chain + " => " + e.ref};
				}
			for (const Material &m : in.materials)
				if (m.name == e.ref) return generateMaterial(m, index);
			return Error{POLARIS_E_BAD_SCENE, "material \"" + mat.name + "\" references undefined material \"" + e.ref + "\""};
		}
		case Expr::Bxdf:
			node.type = e.bxdfType;
			switch (e.bxdfType) { // defaults, compiler.go:363-387
			case POLARIS_BXDF_DIFFUSE: set4(node.k, material::DefaultReflectance); break;
			case POLARIS_BXDF_CONDUCTOR: set4(node.k, material::DefaultSpecularity); break;
			case POLARIS_BXDF_DIELECTRIC: set4(node.k, material::DefaultSpecularity); set4(node.t, material::DefaultTransmittance); break;
			case POLARIS_BXDF_ROUGH_CONDUCTOR: set4(node.k, material::DefaultSpecularity); node.scale = material::DefaultRoughness; break;
			case POLARIS_BXDF_ROUGH_DIELECTRIC:
				set4(node.k, material::DefaultSpecularity); set4(node.t, material::DefaultTransmittance); node.scale = material::DefaultRoughness;
				break;
			case POLARIS_BXDF_EMISSIVE: set4(node.k, material::DefaultRadiance); node.scale = material::DefaultRadianceScaler; break;
			}
			for (const material::Param &p : e.params)
				if (Error err = setParameter(mat, &node, p)) return err;
			break;
		case Expr::Mix:
		case Expr::MixMap:
			node.type = e.kind == Expr::Mix ? POLARIS_MAT_OP_MIX : POLARIS_MAT_OP_MIX_MAP;
			if (Error err = child(e.left, &tmp)) return err;
			node.left_child = uint32_t(tmp);
			if (Error err = child(e.right, &node.right_child)) return err;
			if (e.kind == Expr::Mix) node.k[0] = e.weight;
			else if (Error err = bakeTexture(mat, e.texture, &node.tex)) return err;
			break;
		case Expr::BumpMap:
		case Expr::NormalMap:
			node.type = e.kind == Expr::BumpMap ? POLARIS_MAT_OP_BUMP_MAP : POLARIS_MAT_OP_NORMAL_MAP;
			if (Error err = child(e.left, &tmp)) return err;
			node.left_child = uint32_t(tmp);
			if (Error err = bakeTexture(mat, e.texture, &node.tex)) return err;
			break;
		case Expr::Disperse:
			node.type = POLARIS_MAT_OP_DISPERSE;
			if (Error err = child(e.left, &tmp)) return err;
			node.left_child = uint32_t(tmp);
			for (int k = 0; k < 3; k++) { node.k[k] = e.intIOR[k]; node.t[k] = e.extIOR[k]; }
			break;
		}
		out.materialNodes.push_back(node);
		*index = int32_t(out.materialNodes.size() - 1);
		return Error::Nil();
	}

	// createLayeredMaterialTrees, compiler.go:271-309
	Error run() {
		for (size_t i = 0; i < in.materials.size(); i++) {
			const Material &mat = in.materials[i];
			if (!mat.used) continue; // reached lazily through material references
			matRefList.clear();
			int32_t root = -1;
			if (Error e = generateMaterial(mat, &root)) return e;
			matIndexToMatRoot[int(i)] = root;
			if (mat.name == SceneDiffuseMaterialName) out.sceneDiffuseMatIndex = root;
			else if (mat.name == SceneEmissiveMaterialName) out.sceneEmissiveMatIndex = root;
		}
		return Error::Nil();
	}
};

} // namespace

Error CompileScene(const ParsedScene &in, Output *out) { // compiler.Compile, compiler.go:44-75
	if (!out) return Error{POLARIS_E_BAD_ARGUMENT, "output is null"};
	Output mats;
	MaterialCompiler mc{in, mats, {}, {}, {}};
	if (Error e = mc.run()) return e;

	Input geo;
	geo.meshes = in.meshes;
	geo.instances = in.instances;
	geo.materialNodes = mats.materialNodes;
	geo.textureMeta = mats.textureMeta;
	geo.textureData = mats.textureData;
	geo.sceneDiffuseMatIndex = mats.sceneDiffuseMatIndex;
	geo.sceneEmissiveMatIndex = mats.sceneEmissiveMatIndex;
	geo.minPrimitivesPerLeaf = in.minPrimitivesPerLeaf;
	geo.materialRoots.assign(in.materials.size(), -1);
	for (const auto &kv : mc.matIndexToMatRoot) geo.materialRoots[size_t(kv.first)] = kv.second;
	for (const Mesh &m : in.meshes)
		for (const Primitive &p : m.primitives)
			if (p.materialIndex < 0 || size_t(p.materialIndex) >= geo.materialRoots.size() || geo.materialRoots[size_t(p.materialIndex)] < 0)
				return Error{POLARIS_E_BAD_SCENE, "mesh \"" + m.name + "\": a primitive references a material that was not compiled"};
	if (Error e = Compile(geo, out)) return e;
	out->warnings = mats.warnings;

	out->camera = scene::Camera(in.camera.fov); // setupCamera, compiler.go:233-241
	out->camera.Position = in.camera.eye;
	out->camera.LookAt = in.camera.look;
	out->camera.Up = in.camera.up;
	return Error::Nil();
}

} // namespace compiler
} // namespace polaris
