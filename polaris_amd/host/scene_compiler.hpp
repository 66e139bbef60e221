// scene_compiler.hpp -- C++ restatement of the geometry half of the reference's scene compiler
// (asset/compiler/compiler.go:81-231 partitionGeometry + asset/compiler/bvh/bvh_builder.go):
// the step BEFORE the tracer path, producing exactly the arrays the tracer uploads
// (asset/scene/optimized_scene.go:167-190).  SURVEY.md section 8f-2.
//
// In:  triangle meshes, mesh instances (world transforms), an already flattened material node
//      table with the root node of every material (material-expression parsing is a front-end
//      concern and stays out of scope).
// Out: two-level BVH (top tree over instances, one tree per mesh, pre-order nodes, leaf triangles
//      in depth-first order), inverse instance matrices, emissive list, flat geometry arrays.
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "polaris_types.h"
#include "polaris_hip.h"
#include "tracer.hpp"

namespace polaris {
namespace compiler {

struct Vec3 { float x, y, z; };

namespace bvh { // asset/compiler/bvh/bvh_builder.go

struct BoundedVolume { // bvh_builder.go:35-38
	Vec3 bbox[2];
	Vec3 center;
};

// Called for every leaf with the indices (into the input list) of its items, in work-list order.
using LeafCallback = std::function<void(PolarisBvhNode *leaf, const std::vector<uint32_t> &items)>;

// Build (bvh_builder.go:100-124): surface-area-heuristic splits on up to 1024/(depth+1) candidate
// planes per axis, leaves of at most minLeafItems items (or when no split improves the score),
// nodes in pre-order with the left subtree first.
std::vector<PolarisBvhNode> Build(const std::vector<BoundedVolume> &workList, int minLeafItems, const LeafCallback &leafCb);

} // namespace bvh

struct Primitive { // asset/compiler/input/raw_scene.go:22-31
	Vec3 vertices[3], normals[3];
	float uvs[3][2];
	int materialIndex;
};
struct Mesh { std::vector<Primitive> primitives; };
struct MeshInstance { // raw_scene.go:63-69
	uint32_t meshIndex;
	float transform[16]; // column major, local -> world
};

struct Input {
	std::vector<Mesh> meshes;
	std::vector<MeshInstance> instances;
	std::vector<PolarisMaterialNode> materialNodes; // flattened trees (compiler.go:330-438 output)
	std::vector<int32_t> materialRoots;             // material index -> root node (matIndexToMatRoot)
	std::vector<PolarisTextureMetadata> textureMeta;
	std::vector<uint8_t> textureData;
	int32_t sceneDiffuseMatIndex = -1, sceneEmissiveMatIndex = -1; // root nodes or -1
	int minPrimitivesPerLeaf = 10;                  // compiler.go:19
};

struct Output { // asset/scene/optimized_scene.go:167-190
	std::vector<PolarisBvhNode> bvhNodes;
	std::vector<PolarisMeshInstance> meshInstances;
	std::vector<PolarisMaterialNode> materialNodes;
	std::vector<PolarisEmissive> emissives;
	std::vector<PolarisTextureMetadata> textureMeta;
	std::vector<uint8_t> textureData;
	std::vector<float> vertices, normals, uvs; // float4 / float4 / float2 per vertex
	std::vector<uint32_t> materialIndex;
	int32_t sceneDiffuseMatIndex = -1, sceneEmissiveMatIndex = -1;
	PolarisSceneView View() const;
};

// findMaterialNodeByBxdf (compiler.go:246-268)
int32_t FindMaterialNodeByBxdf(const std::vector<PolarisMaterialNode> &nodes, uint32_t nodeIndex, uint32_t bxdf);

Error Compile(const Input &in, Output *out);

} // namespace compiler
} // namespace polaris
