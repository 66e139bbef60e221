// scene_compiler.hpp -- C++ restatement of the reference's scene compiler
// (asset/compiler/compiler.go + asset/compiler/bvh/bvh_builder.go): the step BEFORE the tracer
// path, producing exactly the arrays the tracer uploads (asset/scene/optimized_scene.go:167-190).
// SURVEY.md section 8f-2.
//
// Two entry points:
//   Compile(Input)        geometry half only (partitionGeometry, compiler.go:81-231): meshes,
//                         instances and an already flattened material node table in; two-level
//                         BVH (top tree over instances, one tree per mesh, pre-order nodes, leaf
//                         triangles in depth-first order), inverse instance matrices, emissive
//                         list and flat geometry arrays out.
//   CompileScene(Parsed)  the whole of compiler.Compile (compiler.go:44-75): material
//                         expressions -> layered material trees + baked textures
//                         (createLayeredMaterialTrees, :271-438; bakeTexture, :496-552), then the
//                         geometry half, then setupCamera (:233-241).  Input is what the
//                         Wavefront reader produces (wavefront_reader.hpp).
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "camera.hpp"
#include "polaris_hip.h"
#include "polaris_types.h"
#include "tracer.hpp"
#include "types.hpp"

namespace polaris {
namespace compiler {

using Vec3 = types::Vec3;

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
	float uvs[3][2] = {{0, 0}, {0, 0}, {0, 0}};
	int materialIndex = 0;
	// bbox/center as the scene reader set them (SetBBox/SetCenter); when hasBounds is false the
	// compiler derives the box from the vertices and uses its midpoint as the center.
	bool hasBounds = false;
	Vec3 bbox[2], center;
};
struct Mesh {
	std::string name;
	std::vector<Primitive> primitives;
};
struct MeshInstance { // raw_scene.go:63-69
	uint32_t meshIndex = 0;
	float transform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; // column major, local -> world
	bool hasBounds = false; // as for Primitive; default = world box of the mesh box's 8 corners
	Vec3 bbox[2], center;
};
struct Material { // raw_scene.go:10-20
	std::string name, expression;
	std::string assetRelPath; // file the material came from: textures resolve relative to its directory
	bool used = false;
};
struct ParsedCamera { // raw_scene.go:133-138, defaults of NewScene :150-160
	float fov = 45.0f;
	Vec3 eye{0, 0, 0}, look{0, 0, -1}, up{0, 1, 0};
};
struct ParsedScene { // input.Scene, raw_scene.go:142-147
	std::vector<Mesh> meshes;
	std::vector<MeshInstance> instances;
	std::vector<Material> materials;
	ParsedCamera camera;
	int minPrimitivesPerLeaf = 10;
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
	scene::Camera camera;              // CompileScene only (setupCamera)
	std::vector<std::string> warnings; // e.g. skipped missing textures (the reference logs them)
	PolarisSceneView View() const;
};

constexpr const char *SceneDiffuseMaterialName = "scene_diffuse_material";   // compiler.go:20
constexpr const char *SceneEmissiveMaterialName = "scene_emissive_material"; // compiler.go:21

// findMaterialNodeByBxdf (compiler.go:246-268)
int32_t FindMaterialNodeByBxdf(const std::vector<PolarisMaterialNode> &nodes, uint32_t nodeIndex, uint32_t bxdf);

Error Compile(const Input &in, Output *out);
Error CompileScene(const ParsedScene &in, Output *out);

} // namespace compiler
} // namespace polaris
