// wavefront_reader.hpp -- C++ restatement of the reference's Wavefront OBJ/MTL scene reader
// (asset/scene/reader/wavefront.go, reader.go).  SURVEY.md section 8f-3: the on-disk format in
// front of the path.  Supported statements, as in the reference:
//   .obj : v vn vt f(3|4 vertices) g o usemtl mtllib call
//          camera_fov camera_eye camera_look camera_up
//          instance <mesh> tX tY tZ yaw pitch roll sX sY sZ
//   .mtl : newmtl include Kd Ks Ke Tf Ni KeScaler map_Kd map_Ks map_Ke map_Tf map_bump map_normal
//          mat_expr <layered material expression>
// Anything else is ignored.  Files are read from the local file system (the reference can also
// stream http(s) resources; there is no network here).  The compiled-scene .zip format of the
// reference (encoding/gob inside a zip, reader/zip.go) is Go-specific and not read.
#pragma once

#include <array>
#include <map>
#include <string>
#include <vector>

#include "scene_compiler.hpp"

namespace polaris {
namespace reader {

struct WavefrontMaterial { // wavefront.go:20-55
	std::string Name;
	types::Vec3 Kd, Ks, Ke, Tf;
	float KeScaler = 0, Ni = 0;
	std::string KdTex, KsTex, KeTex, TfTex, BumpTex, NormalTex;
	std::string MaterialExpression;
	std::string AssetRelPath;
	bool Used = false;

	std::string GetExpression() const; // wavefront.go:58-124
};

class WavefrontSceneReader { // wavefront.go:126-148
public:
	compiler::ParsedScene rawScene;
	std::vector<WavefrontMaterial> materials;

	// Read, wavefront.go:164-187: parse, default instances, prune materials, compile.
	Error Read(const std::string &path, compiler::Output *out);
	// The same from memory ("embedded" resources of the reference's tests); includes resolve
	// relative to the current directory.
	Error ReadString(const std::string &name, const std::string &content, compiler::Output *out);

	Error Parse(const std::string &name, const std::string &content);          // parse, :307-440
	Error ParseMaterials(const std::string &name, const std::string &content); // parseMaterials, :651-761
	void CreateDefaultMeshInstances();                                         // :246-258
	void ProcessMaterials();                                                   // :191-243

private:
	std::map<std::string, int> matNameToIndex;
	int curMaterial = -1;
	std::vector<types::Vec3> vertexList, normalList;
	std::vector<std::array<float, 2>> uvList;
	std::vector<std::string> errStack;

	Error emitError(const std::string &file, int line, const std::string &msg) const;
	int defaultMaterial();
	void verifyLastParsedMesh();
	Error parseFile(const std::string &path, bool asMaterials);
	Error parseMeshInstance(const std::vector<std::string> &tok, compiler::MeshInstance *out);
	Error parseFace(const std::vector<std::string> &tok, int relV, int relUv, int relN, std::vector<compiler::Primitive> *out);
	Error finish(compiler::Output *out);
};

// selectFaceCoordIndex, wavefront.go:767-783 (exposed for tests)
Error SelectFaceCoordIndex(const std::string &token, int coordListLen, int relOffset, int *out);

// ReadScene, reader.go:18-35
Error ReadScene(const std::string &filename, compiler::Output *out);

} // namespace reader
} // namespace polaris
