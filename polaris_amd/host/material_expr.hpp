// material_expr.hpp -- C++ restatement of the reference's layered-material expression language
// (asset/material/material_expr.y:76-176 grammar, :196-322 lexer, asset/material/node.go
// validation).  SURVEY.md section 8f-3: the text format in which .mtl files (`mat_expr ...`) and
// the Wavefront reader (wavefront.go:57-124) describe the material trees the tracer's
// material_sampler walks.
//
//   material_def  := bxdf_spec | op_spec
//   bxdf_spec     := bxdf_type '(' [ param { ',' param } ] ')'
//   bxdf_type     := diffuse | conductor | roughConductor | dielectric | roughDielectric | emissive
//   param         := reflectance|specularity|transmittance|radiance ':' ( '{' f ',' f ',' f '}' | "texture" )
//                  | intIOR|extIOR ':' ( f | "material name" )
//                  | scale ':' f
//                  | roughness ':' ( f | "texture" )
//   op_spec       := mix '(' arg ',' arg ',' f ')'
//                  | mixMap '(' arg ',' arg ',' "texture" ')'
//                  | bumpMap '(' arg ',' "texture" ')' | normalMap '(' arg ',' "texture" ')'
//                  | disperse '(' arg ',' intIOR ':' float3 ',' extIOR ':' float3 ')'
//   arg           := bxdf_spec | op_spec | "material name"
//
// A quoted string is a texture when it ends in one of the reference's image extensions
// (asset/material/texture.go:5-8), otherwise a material name.  Numbers cannot be negative (the
// reference's lexer only starts a number at a digit or '.').
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "tracer.hpp"

namespace polaris {
namespace material {

// Parameter names, asset/material/node.go:10-19
extern const char *const ParamReflectance, *const ParamSpecularity, *const ParamTransmittance, *const ParamRadiance,
    *const ParamIntIOR, *const ParamExtIOR, *const ParamScale, *const ParamRoughness;

// asset/material/defaults.go
constexpr float DefaultRoughness = 0.1f;
constexpr float DefaultReflectance[4] = {0.2f, 0.2f, 0.2f, 0.0f};
constexpr float DefaultSpecularity[4] = {1.0f, 1.0f, 1.0f, 0.0f};
constexpr float DefaultTransmittance[4] = {1.0f, 1.0f, 1.0f, 0.0f};
constexpr float DefaultRadiance[4] = {1.0f, 1.0f, 1.0f, 0.0f};
constexpr float DefaultRadianceScaler = 1.0f;
constexpr float DefaultIntIOR = 1.51714f;   // KnownIORs["Glass"]
constexpr float DefaultExtIOR = 1.0002926f; // KnownIORs["Air"]

// Case-insensitive lookup of a named index of refraction (asset/material/ior.go:264-270).  This
// build carries the common entries of the reference's table, not all ~270 of them; an unknown
// name is the same error the reference raises for a name outside its table.
Error IOR(const std::string &name, float *out);

// true when `s` ends in an image extension the reference treats as a texture file name
bool IsTextureName(const std::string &s);

struct Param { // BxdfParamNode, node.go:70-73
	enum Kind { Vec3, Float, MaterialName, Texture };
	std::string name;
	Kind kind = Float;
	float v[3] = {0, 0, 0}; // Vec3
	float f = 0;            // Float
	std::string s;          // MaterialName | Texture
};

struct Expr { // ExprNode implementations, node.go:58-106
	enum Kind { Bxdf, Mix, MixMap, BumpMap, NormalMap, Disperse, MaterialRef };
	Kind kind = Bxdf;
	uint32_t bxdfType = 0;           // Bxdf: POLARIS_BXDF_*
	std::vector<Param> params;       // Bxdf
	std::unique_ptr<Expr> left, right; // operators: arguments
	float weight = 0;                // Mix
	std::string texture;             // MixMap | BumpMap | NormalMap
	float intIOR[3] = {0, 0, 0}, extIOR[3] = {0, 0, 0}; // Disperse
	std::string ref;                 // MaterialRef

	Error Validate() const; // node.go:108-258
};

// ParseExpression, material_expr.y:353-361.  On failure returns the first error met.
Error ParseExpression(const std::string &input, std::unique_ptr<Expr> *out);

const char *BxdfName(uint32_t bxdfType);           // bxdf.go:41-58
uint32_t BxdfTypeFromName(const std::string &name); // bxdf.go:20-38 (0 = invalid)

} // namespace material
} // namespace polaris
