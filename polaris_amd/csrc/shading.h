// shading.h -- device-side samplers, BxDFs and light sampling of the polaris path, gfx950.
//
// Each function states the reference code (tracer/opencl/CL/...) whose RESULT it must
// reproduce bit for bit: same IEEE binary32 operations in the same order, built-ins per
// include/polaris_math.h, compiled with -ffp-contract=off.  The structure is not the
// reference's: materials are read in place from the node table (no 64-byte private copy),
// the texture address computation is shared by the three fetch flavours, the two GGX
// evaluation tails are shared by sample/eval, and everything is __forceinline__ into the one
// shade kernel so the 5-way BxDF switch is resolved once per path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "polaris_math.h"
#include "polaris_types.h"

namespace pol {

#define PD __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

PD f3 mk3(float x, float y, float z) { return {x, y, z}; }
PD f3 splat(float s) { return {s, s, s}; }
PD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PD f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
PD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
PD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PD f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PD f3 normalize(f3 v) { float inv = 1.0f / pm_sqrt(dot(v, v)); return {v.x * inv, v.y * inv, v.z * inv}; }
PD float maxcomp(f3 v) { return pm_max(v.x, pm_max(v.y, v.z)); }
PD f3 xyz(float4 v) { return {v.x, v.y, v.z}; }

constexpr float kPi = 3.14159265358979323846f;        // constants.cl:11
constexpr float kTwoPi = 6.28318530718f;               // constants.cl:12
constexpr float kInvPi = 0.31830988618379067154f;      // constants.cl:15
constexpr float kEps = 0.00001f;                       // constants.cl:23 INTERSECTION_EPSILON
constexpr float kLightEps = kEps * 1e3f;               // constants.cl:24
constexpr float kMinRoughness = 0.1f;                  // constants.cl:27
constexpr float kFltMax = 3.402823466e+38f;

// The three small tables every shaded ray walks through dependent loads -- material nodes, emissive
// records, texture metadata -- are read either from global memory or, when they all fit, from the
// copy the shade kernels stage in LDS.  The address space is part of the pointer TYPE (LDS = true:
// address_space(3), so the accesses compile to ds_read; generic pointers would compile to FLAT loads,
// measured 10 % slower shading), hence the template parameter on everything that touches a table.
template <bool LDS> struct Tbl;
template <> struct Tbl<false> {
	typedef const PolarisMaterialNode *Node;
	typedef const PolarisEmissive *Light;
	typedef const PolarisTextureMetadata *TexMeta;
	typedef const float *F;
};
template <> struct Tbl<true> {
	typedef const __attribute__((address_space(3))) PolarisMaterialNode *Node;
	typedef const __attribute__((address_space(3))) PolarisEmissive *Light;
	typedef const __attribute__((address_space(3))) PolarisTextureMetadata *TexMeta;
	typedef const __attribute__((address_space(3))) float *F;
};

// Scene tables as the shade kernel sees them (device pointers).
template <bool LDS> struct SceneT {
	const float4 *vertices;             // [3T] original vertices (surface interpolation)
	const float4 *normals;              // [3T]
	const float2 *uvs;                  // [3T]
	const uint32_t *mat_index;          // [T]
	typename Tbl<LDS>::Node nodes;
	typename Tbl<LDS>::Light emissives;
	typename Tbl<LDS>::TexMeta tex_meta;
	const uint8_t *tex_data;
	uint32_t num_emissives;
	int32_t bg_node;                    // scene_diffuse_mat_index or -1
	uint32_t num_nodes, num_textures;   // table sizes (LDS staging in k_shade)
	// [num_emissives][kLightGeoFloats] the triangle of every area light, packed at upload (polaris_hip.hip, pack_light_geometry):
	// v0 v1 v2 | n0 n1 n2 (xyz + pad each) | uv0 uv1 uv2, 1 / area | and what areaLightGetPdf derives from the light alone:
	// transformed v0, transformed edges e1 e2, their unit normal (xyz + pad each).  light_sample / light_pdf read it instead
	// of chasing emissive -> tri_index -> vertices / normals / uvs: with the table staged in LDS the light's geometry costs
	// a shaded ray no memory round trip (it used to cost two dependent ones, one per function), and the per-light
	// constants -- three point transforms, a cross product, a normalisation and two reciprocals per shaded ray in the
	// reference -- are computed once, on the host, with the same operations (polaris_math.h is bit-identical there).
	typename Tbl<LDS>::F light_geo;
	float sel_pdf; // 1 / number of emissives (emissiveSelect, emissive_sampler.cl:226-237)
	// a hit record's last word = scene triangle index | shading class << tri_bits (scene_layout.h; 31 = scenes too big to carry a class)
	uint32_t tri_bits;
};
constexpr uint32_t kLightGeoFloats = 48;
typedef SceneT<false> SceneDev; // what the host fills in

// ---- PRNG: samplers/random_sampler.cl:7-16 ---------------------------------------------
struct Rng { uint32_t sx, sy; };
PD f2 rng_next(Rng &r) {
	uint32_t x = r.sx * 17u + r.sy * 13123u;
	r.sx = (x << 13) ^ x;
	r.sy ^= (x << 7);
	uint32_t a = x * (x * x * 15731u + 74323u) + 871483u;
	uint32_t b = x * (x * x * 13734u + 37828u) + 234234u;
	const float inv = 1.0f / 4294967296.0f;
	return {(float)a * inv, (float)b * inv};
}

// ---- util/transform.cl ----------------------------------------------------------------
template <class FP> PD f3 xform_point(f3 v, FP m) { // mul4x1, transform.cl:9-16 (column-major 4x4)
	return {m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12], m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13],
	        m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14]};
}
PD f2 latlong_uv(f3 d) { // rayToLatLongUV, transform.cl:28-37
	float at2 = pm_atan2(d.x, d.z);
	float r = pm_sqrt(dot(d, d));
	return {(at2 >= 0.0f ? at2 : (at2 + kTwoPi)) / kTwoPi, pm_acos(d.y / r) / kPi};
}
PD void tangent_frame(f3 n, f3 &u, f3 &v) { // TANGENT_VECTORS, util/surface.cl:4-6
	f3 a = pm_fabs(n.z) < .999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
	u = normalize(cross(a, n));
	v = cross(n, u);
}
PD float schlick(float etaI, float etaT, float iDotN) { // fresnelForDielectric, util/fresnel.cl:8-16
	float eta = etaI / etaT;
	float r0 = ((1.0f - eta) * (1.0f - eta)) / ((1.0f + eta) * (1.0f + eta));
	float c = 1.0f - pm_fabs(iDotN);
	float c1 = c * c;
	return r0 + (1.0f - r0) * c1 * c1 * c;
}

// ---- textures: samplers/texture_sampler.cl --------------------------------------------
// One copy of the address arithmetic (:15-38) and of the texel loads serves the three fetch flavours: a TexQuad holds the
// four texels of the bilinear footprint as un-normalised floats.  All four texels are fetched by the SAME instruction
// sequence whatever the format -- the three dwords at the texel's (dword-aligned) address, decoded afterwards -- so the
// twelve loads are in flight together and a texture sample is ONE memory round trip; a `switch` on the format per texel
// compiled to four serialised load -> wait -> convert steps.  (Luminance-8 texels sit at any byte: the aligned dword that
// holds them is fetched and shifted.  The upload pads the blob so the two spare dwords of the last texel are readable,
// and checks that the other formats are dword aligned.)  texGetSample1f (:114-184) is, operation for operation, the red
// channel of texGetSample3f (:14-110), so scalar parameters read `.x` of the same routine.
struct TexQuad {
	f3 tl, tr, bl, br; // texels (tx,ty) (bx,ty) (tx,by) (bx,by); luminance formats: x only
	float cx, cy;
	bool rgba, bytes;
};
struct RawTexel { uint32_t w0, w1, w2, shift; };
PD RawTexel texel_load(const uint8_t *blob, uint32_t byte_off) {
	const uint32_t *p = reinterpret_cast<const uint32_t *>(blob + (byte_off & ~3u));
	return {p[0], p[1], p[2], (byte_off & 3u) * 8u};
}
PD f3 texel_decode(const RawTexel &r, uint32_t fmt) {
	const bool bytes = fmt == POLARIS_TEX_RGBA8 || fmt == POLARIS_TEX_L8, rgba8 = fmt == POLARIS_TEX_RGBA8;
	const uint32_t b0 = (r.w0 >> r.shift) & 255u; // (shift is 0 for every format but L8)
	return {bytes ? (float)b0 : __uint_as_float(r.w0), rgba8 ? (float)((r.w0 >> 8) & 255u) : __uint_as_float(r.w1),
	        rgba8 ? (float)((r.w0 >> 16) & 255u) : __uint_as_float(r.w2)};
}
template <bool LDS> PD TexQuad tex_fetch(f2 uv, int tex, const SceneT<LDS> &S) { // texture_sampler.cl:15-38 (shared prologue) + the texel loads
	const uint32_t fmt = S.tex_meta[tex].format, w = S.tex_meta[tex].width, h = S.tex_meta[tex].height; // field by field: the record may live in LDS
	const uint32_t off = S.tex_meta[tex].data_offset;
	const uint32_t stride = fmt == POLARIS_TEX_L8 ? 1u : (fmt == POLARIS_TEX_RGBA32F ? 16u : 4u);
	TexQuad q;
	const float sx = (uv.x - pm_floor(uv.x)) * (float)w;
	const float sy = (uv.y - pm_floor(uv.y)) * (float)h;
	const uint32_t tx = pm_clampu((uint32_t)sx, 0u, w - 1), ty = pm_clampu((uint32_t)sy, 0u, h - 1);
	const uint32_t bx = pm_clampu(tx + 1, 0u, w - 1), by = pm_clampu(ty + 1, 0u, h - 1);
	q.cx = sx - (float)tx;
	q.cy = sy - (float)ty;
	q.rgba = fmt == POLARIS_TEX_RGBA8 || fmt == POLARIS_TEX_RGBA32F;
	q.bytes = fmt == POLARIS_TEX_RGBA8 || fmt == POLARIS_TEX_L8;
	const RawTexel r0 = texel_load(S.tex_data, off + (ty * w + tx) * stride), r1 = texel_load(S.tex_data, off + (ty * w + bx) * stride);
	const RawTexel r2 = texel_load(S.tex_data, off + (by * w + tx) * stride), r3 = texel_load(S.tex_data, off + (by * w + bx) * stride);
	q.tl = texel_decode(r0, fmt); q.tr = texel_decode(r1, fmt); q.bl = texel_decode(r2, fmt); q.br = texel_decode(r3, fmt);
	return q;
}
PD float bilerp(float tl, float tr, float bl, float br, float cx, float cy) {
	return pm_mix(pm_mix(tl, bl, cy), pm_mix(tr, br, cy), cx);
}
PD f3 quad_sample3(const TexQuad &q) { // texGetSample3f, texture_sampler.cl:40-107; .x = texGetSample1f, :140-184
	f3 r;
	r.x = bilerp(q.tl.x, q.tr.x, q.bl.x, q.br.x, q.cx, q.cy);
	if (q.rgba) {
		r.y = bilerp(q.tl.y, q.tr.y, q.bl.y, q.br.y, q.cx, q.cy);
		r.z = bilerp(q.tl.z, q.tr.z, q.bl.z, q.br.z, q.cx, q.cy);
		if (q.bytes) r = r / 255.0f;
	} else {
		if (q.bytes) r.x = r.x / 255.0f;
		r.y = r.z = r.x;
	}
	return r;
}
PD f3 quad_bump3(const TexQuad &q) { // texGetBumpSample3f, texture_sampler.cl:187-252
	float s0 = q.tl.x, s1 = q.tr.x, s2 = q.bl.x;
	if (q.bytes) { s0 = s0 / 255.0f; s1 = s1 / 255.0f; s2 = s2 / 255.0f; }
	return splat(0.5f) + 0.5f * normalize(mk3(s1 - s0, s2 - s0, 1.0f));
}
template <bool LDS> PD f3 tex_sample3(f2 uv, int tex, const SceneT<LDS> &S) { return quad_sample3(tex_fetch(uv, tex, S)); }
template <bool LDS> PD f3 mat_color(f2 uv, typename Tbl<LDS>::F def, int tex, const SceneT<LDS> &S) { // matGetSample3f, material_sampler.cl:97-104
	return tex == -1 ? mk3(def[0], def[1], def[2]) : tex_sample3(uv, tex, S);
}

// ---- surface + selected material -----------------------------------------------------
struct Surf { f3 p, n; f2 uv; };
template <bool LDS> struct MatT {  // the leaf selected by the material-tree walk
	typename Tbl<LDS>::Node nd;    // read in place
	uint32_t type;
	float int_ior, ext_ior;        // after the dispersion override (material_sampler.cl:92-94)
	// The leaf's textured parameters.  Every BxDF entry point of the reference re-samples them (sample, pdf and eval of one
	// shaded hit fetch the same reflectance up to three times); they are pure functions of (uv, node), so material_params()
	// evaluates each ONCE per ray and the BxDF code below reads these:
	f3 kcol;                       // matGetSample3f(uv, nd->k, nd->tex): reflectance | specularity | radiance
	f3 tcol;                       // matGetSample3f(uv, nd->t, nd->right_child): transmittance (dielectrics)
	float alpha;                   // the "Disney remapping" of the roughness, rough_conductor.cl:11-13 (rough BxDFs)
};

// matSelectNode, samplers/material_sampler.cl:21-95.  `flags` are the path's dispersion
// flags (PATH_FLAG_DISPERSE_R/G/B = 1/2/4, util/path.cl:4-6), updated in place.
template <bool LDS> PD MatT<LDS> select_material(uint32_t root, Surf &sf, uint32_t &flags, f3 &tint, Rng &rng, const SceneT<LDS> &S) {
	typename Tbl<LDS>::Node nd = S.nodes + root;
	float forceInt = 0.0f, forceExt = 0.0f;
	uint32_t type = nd->type;
	for (int guard = 0; type >= POLARIS_MAT_OP_MIX && guard < 64; ++guard) {
		if (type == POLARIS_MAT_OP_MIX) {
			f2 s = rng_next(rng);
			nd = S.nodes + (s.x < nd->k[0] ? nd->left_child : (uint32_t)nd->right_child);
		} else if (type == POLARIS_MAT_OP_MIX_MAP || type == POLARIS_MAT_OP_BUMP_MAP || type == POLARIS_MAT_OP_NORMAL_MAP) {
			const TexQuad q = tex_fetch(sf.uv, nd->tex, S); // the one texel fetch of the walk
			if (type == POLARIS_MAT_OP_MIX_MAP) {
				f2 s = rng_next(rng);
				float w = quad_sample3(q).x; // texGetSample1f
				nd = S.nodes + (s.x < w ? nd->left_child : (uint32_t)nd->right_child);
			} else {
				f3 u, v;
				tangent_frame(sf.n, u, v);
				if (type == POLARIS_MAT_OP_BUMP_MAP) { // matGetBumpSample3f, :124-131
					f3 s = (quad_bump3(q) * 2.0f) - splat(1.0f);
					sf.n = normalize(u * s.x + v * s.y + sf.n * s.z);
				} else {                                // matGetNormalSample3f, :111-121
					f3 s = (quad_sample3(q) * 2.0f) - splat(1.0f);
					sf.n = normalize(u * s.x + v * s.y + 0.5f * sf.n * s.z);
				}
				nd = S.nodes + nd->left_child;
			}
		} else if (type == POLARIS_MAT_OP_DISPERSE) {
			int ch;
			if (flags & 1u) ch = 0;
			else if (flags & 2u) ch = 1;
			else if (flags & 4u) ch = 2;
			else {
				f2 s = rng_next(rng);
				ch = s.x < 0.333f ? 0 : (s.x < 0.666f ? 1 : 2);
				flags |= 1u << ch;
			}
			tint = mk3(ch == 0 ? 1.0f : 0.0f, ch == 1 ? 1.0f : 0.0f, ch == 2 ? 1.0f : 0.0f);
			forceInt = nd->k[ch];
			forceExt = nd->t[ch];
			nd = S.nodes + nd->left_child;
		} else {
			return {nd, POLARIS_BXDF_INVALID, 0.0f, 0.0f, splat(0.0f), splat(0.0f), 0.0f};
		}
		type = nd->type;
	}
	if (type >= POLARIS_MAT_OP_MIX) type = POLARIS_BXDF_INVALID; // malformed tree (cycle): reject instead of spinning
	return {nd, type, pm_max(nd->int_ior, forceInt), pm_max(nd->ext_ior, forceExt), splat(0.0f), splat(0.0f), 0.0f};
}

// The selected leaf's parameters (see MatT).  Untextured parameters are the node's constants; the textured ones go through
// ONE copy of the sampling code (a three-trip loop that is not unrolled: shade kernels whose code does not fit the
// instruction cache pay for every inlined copy of the texel arithmetic).
template <bool LDS> PD void material_params(const Surf &sf, MatT<LDS> &m, const SceneT<LDS> &S) {
	const bool rough = (m.type & (POLARIS_BXDF_ROUGH_CONDUCTOR | POLARIS_BXDF_ROUGH_DIELECTRIC)) != 0;
	const bool transmits = (m.type & (POLARIS_BXDF_DIELECTRIC | POLARIS_BXDF_ROUGH_DIELECTRIC)) != 0;
	const bool valid = m.type != POLARIS_BXDF_INVALID;
	m.kcol = mk3(m.nd->k[0], m.nd->k[1], m.nd->k[2]);
	m.tcol = mk3(m.nd->t[0], m.nd->t[1], m.nd->t[2]);
	float r = m.nd->scale;
	const int tk = valid ? m.nd->tex : -1, tt = transmits ? m.nd->right_child : -1, tr = rough ? m.nd->roughness_tex : -1;
	if (tk != -1 || tt != -1 || tr != -1) {
#pragma clang loop unroll(disable)
		for (int which = 0; which < 3; which++) {
			const int tex = which == 0 ? tk : (which == 1 ? tt : tr);
			if (tex != -1) {
				const f3 v = tex_sample3(sf.uv, tex, S);
				if (which == 0) m.kcol = v;
				else if (which == 1) m.tcol = v;
				else r = v.x; // matGetSample1f, material_sampler.cl:108-114
			}
		}
	}
	r = pm_clamp(r, kMinRoughness, 1.0f);
	m.alpha = r * r;
}

// ---- distributions: samplers/distribution_sampler.cl ----------------------------------
PD float ggx_g1(float a, f3 v, f3 n, f3 m) { // _ggxGetG1, :20-33
	float nDotV = dot(n, v), mDotV = dot(m, v);
	if (nDotV * mDotV <= 0.0f) return 0.0f;
	float sq = nDotV * nDotV;
	float tanSq = sq > 0.0f ? (1.0f - sq) / sq : 0.0f;
	return 2.0f / (1.0f + pm_sqrt(1.0f + a * a * tanSq));
}
PD float ggx_g(float a, f3 i, f3 o, f3 n, f3 m) { return ggx_g1(a, i, n, m) * ggx_g1(a, o, n, m); } // :37-39
PD float ggx_d(float a, f3 n, f3 m) { // ggxGetD, :42-56
	float nDotM = dot(n, m);
	if (nDotM <= 0.0f) return 0.0f;
	float sq = nDotM * nDotM;
	float tanSq = nDotM != 0.0f ? ((1.0f - sq) / sq) : 0.0f;
	float aSq = a * a;
	float denom = kPi * sq * sq * (aSq + tanSq) * (aSq + tanSq);
	return denom > 0.0f ? (aSq / denom) : 0.0f;
}
PD f3 ggx_sample(float a, f3 n, f2 rnd) { // ggxGetSample, :59-76 (sinPhi >= 0 quirk kept)
	f3 u, v;
	tangent_frame(n, u, v);
	float theta = pm_atan(a * pm_sqrt(rnd.x / (1.0f - rnd.x)));
	theta = theta >= 0.0f ? theta : (theta + kTwoPi);
	float cosTheta = pm_cos(theta);
	float sinTheta = pm_sqrt(1.0f - cosTheta * cosTheta);
	float cosPhi = pm_cos(kTwoPi * rnd.y);
	float sinPhi = pm_sqrt(1.0f - cosPhi * cosPhi);
	return normalize(u * sinTheta * cosPhi + v * sinTheta * sinPhi + n * cosTheta);
}
PD float ggx_reflect_pdf(float a, f3 o, f3 n, f3 h) { // ggxGetReflectionPdf, :78-87
	float nDotH = pm_fabs(dot(n, h)), oDotH = pm_fabs(dot(o, h));
	float denom = 4.0f * oDotH;
	return denom == 0.0f ? 0.0f : ggx_d(a, n, h) * nDotH / denom;
}
PD float ggx_refract_pdf(float a, float etaI, float etaT, f3 i, f3 o, f3 n, f3 h) { // ggxGetRefractionPdf, :89-98
	float iDotH = pm_fabs(dot(i, h)), oDotH = pm_fabs(dot(o, h)), hDotN = pm_fabs(dot(h, n));
	float denom = (etaI * iDotH + etaT * oDotH) * (etaI * iDotH + etaT * oDotH);
	return denom > 0.0f ? ggx_d(a, n, h) * hDotN * oDotH * etaT * etaT / denom : 0.0f;
}
PD f3 cosine_hemisphere(f3 n, f2 rnd) { // cosWeightedHemisphereGetSample, :101-112
	float rd = pm_sqrt(rnd.x);
	float phi = kTwoPi * rnd.y;
	f3 u, v;
	tangent_frame(n, u, v);
	return normalize(u * rd * pm_cos(phi) + v * rd * pm_sin(phi) + n * pm_sqrt(1 - rnd.x));
}

// ---- BxDFs: bxdf/*.cl ------------------------------------------------------------------
PD f3 specular_tail(const Surf &sf, float a, f3 ks, float f, f3 i, f3 o, f3 h) { // eq. 20: rough_conductor.cl:27-39
	float iDotN = dot(i, sf.n), oDotN = dot(o, sf.n);
	float d = ggx_d(a, sf.n, h);
	float g = ggx_g(a, i, o, sf.n, h);
	float denom = 4.0f * iDotN * oDotN;
	return denom > 0.0f ? ks * f * d * g / denom : splat(0.0f);
}
template <bool LDS> PD f3 transmit_tail(const Surf &sf, const MatT<LDS> &m, const SceneT<LDS> &S, float a, float etaI, float etaT, float f, float iDotN,
                    f3 i, f3 o, f3 h) { // eq. 21: rough_dielectric.cl:73-93
	float iDotH = pm_fabs(dot(i, h)), oDotH = pm_fabs(dot(o, h));
	float oDotN = dot(o, sf.n);
	float fd = iDotN * oDotN * (etaI * iDotH + etaT * oDotH) * (etaI * iDotH + etaT * oDotH);
	if (fd == 0.0f) return splat(0.0f);
	float focus = pm_fabs(etaT * etaT * iDotH * oDotH / fd);
	float d = ggx_d(a, sf.n, h);
	float g = ggx_g(a, i, o, sf.n, h);
	const f3 tf = m.tcol;
	return tf * (1.0f - f) * d * g * focus;
}
template <bool LDS> PD f3 mirror_value(const Surf &sf, const MatT<LDS> &m, const SceneT<LDS> &S, float iDotN) { // conductor.cl:23-29
	float f = m.int_ior != 0.0f ? schlick(m.ext_ior, m.int_ior, iDotN) : 1.0f;
	const f3 ks = m.kcol;
	return iDotN != 0.0f ? f * ks / iDotN : splat(0.0f);
}

// bxdfGetSample, bxdf/bxdf.cl:31-55.
// The five BxDFs of the reference are five separate functions; a wave that holds several of them executes each one's code in
// turn.  Where two of them perform the SAME operations on the same operands -- the Fresnel / total-reflection prologue of
// the two dielectrics, the GGX half-vector sample, and the whole reflection branch of the two rough BxDFs (mirror the
// incoming direction about the sampled half vector, re-derive the half vector, pdf, equation 20) -- the code exists once and
// the lanes of both types run it together; every lane still performs exactly its own BxDF's operation sequence.
template <bool LDS> PD f3 bxdf_sample(const Surf &sf, const MatT<LDS> &m, const SceneT<LDS> &S, f2 rnd, f3 i, f3 &o, float &pdf) {
	const f3 n = sf.n;
	if (m.type == POLARIS_BXDF_DIFFUSE) { // diffuse.cl:12-20
		o = cosine_hemisphere(n, rnd);
		pdf = dot(n, o) * kInvPi;
		return m.kcol * kInvPi;
	}
	const float iDotN = dot(i, n);
	if (m.type == POLARIS_BXDF_CONDUCTOR) { // conductor.cl:12-30
		o = 2.0f * iDotN * n - i;
		pdf = 1.0f;
		return mirror_value(sf, m, S, iDotN);
	}
	const bool rough = (m.type & (POLARIS_BXDF_ROUGH_CONDUCTOR | POLARIS_BXDF_ROUGH_DIELECTRIC)) != 0;
	const bool diel = (m.type & (POLARIS_BXDF_DIELECTRIC | POLARIS_BXDF_ROUGH_DIELECTRIC)) != 0;
	if (!rough && !diel) return splat(0.0f); // emissive / invalid leaves have no BxDF (bxdf.cl:53)
	// dielectric.cl:13-27 == rough_dielectric.cl:11-30: orient the interface, Fresnel term, total internal reflection
	// (cosTSq uses eta, not eta^2: quirk kept); rough_conductor.cl:22-24: Fresnel term of the conductor
	float etaI = m.ext_ior, etaT = m.int_ior, eta = 0.0f, f, cosTSq = 1.0f;
	if (diel) {
		if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
		eta = etaI / etaT;
		f = schlick(etaI, etaT, iDotN);
		cosTSq = 1.0f + eta * (iDotN * iDotN - 1.0f);
	} else {
		f = m.int_ior != 0.0f ? schlick(m.ext_ior, m.int_ior, iDotN) : 1.0f;
	}
	const bool reflects = !diel || cosTSq <= 0.0f || rnd.x <= f;
	if (!rough) { // dielectric.cl:28-45
		f3 kVal;
		if (reflects) {
			o = -pm_sign(iDotN) * 2.0f * iDotN * n - i;
			kVal = m.kcol;
			pdf = cosTSq <= 0.0f ? 1.0f : f;
		} else {
			o = (eta * iDotN - pm_sign(iDotN) * pm_sqrt(cosTSq)) * n - eta * i;
			kVal = eta * eta * m.tcol;
			pdf = 1.0f - f;
		}
		return iDotN != 0.0f ? pdf * kVal / pm_fabs(iDotN) : splat(0.0f);
	}
	const float a = m.alpha;
	const f3 hs = ggx_sample(a, n, rnd); // rough_conductor.cl:14, rough_dielectric.cl:24
	if (reflects) { // rough_conductor.cl:15-39 and the reflection branch of rough_dielectric.cl:32-56
		o = 2.0f * dot(i, hs) * hs - i;
		const f3 h = normalize(i + o);
		// the conductor's pdf is taken about the SAMPLED half vector, the dielectric's about the re-derived one
		const bool conductor = m.type == POLARIS_BXDF_ROUGH_CONDUCTOR;
		const f3 hp = {conductor ? hs.x : h.x, conductor ? hs.y : h.y, conductor ? hs.z : h.z};
		const float p = ggx_reflect_pdf(a, o, n, hp);
		pdf = (!conductor && cosTSq <= 0.0f) ? 1.0f : p;
		return specular_tail(sf, a, m.kcol, f, i, o, h);
	}
	// refraction branch of rough_dielectric.cl:58-93
	o = (eta * iDotN - pm_sign(iDotN) * pm_sqrt(cosTSq)) * hs - eta * i;
	const f3 h = normalize(-(etaI * i + etaT * o));
	pdf = ggx_refract_pdf(a, etaI, etaT, i, o, n, h);
	return transmit_tail(sf, m, S, a, etaI, etaT, f, iDotN, i, o, h);
}

// bxdfGetPdf (bxdf.cl:58-78) and bxdfEval (bxdf.cl:82-105) for a given outgoing direction,
// evaluated together (the NEE path of shadeHits needs both for the same direction).  The reflection side of the rough
// dielectric is, operation for operation, the rough conductor: one copy (see bxdf_sample).
template <bool LDS> PD void bxdf_pdf_eval(const Surf &sf, const MatT<LDS> &m, const SceneT<LDS> &S, f3 i, f3 o, bool want_eval, float &pdf, f3 &val) {
	const f3 n = sf.n;
	pdf = 0.0f;
	val = splat(0.0f);
	if (m.type == POLARIS_BXDF_DIFFUSE) { // diffuse.cl:24-32
		pdf = dot(n, o) * kInvPi;
		if (want_eval) val = m.kcol * kInvPi;
		return;
	}
	const float iDotN = dot(i, n);
	if (m.type == POLARIS_BXDF_CONDUCTOR) { // conductor.cl:33-62
		f3 e = 2.0f * iDotN * n - i;
		float ed = dot(e, o);
		bool match = ed >= 0.0f && ed <= 0.001f;
		pdf = match ? 1.0f : 0.0f;
		if (want_eval && match) val = mirror_value(sf, m, S, iDotN);
		return;
	}
	if ((m.type & (POLARIS_BXDF_ROUGH_CONDUCTOR | POLARIS_BXDF_ROUGH_DIELECTRIC)) == 0) return; // dielectric.cl:49-60: always 0
	const float a = m.alpha;
	const bool conductor = m.type == POLARIS_BXDF_ROUGH_CONDUCTOR;
	if (conductor || iDotN > 0.0f) { // rough_conductor.cl:43-78; rough_dielectric.cl:97-166, incoming direction on the outside
		const f3 h = normalize(i + o);
		pdf = ggx_reflect_pdf(a, o, n, h);
		if (want_eval) {
			// (the dielectric's interface is not flipped on this side: its Fresnel term is the conductor's expression)
			const float f = (conductor && m.int_ior == 0.0f) ? 1.0f : schlick(m.ext_ior, m.int_ior, iDotN);
			val = specular_tail(sf, a, m.kcol, f, i, o, h);
		}
		return;
	}
	float etaI = m.ext_ior, etaT = m.int_ior;
	if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
	const f3 h = normalize(-(etaI * i + etaT * o));
	pdf = ggx_refract_pdf(a, etaI, etaT, i, o, n, h);
	if (want_eval) {
		const float f = schlick(etaI, etaT, iDotN);
		val = transmit_tail(sf, m, S, a, etaI, etaT, f, iDotN, i, o, h);
	}
}

// ---- lights: samplers/emissive_sampler.cl ---------------------------------------------
struct LightSample { f3 radiance, dir; float pdf, dist; };

template <bool LDS> PD LightSample light_sample(const Surf &sf, typename Tbl<LDS>::Light em, uint32_t ei, const SceneT<LDS> &S, f2 rnd) { // emissiveGetSample, :176-198
	LightSample L;
	typename Tbl<LDS>::Node mn = S.nodes + em->mat_node_index;
	const uint32_t ltype = em->type;
	if (ltype != POLARIS_EMISSIVE_ENVIRONMENT && ltype != POLARIS_EMISSIVE_AREA) return {splat(0.0f), splat(0.0f), 0.0f, 0.0f};
	// both light types end in matGetSample3f(uv, mn->k, mn->tex) for a uv of their own: computed per type, sampled in one place
	f2 luv;
	f3 en = splat(0.0f);
	float d2 = 0.0f, g_inv_area = 0.0f;
	if (ltype == POLARIS_EMISSIVE_ENVIRONMENT) { // :16-37
		L.dir = cosine_hemisphere(sf.n, rnd);
		L.pdf = pm_max(0.0f, dot(sf.n, L.dir)) * kInvPi;
		L.dist = kFltMax;
		luv = latlong_uv(L.dir);
	} else { // area light, :51-113 (normal goes through the point transform: quirk a-9(4) kept)
		float r1 = pm_sqrt(rnd.x);
		float ru = (1.0f - rnd.y) * r1, rv = rnd.y * r1;
		float w0 = 1.0f - ru - rv;
		typename Tbl<LDS>::F g = S.light_geo + ei * kLightGeoFloats; // the light's triangle (vertices, normals, uvs)
		f3 a = mk3(g[0], g[1], g[2]), b = mk3(g[4], g[5], g[6]), c = mk3(g[8], g[9], g[10]);
		f3 p = mk3(w0 * a.x + ru * b.x + rv * c.x, w0 * a.y + ru * b.y + rv * c.y, w0 * a.z + ru * b.z + rv * c.z);
		f3 ep = xform_point(p, em->transform);
		a = mk3(g[12], g[13], g[14]); b = mk3(g[16], g[17], g[18]); c = mk3(g[20], g[21], g[22]);
		f3 nn = mk3(w0 * a.x + ru * b.x + rv * c.x, w0 * a.y + ru * b.y + rv * c.y, w0 * a.z + ru * b.z + rv * c.z);
		en = xform_point(nn, em->transform);
		f2 ua = {g[24], g[25]}, ub = {g[26], g[27]}, uc = {g[28], g[29]};
		g_inv_area = g[30];
		luv = {w0 * ua.x + ru * ub.x + rv * uc.x, w0 * ua.y + ru * ub.y + rv * uc.y};
		f3 er = ep - sf.p;
		d2 = dot(er, er);
		L.dir = normalize(er);
		L.dist = pm_sqrt(d2);
	}
	const f3 ke = mat_color(luv, mn->k, mn->tex, S);
	if (ltype == POLARIS_EMISSIVE_ENVIRONMENT) {
		L.radiance = mn->scale * ke * kInvPi;
	} else {
		float nDotOut = dot(en, -L.dir);
		if (nDotOut > 0.0f) {
			L.pdf = g_inv_area; // 1.0f / em->area, from the table
			L.radiance = mn->scale * ke * nDotOut / d2;
		} else {
			L.pdf = 0.0f;
			L.radiance = splat(0.0f);
		}
	}
	return L;
}

template <bool LDS> PD float light_pdf(const Surf &sf, typename Tbl<LDS>::Light em, uint32_t ei, const SceneT<LDS> &S, f3 o) { // emissiveGetPdf, :201-223
	if (em->type == POLARIS_EMISSIVE_ENVIRONMENT) return pm_max(0.0f, dot(sf.n, o) * kInvPi); // :39-47
	if (em->type != POLARIS_EMISSIVE_AREA) return 0.0f;
	// areaLightGetPdf, :117-173 (edges go through the point transform: quirk kept)
	// v0, e1 = v1 - v0, e2 = v2 - v0 through the point transform, and normalize(cross(e1, e2)): per-light constants (:121-131, :165)
	typename Tbl<LDS>::F g = S.light_geo + ei * kLightGeoFloats;
	const f3 v0 = mk3(g[32], g[33], g[34]), e1 = mk3(g[36], g[37], g[38]), e2 = mk3(g[40], g[41], g[42]);
	f3 pv = cross(o, e2);
	float det = dot(e1, pv);
	if (pm_fabs(det) < kEps) return 0.0f;
	float inv = pm_rcp(det);
	f3 tv = sf.p - v0;
	float u = dot(tv, pv) * inv;
	if (u < 0.0f || u > 1.0f) return 0.0f;
	f3 qv = cross(tv, e1);
	float v = dot(o, qv) * inv;
	if (v < 0.0f || u + v > 1.0f) return 0.0f;
	float t = dot(e2, qv) * inv;
	if (t < kEps) return 0.0f;
	const f3 en = mk3(g[44], g[45], g[46]);
	float denom = em->area * pm_fabs(dot(en, o));
	return denom > 0.0f ? (t * t) / denom : 0.0f;
}

} // namespace pol
