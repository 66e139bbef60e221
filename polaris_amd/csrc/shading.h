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
};
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
struct Texel4 { float x, y, z, w; };
struct TexFetch {
	const uint8_t *base;
	uint32_t fmt, w, i00, i10, i01, i11; // element indices of TL, TR(bx,ty), BL(tx,by), BR
	float cx, cy;
};
template <bool LDS> PD TexFetch tex_setup(f2 uv, int tex, const SceneT<LDS> &S) { // texture_sampler.cl:15-38 (shared prologue)
	PolarisTextureMetadata m; // field by field: the record may live in LDS
	m.format = S.tex_meta[tex].format; m.width = S.tex_meta[tex].width; m.height = S.tex_meta[tex].height; m.data_offset = S.tex_meta[tex].data_offset;
	TexFetch t;
	float sx = (uv.x - pm_floor(uv.x)) * (float)m.width;
	float sy = (uv.y - pm_floor(uv.y)) * (float)m.height;
	uint32_t tx = pm_clampu((uint32_t)sx, 0u, m.width - 1), ty = pm_clampu((uint32_t)sy, 0u, m.height - 1);
	uint32_t bx = pm_clampu(tx + 1, 0u, m.width - 1), by = pm_clampu(ty + 1, 0u, m.height - 1);
	t.cx = sx - (float)tx;
	t.cy = sy - (float)ty;
	t.base = S.tex_data + m.data_offset;
	t.fmt = m.format;
	t.w = m.width;
	t.i00 = ty * m.width + tx; t.i10 = ty * m.width + bx; t.i01 = by * m.width + tx; t.i11 = by * m.width + bx;
	return t;
}
PD float bilerp(float tl, float tr, float bl, float br, float cx, float cy) {
	return pm_mix(pm_mix(tl, bl, cy), pm_mix(tr, br, cy), cx);
}
PD float tex_chan(const TexFetch &t, uint32_t i, uint32_t c) { // one channel of texel i as float (un-normalised)
	switch (t.fmt) {
	case POLARIS_TEX_RGBA8: return (float)t.base[4 * i + c];
	case POLARIS_TEX_RGBA32F: return ((const float *)t.base)[4 * i + c];
	case POLARIS_TEX_L8: return (float)t.base[i];
	default: return ((const float *)t.base)[i];
	}
}
template <bool LDS> PD f3 tex_sample3(f2 uv, int tex, const SceneT<LDS> &S) { // texGetSample3f, texture_sampler.cl:14-110
	TexFetch t = tex_setup(uv, tex, S);
	if (t.fmt > POLARIS_TEX_RGBA32F) return splat(0.0f);
	bool rgba = t.fmt == POLARIS_TEX_RGBA8 || t.fmt == POLARIS_TEX_RGBA32F;
	bool bytes = t.fmt == POLARIS_TEX_RGBA8 || t.fmt == POLARIS_TEX_L8;
	f3 r;
	r.x = bilerp(tex_chan(t, t.i00, 0), tex_chan(t, t.i10, 0), tex_chan(t, t.i01, 0), tex_chan(t, t.i11, 0), t.cx, t.cy);
	if (rgba) {
		r.y = bilerp(tex_chan(t, t.i00, 1), tex_chan(t, t.i10, 1), tex_chan(t, t.i01, 1), tex_chan(t, t.i11, 1), t.cx, t.cy);
		r.z = bilerp(tex_chan(t, t.i00, 2), tex_chan(t, t.i10, 2), tex_chan(t, t.i01, 2), tex_chan(t, t.i11, 2), t.cx, t.cy);
		if (bytes) r = r / 255.0f;
	} else {
		if (bytes) r.x = r.x / 255.0f;
		r.y = r.z = r.x;
	}
	return r;
}
template <bool LDS> PD float tex_sample1(f2 uv, int tex, const SceneT<LDS> &S) { // texGetSample1f, texture_sampler.cl:114-184 (red channel)
	TexFetch t = tex_setup(uv, tex, S);
	if (t.fmt > POLARIS_TEX_RGBA32F) return 0.0f;
	float r = bilerp(tex_chan(t, t.i00, 0), tex_chan(t, t.i10, 0), tex_chan(t, t.i01, 0), tex_chan(t, t.i11, 0), t.cx, t.cy);
	return (t.fmt == POLARIS_TEX_RGBA8 || t.fmt == POLARIS_TEX_L8) ? r / 255.0f : r;
}
template <bool LDS> PD f3 tex_bump3(f2 uv, int tex, const SceneT<LDS> &S) { // texGetBumpSample3f, texture_sampler.cl:187-252
	TexFetch t = tex_setup(uv, tex, S);
	if (t.fmt > POLARIS_TEX_RGBA32F) return splat(0.0f);
	float s0 = tex_chan(t, t.i00, 0), s1 = tex_chan(t, t.i10, 0), s2 = tex_chan(t, t.i01, 0);
	if (t.fmt == POLARIS_TEX_RGBA8 || t.fmt == POLARIS_TEX_L8) { s0 = s0 / 255.0f; s1 = s1 / 255.0f; s2 = s2 / 255.0f; }
	return splat(0.5f) + 0.5f * normalize(mk3(s1 - s0, s2 - s0, 1.0f));
}
template <bool LDS> PD f3 mat_color(f2 uv, typename Tbl<LDS>::F def, int tex, const SceneT<LDS> &S) { // matGetSample3f, material_sampler.cl:97-104
	return tex == -1 ? mk3(def[0], def[1], def[2]) : tex_sample3(uv, tex, S);
}
template <bool LDS> PD float mat_scalar(f2 uv, float def, int tex, const SceneT<LDS> &S) { // matGetSample1f, material_sampler.cl:108-114
	return tex == -1 ? def : tex_sample1(uv, tex, S);
}

// ---- surface + selected material -----------------------------------------------------
struct Surf { f3 p, n; f2 uv; };
template <bool LDS> struct MatT {  // the leaf selected by the material-tree walk
	typename Tbl<LDS>::Node nd;    // read in place
	uint32_t type;
	float int_ior, ext_ior;        // after the dispersion override (material_sampler.cl:92-94)
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
		} else if (type == POLARIS_MAT_OP_MIX_MAP) {
			f2 s = rng_next(rng);
			float w = tex_sample1(sf.uv, nd->tex, S);
			nd = S.nodes + (s.x < w ? nd->left_child : (uint32_t)nd->right_child);
		} else if (type == POLARIS_MAT_OP_BUMP_MAP || type == POLARIS_MAT_OP_NORMAL_MAP) {
			f3 u, v;
			tangent_frame(sf.n, u, v);
			if (type == POLARIS_MAT_OP_BUMP_MAP) { // matGetBumpSample3f, :124-131
				f3 s = (tex_bump3(sf.uv, nd->tex, S) * 2.0f) - splat(1.0f);
				sf.n = normalize(u * s.x + v * s.y + sf.n * s.z);
			} else {                                // matGetNormalSample3f, :111-121
				f3 s = (tex_sample3(sf.uv, nd->tex, S) * 2.0f) - splat(1.0f);
				sf.n = normalize(u * s.x + v * s.y + 0.5f * sf.n * s.z);
			}
			nd = S.nodes + nd->left_child;
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
			return {nd, POLARIS_BXDF_INVALID, 0.0f, 0.0f};
		}
		type = nd->type;
	}
	if (type >= POLARIS_MAT_OP_MIX) type = POLARIS_BXDF_INVALID; // malformed tree (cycle): reject instead of spinning
	return {nd, type, pm_max(nd->int_ior, forceInt), pm_max(nd->ext_ior, forceExt)};
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
template <bool LDS> PD float alpha_of(const Surf &sf, const MatT<LDS> &m, const SceneT<LDS> &S) { // "Disney remapping", rough_conductor.cl:11-13
	float r = pm_clamp(mat_scalar(sf.uv, m.nd->scale, m.nd->roughness_tex, S), kMinRoughness, 1.0f);
	return r * r;
}
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
	f3 tf = mat_color(sf.uv, m.nd->t, m.nd->right_child, S);
	return tf * (1.0f - f) * d * g * focus;
}
template <bool LDS> PD f3 mirror_value(const Surf &sf, const MatT<LDS> &m, const SceneT<LDS> &S, float iDotN) { // conductor.cl:23-29
	float f = m.int_ior != 0.0f ? schlick(m.ext_ior, m.int_ior, iDotN) : 1.0f;
	f3 ks = mat_color(sf.uv, m.nd->k, m.nd->tex, S);
	return iDotN != 0.0f ? f * ks / iDotN : splat(0.0f);
}

// bxdfGetSample, bxdf/bxdf.cl:31-55
template <bool LDS> PD f3 bxdf_sample(const Surf &sf, const MatT<LDS> &m, const SceneT<LDS> &S, f2 rnd, f3 i, f3 &o, float &pdf) {
	const f3 n = sf.n;
	switch (m.type) {
	case POLARIS_BXDF_DIFFUSE: { // diffuse.cl:12-20
		o = cosine_hemisphere(n, rnd);
		pdf = dot(n, o) * kInvPi;
		return mat_color(sf.uv, m.nd->k, m.nd->tex, S) * kInvPi;
	}
	case POLARIS_BXDF_CONDUCTOR: { // conductor.cl:12-30
		float iDotN = dot(i, n);
		o = 2.0f * iDotN * n - i;
		pdf = 1.0f;
		return mirror_value(sf, m, S, iDotN);
	}
	case POLARIS_BXDF_DIELECTRIC: { // dielectric.cl:12-45 (cosTSq uses eta, not eta^2: quirk kept)
		float iDotN = dot(i, n);
		float etaI = m.ext_ior, etaT = m.int_ior;
		if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
		float eta = etaI / etaT;
		float f = schlick(etaI, etaT, iDotN);
		f3 kVal;
		float cosTSq = 1.0f + eta * (iDotN * iDotN - 1.0f);
		if (cosTSq <= 0.0f || rnd.x <= f) {
			o = -pm_sign(iDotN) * 2.0f * iDotN * n - i;
			kVal = mat_color(sf.uv, m.nd->k, m.nd->tex, S);
			pdf = cosTSq <= 0.0f ? 1.0f : f;
		} else {
			o = (eta * iDotN - pm_sign(iDotN) * pm_sqrt(cosTSq)) * n - eta * i;
			kVal = eta * eta * mat_color(sf.uv, m.nd->t, m.nd->right_child, S);
			pdf = 1.0f - f;
		}
		return iDotN != 0.0f ? pdf * kVal / pm_fabs(iDotN) : splat(0.0f);
	}
	case POLARIS_BXDF_ROUGH_CONDUCTOR: { // rough_conductor.cl:10-40
		float a = alpha_of(sf, m, S);
		f3 ks = mat_color(sf.uv, m.nd->k, m.nd->tex, S);
		f3 h = ggx_sample(a, n, rnd);
		o = 2.0f * dot(i, h) * h - i;
		pdf = ggx_reflect_pdf(a, o, n, h);
		float iDotN = dot(i, n);
		h = normalize(i + o);
		float f = m.int_ior != 0.0f ? schlick(m.ext_ior, m.int_ior, iDotN) : 1.0f;
		return specular_tail(sf, a, ks, f, i, o, h);
	}
	case POLARIS_BXDF_ROUGH_DIELECTRIC: { // rough_dielectric.cl:10-94
		float iDotN = dot(i, n);
		float a = alpha_of(sf, m, S);
		float etaI = m.ext_ior, etaT = m.int_ior;
		if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
		float eta = etaI / etaT;
		f3 h = ggx_sample(a, n, rnd);
		float f = schlick(etaI, etaT, iDotN);
		float cosTSq = 1.0f + eta * (iDotN * iDotN - 1.0f);
		if (cosTSq <= 0.0f || rnd.x <= f) {
			o = 2.0f * dot(i, h) * h - i;
			f3 ks = mat_color(sf.uv, m.nd->k, m.nd->tex, S);
			h = normalize(i + o);
			pdf = cosTSq <= 0.0f ? 1.0f : ggx_reflect_pdf(a, o, n, h);
			return specular_tail(sf, a, ks, f, i, o, h);
		}
		o = (eta * iDotN - pm_sign(iDotN) * pm_sqrt(cosTSq)) * h - eta * i;
		h = normalize(-(etaI * i + etaT * o));
		pdf = ggx_refract_pdf(a, etaI, etaT, i, o, n, h);
		return transmit_tail(sf, m, S, a, etaI, etaT, f, iDotN, i, o, h);
	}
	}
	return splat(0.0f);
}

// bxdfGetPdf (bxdf.cl:58-78) and bxdfEval (bxdf.cl:82-105) for a given outgoing direction,
// evaluated together (the NEE path of shadeHits needs both for the same direction).
template <bool LDS> PD void bxdf_pdf_eval(const Surf &sf, const MatT<LDS> &m, const SceneT<LDS> &S, f3 i, f3 o, bool want_eval, float &pdf, f3 &val) {
	const f3 n = sf.n;
	pdf = 0.0f;
	val = splat(0.0f);
	switch (m.type) {
	case POLARIS_BXDF_DIFFUSE: // diffuse.cl:24-32
		pdf = dot(n, o) * kInvPi;
		if (want_eval) val = mat_color(sf.uv, m.nd->k, m.nd->tex, S) * kInvPi;
		return;
	case POLARIS_BXDF_CONDUCTOR: { // conductor.cl:33-62
		float iDotN = dot(i, n);
		f3 e = 2.0f * iDotN * n - i;
		float ed = dot(e, o);
		bool match = ed >= 0.0f && ed <= 0.001f;
		pdf = match ? 1.0f : 0.0f;
		if (want_eval && match) val = mirror_value(sf, m, S, iDotN);
		return;
	}
	case POLARIS_BXDF_DIELECTRIC: // dielectric.cl:49-60: always 0
		return;
	case POLARIS_BXDF_ROUGH_CONDUCTOR: { // rough_conductor.cl:43-78
		float a = alpha_of(sf, m, S);
		f3 h = normalize(i + o);
		pdf = ggx_reflect_pdf(a, o, n, h);
		if (want_eval) {
			f3 ks = mat_color(sf.uv, m.nd->k, m.nd->tex, S);
			float iDotN = dot(i, n);
			float f = m.int_ior != 0.0f ? schlick(m.ext_ior, m.int_ior, iDotN) : 1.0f;
			val = specular_tail(sf, a, ks, f, i, o, h);
		}
		return;
	}
	case POLARIS_BXDF_ROUGH_DIELECTRIC: { // rough_dielectric.cl:97-166
		float iDotN = dot(i, n);
		float a = alpha_of(sf, m, S);
		float etaI = m.ext_ior, etaT = m.int_ior;
		if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
		if (iDotN > 0.0f) {
			f3 h = normalize(i + o);
			pdf = ggx_reflect_pdf(a, o, n, h);
			if (want_eval) {
				float f = schlick(etaI, etaT, iDotN);
				f3 ks = mat_color(sf.uv, m.nd->k, m.nd->tex, S);
				val = specular_tail(sf, a, ks, f, i, o, h);
			}
		} else {
			f3 h = normalize(-(etaI * i + etaT * o));
			pdf = ggx_refract_pdf(a, etaI, etaT, i, o, n, h);
			if (want_eval) {
				float f = schlick(etaI, etaT, iDotN);
				val = transmit_tail(sf, m, S, a, etaI, etaT, f, iDotN, i, o, h);
			}
		}
		return;
	}
	}
}

// ---- lights: samplers/emissive_sampler.cl ---------------------------------------------
struct LightSample { f3 radiance, dir; float pdf, dist; };

template <bool LDS> PD LightSample light_sample(const Surf &sf, typename Tbl<LDS>::Light em, const SceneT<LDS> &S, f2 rnd) { // emissiveGetSample, :176-198
	LightSample L;
	typename Tbl<LDS>::Node mn = S.nodes + em->mat_node_index;
	if (em->type == POLARIS_EMISSIVE_ENVIRONMENT) { // :16-37
		L.dir = cosine_hemisphere(sf.n, rnd);
		L.pdf = pm_max(0.0f, dot(sf.n, L.dir)) * kInvPi;
		L.dist = kFltMax;
		f2 uv = latlong_uv(L.dir);
		L.radiance = mn->scale * mat_color(uv, mn->k, mn->tex, S) * kInvPi;
		return L;
	}
	if (em->type != POLARIS_EMISSIVE_AREA) return {splat(0.0f), splat(0.0f), 0.0f, 0.0f};
	// area light, :51-113 (normal goes through the point transform: quirk a-9(4) kept)
	float r1 = pm_sqrt(rnd.x);
	float ru = (1.0f - rnd.y) * r1, rv = rnd.y * r1;
	float w0 = 1.0f - ru - rv;
	uint32_t off = em->tri_index * 3;
	float4 a = S.vertices[off], b = S.vertices[off + 1], c = S.vertices[off + 2];
	f3 p = mk3(w0 * a.x + ru * b.x + rv * c.x, w0 * a.y + ru * b.y + rv * c.y, w0 * a.z + ru * b.z + rv * c.z);
	f3 ep = xform_point(p, em->transform);
	a = S.normals[off]; b = S.normals[off + 1]; c = S.normals[off + 2];
	f3 nn = mk3(w0 * a.x + ru * b.x + rv * c.x, w0 * a.y + ru * b.y + rv * c.y, w0 * a.z + ru * b.z + rv * c.z);
	f3 en = xform_point(nn, em->transform);
	float2 ua = S.uvs[off], ub = S.uvs[off + 1], uc = S.uvs[off + 2];
	f2 euv = {w0 * ua.x + ru * ub.x + rv * uc.x, w0 * ua.y + ru * ub.y + rv * uc.y};
	f3 er = ep - sf.p;
	float d2 = dot(er, er);
	L.dir = normalize(er);
	L.dist = pm_sqrt(d2);
	float nDotOut = dot(en, -L.dir);
	if (nDotOut > 0.0f) {
		L.pdf = 1.0f / em->area;
		f3 ke = mat_color(euv, mn->k, mn->tex, S);
		L.radiance = mn->scale * ke * nDotOut / d2;
	} else {
		L.pdf = 0.0f;
		L.radiance = splat(0.0f);
	}
	return L;
}

template <bool LDS> PD float light_pdf(const Surf &sf, typename Tbl<LDS>::Light em, const SceneT<LDS> &S, f3 o) { // emissiveGetPdf, :201-223
	if (em->type == POLARIS_EMISSIVE_ENVIRONMENT) return pm_max(0.0f, dot(sf.n, o) * kInvPi); // :39-47
	if (em->type != POLARIS_EMISSIVE_AREA) return 0.0f;
	// areaLightGetPdf, :117-173 (edges go through the point transform: quirk kept)
	uint32_t off = em->tri_index * 3;
	f3 v0 = xyz(S.vertices[off]);
	f3 e1 = xyz(S.vertices[off + 1]) - v0;
	f3 e2 = xyz(S.vertices[off + 2]) - v0;
	v0 = xform_point(v0, em->transform);
	e1 = xform_point(e1, em->transform);
	e2 = xform_point(e2, em->transform);
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
	f3 en = normalize(cross(e1, e2));
	float denom = em->area * pm_fabs(dot(en, o));
	return denom > 0.0f ? (t * t) / denom : 0.0f;
}

} // namespace pol
