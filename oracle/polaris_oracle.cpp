// polaris_oracle.cpp -- CPU restatement of the polaris tracer hot path.
//
// *** TEST INFRASTRUCTURE.  This is the parity ORACLE and the CPU baseline, not product
// *** code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// *** libpolaris_oracle.so.  The product path (polaris_amd/csrc, include/polaris_hip.h) never
// *** includes, links or calls anything in this directory.
//
// Parity status: PARITY UNPINNED (by the project's rubric).  The reference's own tests hold no
// golden vector for this path, its Go host cannot run here, and the image has no
// OpenCL runtime, so the reference cannot be built here without a stand-in built-in library.  What
// exists instead: (1) the built-ins of include/polaris_math.h measured against
// double-precision libm over all 2^32 inputs (tests/test_builtins_sweep.py); (2) the reference's OWN
// kernels compiled in place and linked with that built-in library (oracle/refbuild ->
// oracle/_ref/libpolaris_ref_pm.so): tests/test_oracle_vs_reference.py requires this restatement to
// reproduce its trace accumulator, ray counters and primary hit tables BIT FOR BIT on every synthetic
// scene; (3) the same kernels over glibc's libm, compared pixel by pixel; (4) tests/golden/*.npz keeps
// outputs of (2) so the cross-check also holds where /root/reference is absent (the GPU box).
//
// Every function cites the reference file:line it follows (paths relative to
// tracer/opencl/CL/ unless stated).  Arithmetic is IEEE binary32, evaluated in the order
// the reference writes it; the OpenCL built-ins are the single shared definition in
// include/polaris_math.h (dot = x*x'+y*y'+z*z' left to right, normalize = v * (1/sqrt(dot)),
// native_* = correctly rounded).  Compile with -ffp-contract=off, no fast-math.
//
// Execution model: work-items in ascending global id, work-group size 1 -- the reference's
// CPU-device behaviour (tracer/opencl/pipeline.go:105-111).  Rays are therefore compacted in
// stable order, which is what makes a run reproducible at all (the PRNG of shadeHits is
// seeded with the ray's position in the compacted buffer, kernels/pt_integrator.cl:81).
// OpenMP parallelises over work-items where the result does not depend on order; the
// compaction itself is an ordered prefix pass.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "oracle_api.h"
#include "polaris_math.h"

namespace {

// ---------------------------------------------------------------------------------------
// small vector helpers: every operator is the component-wise IEEE operation OpenCL defines
// ---------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct V2 { float x, y; };
struct V4 { float x, y, z, w; };

inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 v3s(float s) { return {s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator+(V3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline V3 operator-(V3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 operator*(float s, V4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator*(float s, V2 a) { return {s * a.x, s * a.y}; }

inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 v) { return pm_sqrt(dot(v, v)); }
inline V3 normalize(V3 v) { float inv = 1.0f / pm_sqrt(dot(v, v)); return {v.x * inv, v.y * inv, v.z * inv}; }
inline V4 normalize4(V4 v) {
	float inv = 1.0f / pm_sqrt(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
	return {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
}
inline V4 mix4(V4 a, V4 b, float t) { return {pm_mix(a.x, b.x, t), pm_mix(a.y, b.y, t), pm_mix(a.z, b.z, t), pm_mix(a.w, b.w, t)}; }
inline V3 xyz(V4 v) { return {v.x, v.y, v.z}; }
inline V4 ld4(const float *p) { return {p[0], p[1], p[2], p[3]}; }
inline V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
inline V2 ld2(const float *p) { return {p[0], p[1]}; }
inline float maxcomp(V3 v) { return pm_max(v.x, pm_max(v.y, v.z)); } // MAX_VEC3_COMPONENT, pt_integrator.cl:4

// constants.cl:6-24
constexpr float C_PI = 3.14159265358979323846f;
constexpr float C_TWO_TIMES_PI = 6.28318530718f;
constexpr float C_1_PI = 0.31830988618379067154f;
constexpr float INTERSECTION_EPSILON = 0.00001f;
constexpr float INTERSECTION_WITH_LIGHT_EPSILON = INTERSECTION_EPSILON * 1e3f;
constexpr float MIN_ROUGHNESS = 0.1f;
constexpr float FLT_MAX_ = 3.402823466e+38f;

// types.cl:4-24, 67-90
struct Ray { float origin[4]; float dir[4]; };
struct Path { float throughput[4]; uint32_t pixelIndex, flags, r1, r2; };
struct Intersection { float wuvt[4]; uint32_t meshInstance, triIndex, r1, r2; };
struct Surface { V3 point; V3 normal; V2 uv; uint32_t matNodeIndex; };
static_assert(sizeof(Ray) == 32 && sizeof(Path) == 32 && sizeof(Intersection) == 32, "layout");

typedef PolarisMaterialNode MatNode;
typedef PolarisTextureMetadata TexMeta;

// ---------------------------------------------------------------------------------------
// samplers/random_sampler.cl:7-16
// ---------------------------------------------------------------------------------------
inline V2 randomGetSample2f(uint32_t st[2]) {
	const float invMaxInt = 1.0f / 4294967296.0f;
	uint32_t x = st[0] * 17u + st[1] * 13123u;
	st[0] = (x << 13) ^ x;
	st[1] ^= (x << 7);
	uint32_t t0 = x * (x * x * 15731u + 74323u) + 871483u;
	uint32_t t1 = x * (x * x * 13734u + 37828u) + 234234u;
	return {(float)t0 * invMaxInt, (float)t1 * invMaxInt}; // convert_float2: round to nearest even
}

// ---------------------------------------------------------------------------------------
// util/transform.cl:9-37
// ---------------------------------------------------------------------------------------
inline V3 mul4x1(V3 v, const float *m /* 16 floats, column major: mat0..mat3 */) {
	V3 o;
	o.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12];
	o.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13];
	o.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14];
	return o;
}
inline V3 mul3x1(V3 v, const float *m) {
	V3 o;
	o.x = m[0] * v.x + m[4] * v.y + m[8] * v.z;
	o.y = m[1] * v.x + m[5] * v.y + m[9] * v.z;
	o.z = m[2] * v.x + m[6] * v.y + m[10] * v.z;
	return o;
}
inline V2 rayToLatLongUV(V3 vec) { // transform.cl:28-37
	float at2 = pm_atan2(vec.x, vec.z);
	float r = length(vec);
	return {(at2 >= 0.0f ? at2 : (at2 + C_TWO_TIMES_PI)) / C_TWO_TIMES_PI, pm_acos(vec.y / r) / C_PI};
}

// util/surface.cl:4-6
inline void tangentVectors(V3 normal, V3 &u, V3 &v) {
	u = normalize(cross((pm_fabs(normal.z) < .999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f)), normal));
	v = cross(normal, u);
}

// util/fresnel.cl:8-16 (Schlick)
inline float fresnelForDielectric(float etaI, float etaT, float iDotN) {
	float eta = etaI / etaT;
	float r0 = ((1.0f - eta) * (1.0f - eta)) / ((1.0f + eta) * (1.0f + eta));
	float c = 1.0f - pm_fabs(iDotN);
	float c1 = c * c;
	return r0 + (1.0f - r0) * c1 * c1 * c;
}

// ---------------------------------------------------------------------------------------
// samplers/texture_sampler.cl
// ---------------------------------------------------------------------------------------
struct TexAddr { uint32_t w, h, tx, ty, bx, by; float cx, cy; const uint8_t *base; uint32_t format; };

inline TexAddr texAddress(V2 uv, int texIndex, const TexMeta *meta, const uint8_t *data) { // texture_sampler.cl:15-38
	TexAddr a;
	a.w = meta[texIndex].width;
	a.h = meta[texIndex].height;
	V2 scaled = {uv.x - pm_floor(uv.x), uv.y - pm_floor(uv.y)};
	scaled.x *= (float)a.w;
	scaled.y *= (float)a.h;
	a.tx = pm_clampu((uint32_t)scaled.x, 0u, a.w - 1);
	a.ty = pm_clampu((uint32_t)scaled.y, 0u, a.h - 1);
	a.bx = pm_clampu(a.tx + 1, 0u, a.w - 1);
	a.by = pm_clampu(a.ty + 1, 0u, a.h - 1);
	a.cx = scaled.x - (float)a.tx;
	a.cy = scaled.y - (float)a.ty;
	a.base = data + meta[texIndex].data_offset;
	a.format = meta[texIndex].format;
	return a;
}
inline float ldf(const uint8_t *p, uint32_t i) { float f; memcpy(&f, p + 4 * (size_t)i, 4); return f; }

inline V3 texGetSample3f(V2 uv, int texIndex, const TexMeta *meta, const uint8_t *data) { // texture_sampler.cl:14-110
	TexAddr a = texAddress(uv, texIndex, meta, data);
	switch (a.format) {
	case POLARIS_TEX_RGBA8: {
		auto ld = [&](uint32_t y, uint32_t x) {
			const uint8_t *p = a.base + 4 * ((size_t)y * a.w + x);
			return V4{(float)p[0], (float)p[1], (float)p[2], (float)p[3]};
		};
		V4 r = mix4(mix4(ld(a.ty, a.tx), ld(a.by, a.tx), a.cy), mix4(ld(a.ty, a.bx), ld(a.by, a.bx), a.cy), a.cx);
		return xyz(r) / 255.0f;
	}
	case POLARIS_TEX_RGBA32F: {
		auto ld = [&](uint32_t y, uint32_t x) {
			uint32_t i = 4 * (y * a.w + x);
			return V4{ldf(a.base, i), ldf(a.base, i + 1), ldf(a.base, i + 2), ldf(a.base, i + 3)};
		};
		V4 r = mix4(mix4(ld(a.ty, a.tx), ld(a.by, a.tx), a.cy), mix4(ld(a.ty, a.bx), ld(a.by, a.bx), a.cy), a.cx);
		return xyz(r);
	}
	case POLARIS_TEX_L8: {
		float tl = (float)a.base[a.ty * a.w + a.tx], tr = (float)a.base[a.ty * a.w + a.bx];
		float bl = (float)a.base[a.by * a.w + a.tx], br = (float)a.base[a.by * a.w + a.bx];
		float r = pm_mix(pm_mix(tl, bl, a.cy), pm_mix(tr, br, a.cy), a.cx) / 255.0f;
		return v3s(r);
	}
	case POLARIS_TEX_L32F: {
		float tl = ldf(a.base, a.ty * a.w + a.tx), tr = ldf(a.base, a.ty * a.w + a.bx);
		float bl = ldf(a.base, a.by * a.w + a.tx), br = ldf(a.base, a.by * a.w + a.bx);
		float r = pm_mix(pm_mix(tl, bl, a.cy), pm_mix(tr, br, a.cy), a.cx);
		return v3s(r);
	}
	}
	return v3s(0.0f);
}

inline float texGetSample1f(V2 uv, int texIndex, const TexMeta *meta, const uint8_t *data) { // texture_sampler.cl:114-184
	TexAddr a = texAddress(uv, texIndex, meta, data);
	float tl, tr, bl, br;
	switch (a.format) {
	case POLARIS_TEX_RGBA8:
		tl = (float)a.base[(a.ty * a.w << 2) + (a.tx << 2)]; tr = (float)a.base[(a.ty * a.w << 2) + (a.bx << 2)];
		bl = (float)a.base[(a.by * a.w << 2) + (a.tx << 2)]; br = (float)a.base[(a.by * a.w << 2) + (a.bx << 2)];
		return pm_mix(pm_mix(tl, bl, a.cy), pm_mix(tr, br, a.cy), a.cx) / 255.0f;
	case POLARIS_TEX_RGBA32F:
		tl = ldf(a.base, (a.ty * a.w << 2) + (a.tx << 2)); tr = ldf(a.base, (a.ty * a.w << 2) + (a.bx << 2));
		bl = ldf(a.base, (a.by * a.w << 2) + (a.tx << 2)); br = ldf(a.base, (a.by * a.w << 2) + (a.bx << 2));
		return pm_mix(pm_mix(tl, bl, a.cy), pm_mix(tr, br, a.cy), a.cx);
	case POLARIS_TEX_L8:
		tl = (float)a.base[a.ty * a.w + a.tx]; tr = (float)a.base[a.ty * a.w + a.bx];
		bl = (float)a.base[a.by * a.w + a.tx]; br = (float)a.base[a.by * a.w + a.bx];
		return pm_mix(pm_mix(tl, bl, a.cy), pm_mix(tr, br, a.cy), a.cx) / 255.0f;
	case POLARIS_TEX_L32F:
		tl = ldf(a.base, a.ty * a.w + a.tx); tr = ldf(a.base, a.ty * a.w + a.bx);
		bl = ldf(a.base, a.by * a.w + a.tx); br = ldf(a.base, a.by * a.w + a.bx);
		return pm_mix(pm_mix(tl, bl, a.cy), pm_mix(tr, br, a.cy), a.cx);
	}
	return 0.0f;
}

inline V3 texGetBumpSample3f(V2 uv, int texIndex, const TexMeta *meta, const uint8_t *data) { // texture_sampler.cl:187-252
	TexAddr a = texAddress(uv, texIndex, meta, data);
	float s0, s1, s2;
	switch (a.format) {
	case POLARIS_TEX_RGBA8:
		s0 = (float)a.base[4 * (a.ty * a.w + a.tx)] / 255.0f;
		s1 = (float)a.base[4 * (a.ty * a.w + a.bx)] / 255.0f;
		s2 = (float)a.base[4 * (a.by * a.w + a.tx)] / 255.0f;
		break;
	case POLARIS_TEX_RGBA32F:
		s0 = ldf(a.base, 4 * (a.ty * a.w + a.tx));
		s1 = ldf(a.base, 4 * (a.ty * a.w + a.bx));
		s2 = ldf(a.base, 4 * (a.by * a.w + a.tx));
		break;
	case POLARIS_TEX_L8:
		s0 = (float)a.base[a.ty * a.w + a.tx] / 255.0f;
		s1 = (float)a.base[a.ty * a.w + a.bx] / 255.0f;
		s2 = (float)a.base[a.by * a.w + a.tx] / 255.0f;
		break;
	case POLARIS_TEX_L32F:
		s0 = ldf(a.base, a.ty * a.w + a.tx);
		s1 = ldf(a.base, a.ty * a.w + a.bx);
		s2 = ldf(a.base, a.by * a.w + a.tx);
		break;
	default:
		return v3s(0.0f);
	}
	return v3s(0.5f) + 0.5f * normalize(v3(s1 - s0, s2 - s0, 1.0f));
}

// samplers/material_sampler.cl:97-131
inline V3 matGetSample3f(V2 uv, V3 def, int texIndex, const TexMeta *meta, const uint8_t *data) {
	return texIndex == -1 ? def : texGetSample3f(uv, texIndex, meta, data);
}
inline float matGetSample1f(V2 uv, float def, int texIndex, const TexMeta *meta, const uint8_t *data) {
	return texIndex == -1 ? def : texGetSample1f(uv, texIndex, meta, data);
}
inline V3 matGetNormalSample3f(V3 normal, V2 uv, int texIndex, const TexMeta *meta, const uint8_t *data) {
	V3 u, v;
	tangentVectors(normal, u, v);
	V3 s = (texGetSample3f(uv, texIndex, meta, data) * 2.0f) - 1.0f;
	return normalize(u * s.x + v * s.y + 0.5f * normal * s.z);
}
inline V3 matGetBumpSample3f(V3 normal, V2 uv, int texIndex, const TexMeta *meta, const uint8_t *data) {
	V3 u, v;
	tangentVectors(normal, u, v);
	V3 s = (texGetBumpSample3f(uv, texIndex, meta, data) * 2.0f) - 1.0f;
	return normalize(u * s.x + v * s.y + normal * s.z);
}

inline V3 K(const MatNode &m) { return ld3(m.k); }
inline V3 TT(const MatNode &m) { return ld3(m.t); }

// samplers/material_sampler.cl:21-95.  PATH_FLAG_DISPERSE_R/G/B = 1,2,4 (util/path.cl:4-6)
inline void matSelectNode(Path *path, Surface *surface, V3 /*inRayDir*/, MatNode *selected, V3 *tint,
                          const MatNode *nodes, uint32_t rnd[2], const TexMeta *meta, const uint8_t *data) {
	const MatNode *node = nodes + surface->matNodeIndex;
	V2 sample;
	V2 forceIOR = {0.0f, 0.0f};
	while (node->type >= POLARIS_MAT_OP_MIX) {
		switch (node->type) {
		case POLARIS_MAT_OP_MIX:
			sample = randomGetSample2f(rnd);
			node = nodes + (sample.x < node->k[0] ? node->left_child : (uint32_t)node->right_child);
			break;
		case POLARIS_MAT_OP_MIX_MAP:
			sample = randomGetSample2f(rnd);
			sample.y = texGetSample1f(surface->uv, node->tex, meta, data);
			node = nodes + (sample.x < sample.y ? node->left_child : (uint32_t)node->right_child);
			break;
		case POLARIS_MAT_OP_BUMP_MAP:
			surface->normal = matGetBumpSample3f(surface->normal, surface->uv, node->tex, meta, data);
			node = nodes + node->left_child;
			break;
		case POLARIS_MAT_OP_NORMAL_MAP:
			surface->normal = matGetNormalSample3f(surface->normal, surface->uv, node->tex, meta, data);
			node = nodes + node->left_child;
			break;
		case POLARIS_MAT_OP_DISPERSE: {
			uint32_t flags = path->flags;
			if ((flags & 1u) != 0) {
				*tint = v3(1.0f, 0.0f, 0.0f);
				forceIOR = {node->k[0], node->t[0]};
			} else if ((flags & 2u) != 0) {
				*tint = v3(0.0f, 1.0f, 0.0f);
				forceIOR = {node->k[1], node->t[1]};
			} else if ((flags & 4u) != 0) {
				*tint = v3(0.0f, 0.0f, 1.0f);
				forceIOR = {node->k[2], node->t[2]};
			} else {
				sample = randomGetSample2f(rnd);
				if (sample.x < 0.333f) {
					*tint = v3(1.0f, 0.0f, 0.0f);
					forceIOR = {node->k[0], node->t[0]};
					path->flags |= 1u;
				} else if (sample.x < 0.666f) {
					*tint = v3(0.0f, 1.0f, 0.0f);
					forceIOR = {node->k[1], node->t[1]};
					path->flags |= 2u;
				} else {
					*tint = v3(0.0f, 0.0f, 1.0f);
					forceIOR = {node->k[2], node->t[2]};
					path->flags |= 4u;
				}
			}
			node = nodes + node->left_child;
			break;
		}
		default: // unknown operator: the reference would spin forever; stop instead
			*selected = *node;
			selected->type = POLARIS_BXDF_INVALID;
			return;
		}
	}
	*selected = *node;
	selected->int_ior = pm_max(selected->int_ior, forceIOR.x);
	selected->ext_ior = pm_max(selected->ext_ior, forceIOR.y);
}

// ---------------------------------------------------------------------------------------
// samplers/distribution_sampler.cl
// ---------------------------------------------------------------------------------------
inline float ggxGetG1(float roughness, V3 v, V3 n, V3 m) { // :20-33
	float nDotV = dot(n, v);
	float mDotV = dot(m, v);
	if (nDotV * mDotV <= 0.0f) return 0.0f;
	float nDotVSq = nDotV * nDotV;
	float tanSq = nDotVSq > 0.0f ? (1.0f - nDotVSq) / nDotVSq : 0.0f;
	float aSq = roughness * roughness;
	return 2.0f / (1.0f + pm_sqrt(1.0f + aSq * tanSq));
}
inline float ggxGetG(float roughness, V3 in, V3 out, V3 n, V3 m) { return ggxGetG1(roughness, in, n, m) * ggxGetG1(roughness, out, n, m); } // :37-39
inline float ggxGetD(float roughness, V3 n, V3 m) { // :42-56
	float nDotM = dot(n, m);
	if (nDotM <= 0.0f) return 0.0f;
	float nDotMSq = nDotM * nDotM;
	float tanSq = nDotM != 0.0f ? ((1.0f - nDotMSq) / nDotMSq) : 0.0f;
	float aSq = roughness * roughness;
	float denom = C_PI * nDotMSq * nDotMSq * (aSq + tanSq) * (aSq + tanSq);
	return denom > 0.0f ? (aSq / denom) : 0.0f;
}
inline V3 ggxGetSample(float roughness, V3 /*in*/, V3 n, V2 rnd) { // :59-76
	V3 u, v;
	tangentVectors(n, u, v);
	float theta = pm_atan(roughness * pm_sqrt(rnd.x / (1.0f - rnd.x)));
	theta = theta >= 0.0f ? theta : (theta + C_TWO_TIMES_PI);
	float cosTheta = pm_cos(theta);
	float sinTheta = pm_sqrt(1.0f - cosTheta * cosTheta);
	float cosPhi = pm_cos(C_TWO_TIMES_PI * rnd.y);
	float sinPhi = pm_sqrt(1.0f - cosPhi * cosPhi);
	return normalize(u * sinTheta * cosPhi + v * sinTheta * sinPhi + n * cosTheta);
}
inline float ggxGetReflectionPdf(float roughness, V3 /*in*/, V3 out, V3 n, V3 h) { // :78-87
	float nDotH = pm_fabs(dot(n, h));
	float oDotH = pm_fabs(dot(out, h));
	float denom = 4.0f * oDotH;
	return denom == 0.0f ? 0.0f : ggxGetD(roughness, n, h) * nDotH / denom;
}
inline float ggxGetRefractionPdf(float roughness, float etaI, float etaT, V3 in, V3 out, V3 n, V3 h) { // :89-98
	float iDotH = pm_fabs(dot(in, h));
	float oDotH = pm_fabs(dot(out, h));
	float hDotN = pm_fabs(dot(h, n));
	float denom = (etaI * iDotH + etaT * oDotH) * (etaI * iDotH + etaT * oDotH);
	return denom > 0.0f ? ggxGetD(roughness, n, h) * hDotN * oDotH * etaT * etaT / denom : 0.0f;
}
inline V3 cosWeightedHemisphereGetSample(V3 normal, V2 rnd) { // :101-112
	float rd = pm_sqrt(rnd.x);
	float phi = C_TWO_TIMES_PI * rnd.y;
	V3 u, v;
	tangentVectors(normal, u, v);
	return normalize(u * rd * pm_cos(phi) + v * rd * pm_sin(phi) + normal * pm_sqrt(1 - rnd.x));
}

// ---------------------------------------------------------------------------------------
// bxdf/*.cl
// ---------------------------------------------------------------------------------------
inline V3 zero3() { return v3s(0.0f); }

// bxdf/diffuse.cl:12-32
inline V3 diffuseSample(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V2 rnd, V3 *out, float *pdf) {
	*out = cosWeightedHemisphereGetSample(s->normal, rnd);
	*pdf = dot(s->normal, *out) * C_1_PI;
	V3 kd = matGetSample3f(s->uv, K(*m), m->tex, tm, td);
	return kd * C_1_PI;
}
inline float diffusePdf(Surface *s, V3 out) { return dot(s->normal, out) * C_1_PI; }
inline V3 diffuseEval(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td) { return matGetSample3f(s->uv, K(*m), m->tex, tm, td) * C_1_PI; }

// bxdf/conductor.cl:12-62
inline V3 conductorValue(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, float iDotN) {
	float f = m->int_ior != 0.0f ? fresnelForDielectric(m->ext_ior, m->int_ior, iDotN) : 1.0f;
	V3 ks = matGetSample3f(s->uv, K(*m), m->tex, tm, td);
	return iDotN != 0.0f ? f * ks / iDotN : zero3();
}
inline V3 conductorSample(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V2, V3 in, V3 *out, float *pdf) {
	float iDotN = dot(in, s->normal);
	*out = 2.0f * iDotN * s->normal - in;
	*pdf = 1.0f;
	return conductorValue(s, m, tm, td, iDotN);
}
inline float conductorPdf(Surface *s, V3 in, V3 out) {
	float iDotN = dot(in, s->normal);
	V3 exp = 2.0f * iDotN * s->normal - in;
	float expDot = dot(exp, out);
	return expDot >= 0.0f && expDot <= 0.001f ? 1.0f : 0.0f;
}
inline V3 conductorEval(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V3 in, V3 out) {
	float iDotN = dot(in, s->normal);
	V3 exp = 2.0f * iDotN * s->normal - in;
	float expDot = dot(exp, out);
	if (expDot < 0.0f || expDot > 0.001f) return zero3();
	return conductorValue(s, m, tm, td, iDotN);
}

// bxdf/dielectric.cl:12-60
inline V3 dielectricSample(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V2 rnd, V3 in, V3 *out, float *pdf) {
	float iDotN = dot(in, s->normal);
	float etaI = m->ext_ior, etaT = m->int_ior;
	if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
	float eta = etaI / etaT;
	float f = fresnelForDielectric(etaI, etaT, iDotN);
	V3 kVal;
	float cosTSq = 1.0f + eta * (iDotN * iDotN - 1.0f);
	if (cosTSq <= 0.0f || rnd.x <= f) {
		*out = -pm_sign(iDotN) * 2.0f * iDotN * s->normal - in;
		kVal = matGetSample3f(s->uv, K(*m), m->tex, tm, td);
		*pdf = cosTSq <= 0.0f ? 1.0f : f;
	} else {
		*out = (eta * iDotN - pm_sign(iDotN) * pm_sqrt(cosTSq)) * s->normal - eta * in;
		kVal = eta * eta * matGetSample3f(s->uv, TT(*m), m->right_child, tm, td);
		*pdf = 1.0f - f;
	}
	return iDotN != 0.0f ? *pdf * kVal / pm_fabs(iDotN) : zero3();
}

// bxdf/rough_conductor.cl
inline float roughnessOf(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td) {
	float r = pm_clamp(matGetSample1f(s->uv, m->scale, m->roughness_tex, tm, td), MIN_ROUGHNESS, 1.0f);
	return r * r;
}
inline V3 microfacetReflect(Surface *s, MatNode *m, float roughness, V3 ks, float f, V3 in, V3 out, V3 h) {
	float iDotN = dot(in, s->normal);
	float oDotN = dot(out, s->normal);
	float d = ggxGetD(roughness, s->normal, h);
	float g = ggxGetG(roughness, in, out, s->normal, h);
	float denom = 4.0f * iDotN * oDotN;
	(void)m;
	return denom > 0.0f ? ks * f * d * g / denom : zero3();
}
inline V3 roughConductorSample(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V2 rnd, V3 in, V3 *out, float *pdf) { // :10-40
	float roughness = roughnessOf(s, m, tm, td);
	V3 ks = matGetSample3f(s->uv, K(*m), m->tex, tm, td);
	V3 h = ggxGetSample(roughness, in, s->normal, rnd);
	*out = 2.0f * dot(in, h) * h - in;
	*pdf = ggxGetReflectionPdf(roughness, in, *out, s->normal, h);
	float iDotN = dot(in, s->normal);
	h = normalize(in + *out);
	float f = m->int_ior != 0.0f ? fresnelForDielectric(m->ext_ior, m->int_ior, iDotN) : 1.0f;
	return microfacetReflect(s, m, roughness, ks, f, in, *out, h);
}
inline float roughConductorPdf(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V3 in, V3 out) { // :43-51
	float roughness = roughnessOf(s, m, tm, td);
	V3 h = normalize(in + out);
	return ggxGetReflectionPdf(roughness, in, out, s->normal, h);
}
inline V3 roughConductorEval(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V3 in, V3 out) { // :54-78
	float roughness = roughnessOf(s, m, tm, td);
	V3 ks = matGetSample3f(s->uv, K(*m), m->tex, tm, td);
	float iDotN = dot(in, s->normal);
	float f = m->int_ior != 0.0f ? fresnelForDielectric(m->ext_ior, m->int_ior, iDotN) : 1.0f;
	V3 h = normalize(in + out);
	return microfacetReflect(s, m, roughness, ks, f, in, out, h);
}

// bxdf/rough_dielectric.cl
inline V3 roughTransmit(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, float roughness, float etaI, float etaT,
                        float f, float iDotN, V3 in, V3 out, V3 h) { // shared tail of :73-93 and :146-165
	float iDotH = pm_fabs(dot(in, h));
	float oDotH = pm_fabs(dot(out, h));
	float oDotN = dot(out, s->normal);
	float focusTermDenom = iDotN * oDotN * (etaI * iDotH + etaT * oDotH) * (etaI * iDotH + etaT * oDotH);
	if (focusTermDenom == 0.0f) return zero3();
	float focusTerm = pm_fabs(etaT * etaT * iDotH * oDotH / focusTermDenom);
	float d = ggxGetD(roughness, s->normal, h);
	float g = ggxGetG(roughness, in, out, s->normal, h);
	V3 tf = matGetSample3f(s->uv, TT(*m), m->right_child, tm, td);
	return tf * (1.0f - f) * d * g * focusTerm;
}
inline V3 roughDielectricSample(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V2 rnd, V3 in, V3 *out, float *pdf) { // :10-94
	float iDotN = dot(in, s->normal);
	float roughness = roughnessOf(s, m, tm, td);
	float etaI = m->ext_ior, etaT = m->int_ior;
	if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
	float eta = etaI / etaT;
	V3 h = ggxGetSample(roughness, in, s->normal, rnd);
	float f = fresnelForDielectric(etaI, etaT, iDotN);
	float cosTSq = 1.0f + eta * (iDotN * iDotN - 1.0f);
	if (cosTSq <= 0.0f || rnd.x <= f) {
		*out = 2.0f * dot(in, h) * h - in;
		V3 ks = matGetSample3f(s->uv, K(*m), m->tex, tm, td);
		h = normalize(in + *out);
		*pdf = cosTSq <= 0.0f ? 1.0f : ggxGetReflectionPdf(roughness, in, *out, s->normal, h);
		return microfacetReflect(s, m, roughness, ks, f, in, *out, h);
	}
	*out = (eta * iDotN - pm_sign(iDotN) * pm_sqrt(cosTSq)) * h - eta * in;
	h = normalize(-(etaI * in + etaT * *out));
	*pdf = ggxGetRefractionPdf(roughness, etaI, etaT, in, *out, s->normal, h);
	return roughTransmit(s, m, tm, td, roughness, etaI, etaT, f, iDotN, in, *out, h);
}
inline float roughDielectricPdf(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V3 in, V3 out) { // :97-121
	float iDotN = dot(in, s->normal);
	float roughness = roughnessOf(s, m, tm, td);
	if (iDotN > 0.0f) {
		V3 h = normalize(in + out);
		return ggxGetReflectionPdf(roughness, in, out, s->normal, h);
	}
	float etaI = m->ext_ior, etaT = m->int_ior;
	if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
	V3 h = normalize(-(etaI * in + etaT * out));
	return ggxGetRefractionPdf(roughness, etaI, etaT, in, out, s->normal, h);
}
inline V3 roughDielectricEval(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V3 in, V3 out) { // :124-166
	float iDotN = dot(in, s->normal);
	float roughness = roughnessOf(s, m, tm, td);
	float etaI = m->ext_ior, etaT = m->int_ior;
	if (iDotN < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
	float f = fresnelForDielectric(etaI, etaT, iDotN);
	if (iDotN > 0.0f) {
		V3 ks = matGetSample3f(s->uv, K(*m), m->tex, tm, td);
		V3 h = normalize(in + out);
		return microfacetReflect(s, m, roughness, ks, f, in, out, h);
	}
	V3 h = normalize(-(etaI * in + etaT * out));
	return roughTransmit(s, m, tm, td, roughness, etaI, etaT, f, iDotN, in, out, h);
}

// bxdf/bxdf.cl:31-105
inline V3 bxdfGetSample(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V2 rnd, V3 in, V3 *out, float *pdf) {
	switch (m->type) {
	case POLARIS_BXDF_DIFFUSE: return diffuseSample(s, m, tm, td, rnd, out, pdf);
	case POLARIS_BXDF_CONDUCTOR: return conductorSample(s, m, tm, td, rnd, in, out, pdf);
	case POLARIS_BXDF_DIELECTRIC: return dielectricSample(s, m, tm, td, rnd, in, out, pdf);
	case POLARIS_BXDF_ROUGH_CONDUCTOR: return roughConductorSample(s, m, tm, td, rnd, in, out, pdf);
	case POLARIS_BXDF_ROUGH_DIELECTRIC: return roughDielectricSample(s, m, tm, td, rnd, in, out, pdf);
	}
	return zero3();
}
inline float bxdfGetPdf(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V3 in, V3 out) {
	switch (m->type) {
	case POLARIS_BXDF_DIFFUSE: return diffusePdf(s, out);
	case POLARIS_BXDF_CONDUCTOR: return conductorPdf(s, in, out);
	case POLARIS_BXDF_DIELECTRIC: return 0.0f; // dielectric.cl:50-53
	case POLARIS_BXDF_ROUGH_CONDUCTOR: return roughConductorPdf(s, m, tm, td, in, out);
	case POLARIS_BXDF_ROUGH_DIELECTRIC: return roughDielectricPdf(s, m, tm, td, in, out);
	}
	return 0.0f;
}
inline V3 bxdfEval(Surface *s, MatNode *m, const TexMeta *tm, const uint8_t *td, V3 in, V3 out) {
	switch (m->type) {
	case POLARIS_BXDF_DIFFUSE: return diffuseEval(s, m, tm, td);
	case POLARIS_BXDF_CONDUCTOR: return conductorEval(s, m, tm, td, in, out);
	case POLARIS_BXDF_DIELECTRIC: return zero3(); // dielectric.cl:58-60
	case POLARIS_BXDF_ROUGH_CONDUCTOR: return roughConductorEval(s, m, tm, td, in, out);
	case POLARIS_BXDF_ROUGH_DIELECTRIC: return roughDielectricEval(s, m, tm, td, in, out);
	}
	return zero3();
}

// ---------------------------------------------------------------------------------------
// samplers/emissive_sampler.cl
// ---------------------------------------------------------------------------------------
struct SceneRefs {
	const PolarisSceneView *sc;
	const float *vertices, *normals, *uvs;
	const MatNode *nodes;
	const TexMeta *tm;
	const uint8_t *td;
};

inline V3 environmentLightGetSample(Surface *s, const PolarisEmissive *em, const SceneRefs &R, V2 rnd, V3 *out, float *pdf, float *dist) { // :16-37
	*out = cosWeightedHemisphereGetSample(s->normal, rnd);
	*pdf = pm_max(0.0f, dot(s->normal, *out)) * C_1_PI;
	*dist = FLT_MAX_;
	V2 uv = rayToLatLongUV(*out);
	MatNode m = R.nodes[em->mat_node_index];
	return m.scale * matGetSample3f(uv, K(m), m.tex, R.tm, R.td) * C_1_PI;
}
inline float environmentLightGetPdf(Surface *s, V3 out) { return pm_max(0.0f, dot(s->normal, out) * C_1_PI); } // :39-47

inline V3 areaLightGetSample(Surface *s, const PolarisEmissive *em, const SceneRefs &R, V2 rnd, V3 *out, float *pdf, float *dist) { // :51-113
	float r1sqrt = pm_sqrt(rnd.x);
	float ru = (1.0f - rnd.y) * r1sqrt;
	float rv = rnd.y * r1sqrt;
	V3 wuv = v3(1.0f - ru - rv, ru, rv);
	size_t off = (size_t)em->tri_index * 3;
	V3 p = xyz(wuv.x * ld4(R.vertices + 4 * off) + wuv.y * ld4(R.vertices + 4 * (off + 1)) + wuv.z * ld4(R.vertices + 4 * (off + 2)));
	V3 emissivePoint = mul4x1(p, em->transform);
	V3 nn = xyz(wuv.x * ld4(R.normals + 4 * off) + wuv.y * ld4(R.normals + 4 * (off + 1)) + wuv.z * ld4(R.normals + 4 * (off + 2)));
	V3 emissiveNormal = mul4x1(nn, em->transform);
	V2 emissiveUV = wuv.x * ld2(R.uvs + 2 * off) + wuv.y * ld2(R.uvs + 2 * (off + 1)) + wuv.z * ld2(R.uvs + 2 * (off + 2));
	MatNode m = R.nodes[em->mat_node_index];
	V3 emissiveRay = emissivePoint - s->point;
	float squaredDistToLight = dot(emissiveRay, emissiveRay);
	*out = normalize(emissiveRay);
	*dist = pm_sqrt(squaredDistToLight);
	float nDotOutRay = dot(emissiveNormal, -*out);
	if (nDotOutRay > 0.0f) {
		*pdf = 1.0f / em->area;
		V3 ke = matGetSample3f(emissiveUV, K(m), m.tex, R.tm, R.td);
		return m.scale * ke * nDotOutRay / squaredDistToLight;
	}
	*pdf = 0.0f;
	return zero3();
}
inline float areaLightGetPdf(Surface *s, const PolarisEmissive *em, const SceneRefs &R, V3 outRayDir) { // :117-173
	size_t off = (size_t)em->tri_index * 3;
	V3 v0 = ld3(R.vertices + 4 * off);
	V3 edge01 = ld3(R.vertices + 4 * (off + 1)) - v0;
	V3 edge02 = ld3(R.vertices + 4 * (off + 2)) - v0;
	v0 = mul4x1(v0, em->transform);
	edge01 = mul4x1(edge01, em->transform);
	edge02 = mul4x1(edge02, em->transform);
	V3 pVec = cross(outRayDir, edge02);
	float det = dot(edge01, pVec);
	if (pm_fabs(det) < INTERSECTION_EPSILON) return 0.0f;
	float invDet = pm_rcp(det);
	V3 tVec = s->point - v0;
	float u = dot(tVec, pVec) * invDet;
	if (u < 0.0f || u > 1.0f) return 0.0f;
	V3 qVec = cross(tVec, edge01);
	float v = dot(outRayDir, qVec) * invDet;
	if (v < 0.0f || u + v > 1.0f) return 0.0f;
	float t = dot(edge02, qVec) * invDet;
	if (t < INTERSECTION_EPSILON) return 0.0f;
	V3 emissiveNormal = normalize(cross(edge01, edge02));
	float denominator = em->area * pm_fabs(dot(emissiveNormal, outRayDir));
	return denominator > 0.0f ? (t * t) / denominator : 0.0f;
}
inline V3 emissiveGetSample(Surface *s, const PolarisEmissive *em, const SceneRefs &R, V2 rnd, V3 *out, float *pdf, float *dist) { // :176-198
	switch (em->type) {
	case POLARIS_EMISSIVE_AREA: return areaLightGetSample(s, em, R, rnd, out, pdf, dist);
	case POLARIS_EMISSIVE_ENVIRONMENT: return environmentLightGetSample(s, em, R, rnd, out, pdf, dist);
	}
	return zero3();
}
inline float emissiveGetPdf(Surface *s, const PolarisEmissive *em, const SceneRefs &R, V3 out) { // :201-223
	switch (em->type) {
	case POLARIS_EMISSIVE_AREA: return areaLightGetPdf(s, em, R, out);
	case POLARIS_EMISSIVE_ENVIRONMENT: return environmentLightGetPdf(s, out);
	}
	return 0.0f;
}
inline int emissiveSelect(int numLights, float rnd, float *pdf) { // :226-237
	*pdf = pm_rcp((float)numLights);
	return pm_clampi((int)(rnd * numLights), 0, numLights - 1);
}

// util/surface.cl:12-33
inline void surfaceInit(Surface *s, const Intersection *is, const SceneRefs &R) {
	V3 wuv = ld3(is->wuvt);
	size_t off = (size_t)is->triIndex * 3;
	s->point = xyz(wuv.x * ld4(R.vertices + 4 * off) + wuv.y * ld4(R.vertices + 4 * (off + 1)) + wuv.z * ld4(R.vertices + 4 * (off + 2)));
	s->normal = normalize(xyz(wuv.x * ld4(R.normals + 4 * off) + wuv.y * ld4(R.normals + 4 * (off + 1)) + wuv.z * ld4(R.normals + 4 * (off + 2))));
	s->uv = wuv.x * ld2(R.uvs + 2 * off) + wuv.y * ld2(R.uvs + 2 * (off + 1)) + wuv.z * ld2(R.uvs + 2 * (off + 2));
	s->matNodeIndex = R.sc->material_index[is->triIndex];
}

// ---------------------------------------------------------------------------------------
// kernels/camera.cl:5-58
// ---------------------------------------------------------------------------------------
inline void generatePrimaryRay(uint32_t gx, uint32_t gy, Ray *rays, Path *paths, const float fr[16], V3 eye, V2 texelDims,
                               uint32_t blockY, uint32_t frameW, uint32_t seed) {
	uint32_t index = gy * frameW + gx;
	uint32_t pixelIndex = (gy + blockY) * frameW + gx;
	uint32_t rnd[2] = {gx + seed, gy + seed};
	V2 s0 = randomGetSample2f(rnd);
	V2 offset = {s0.x < 0.5f ? pm_sqrt(2.0f * s0.x) - 0.5f : 1.5f - pm_sqrt(2.0f - 2.0f * s0.x),
	             s0.y < 0.5f ? pm_sqrt(2.0f * s0.y) - 0.5f : 1.5f - pm_sqrt(2.0f - 2.0f * s0.y)};
	V2 texel = {((float)gx + offset.x) * texelDims.x, ((float)(gy + blockY) + offset.y) * texelDims.y};
	V4 dir = normalize4(mix4(mix4(ld4(fr), ld4(fr + 8), texel.y), mix4(ld4(fr + 4), ld4(fr + 12), texel.y), texel.x));
	Ray &r = rays[index]; // rayNew, util/ray.cl:9-12
	r.origin[0] = eye.x; r.origin[1] = eye.y; r.origin[2] = eye.z; r.origin[3] = FLT_MAX_;
	r.dir[0] = dir.x; r.dir[1] = dir.y; r.dir[2] = dir.z; r.dir[3] = (float)index;
	Path &p = paths[index]; // pathNew, util/path.cl:13-17
	p.throughput[0] = p.throughput[1] = p.throughput[2] = 1.0f;
	p.pixelIndex = pixelIndex;
	p.flags = 0;
}

// ---------------------------------------------------------------------------------------
// kernels/intersect.cl:26-180 (any hit) and :184-347 (closest hit): one traversal, two exits
// ---------------------------------------------------------------------------------------
constexpr int BVH_MAX_STACK_SIZE = 32; // intersect.cl:4

inline float slab(const PolarisBvhNode &c, V3 o, V3 invDir, float maxDist) { // intersect.cl:301-309
	V3 tmin = (ld3(c.min) - o) * invDir;
	V3 tmax = (ld3(c.max) - o) * invDir;
	V3 rmin = {pm_fmin(tmin.x, tmax.x), pm_fmin(tmin.y, tmax.y), pm_fmin(tmin.z, tmax.z)};
	V3 rmax = {pm_fmax(tmin.x, tmax.x), pm_fmax(tmin.y, tmax.y), pm_fmax(tmin.z, tmax.z)};
	float minmax = pm_fmin(pm_fmin(rmax.x, rmax.y), rmax.z);
	float maxmin = pm_fmax(pm_fmax(rmin.x, rmin.y), rmin.z);
	return minmax < 0 || maxmin > minmax ? FLT_MAX_ : (maxmin >= maxDist ? FLT_MAX_ : maxmin);
}

template <bool ANY_HIT>
inline int traverse(const Ray &ray, const PolarisSceneView *sc, Intersection *isOut, bool *overflow) {
	const PolarisBvhNode *bvh = sc->bvh_nodes;
	const float *verts = sc->vertices;
	uint32_t nodeStack[BVH_MAX_STACK_SIZE];
	int stackIndex = 0, meshStart = -1;
	V3 o = ld3(ray.origin), d = ld3(ray.dir);
	const V3 origO = o, origD = d;
	const float maxDist = ray.origin[3];
	Intersection is;
	memset(&is, 0, sizeof is);
	is.wuvt[3] = maxDist;
	uint32_t meshInstanceId = 0;
	PolarisBvhNode cur = bvh[0], child[2];
	int wantLeft, wantRight, gotHit = 0;
	child[0] = child[1] = cur;
	while (stackIndex > -1) {
		if (cur.ldata <= 0) {
			int numTriangles = cur.rdata;
			if (numTriangles == 0) { // top-level leaf: enter the instance (intersect.cl:77-89)
				meshInstanceId = (uint32_t)(-cur.ldata);
				const PolarisMeshInstance &mi = sc->mesh_instances[meshInstanceId];
				meshStart = stackIndex;
				if (stackIndex >= BVH_MAX_STACK_SIZE) { *overflow = true; break; }
				nodeStack[stackIndex++] = mi.bvh_root;
				o = mul4x1(o, mi.inv_transform);
				d = mul3x1(d, mi.inv_transform);
			} else {
				int triStart = -cur.ldata;
				for (int vIndex = triStart * 3; vIndex < (triStart + numTriangles) * 3; vIndex += 3) { // :92-130
					V3 v0 = ld3(verts + 4 * (size_t)vIndex);
					V3 edge01 = ld3(verts + 4 * (size_t)(vIndex + 1)) - v0;
					V3 edge02 = ld3(verts + 4 * (size_t)(vIndex + 2)) - v0;
					V3 pVec = cross(d, edge02);
					float det = dot(edge01, pVec);
					if (pm_fabs(det) < INTERSECTION_EPSILON) continue;
					float invDet = pm_rcp(det);
					V3 tVec = o - v0;
					float u = dot(tVec, pVec) * invDet;
					if (u < 0.0f || u > 1.0f) continue;
					V3 qVec = cross(tVec, edge01);
					float v = dot(d, qVec) * invDet;
					if (v < 0.0f || u + v > 1.0f) continue;
					float t = dot(edge02, qVec) * invDet;
					if (ANY_HIT) {
						if (t > INTERSECTION_EPSILON && t < maxDist) { gotHit = 1; stackIndex = -1; break; }
					} else if (t > INTERSECTION_EPSILON && t < is.wuvt[3]) {
						is.wuvt[0] = 1.0f - (u + v); is.wuvt[1] = u; is.wuvt[2] = v; is.wuvt[3] = t;
						is.triIndex = (uint32_t)(vIndex / 3);
						is.meshInstance = meshInstanceId;
					}
				}
			}
			wantLeft = wantRight = 0;
		} else {
			child[0] = bvh[cur.ldata];
			child[1] = bvh[cur.rdata];
			V3 invDir = {pm_rcp(d.x), pm_rcp(d.y), pm_rcp(d.z)};
			float lHit = slab(child[0], o, invDir, maxDist);
			float rHit = slab(child[1], o, invDir, maxDist);
			wantLeft = lHit < FLT_MAX_ ? 1 : 0;
			wantRight = rHit < FLT_MAX_ ? 1 : 0;
		}
		if (ANY_HIT && stackIndex < 0) break;
		if (wantLeft && wantRight) { // always left first (intersect.cl:324-326)
			if (stackIndex >= BVH_MAX_STACK_SIZE) { *overflow = true; break; }
			nodeStack[stackIndex++] = (uint32_t)cur.rdata;
			cur = child[0];
		} else if (wantLeft || wantRight) {
			cur = wantLeft ? child[0] : child[1];
		} else {
			if (stackIndex == meshStart) { o = origO; d = origD; meshStart = -1; }
			if (--stackIndex >= 0) cur = bvh[nodeStack[stackIndex]];
		}
	}
	if (ANY_HIT) return gotHit;
	*isOut = is;
	return is.wuvt[3] < maxDist ? 1 : 0;
}

// ---------------------------------------------------------------------------------------
// kernels/pt_integrator.cl:17-211 shadeHits, one work-item.  Emits at most one occlusion ray
// (+ sample) and one indirect ray into the per-item slots; the caller compacts them in id
// order, which equals the reference's atomic compaction with work-group size 1.
// ---------------------------------------------------------------------------------------
struct ShadeOut {
	bool occl, indirect, emitterHit;
	Ray occlRay, indirectRay;
	float sample[4];
};

inline void shadeHit(int gid, const Ray *rays, Path *paths, const Intersection *isects, const SceneRefs &R, uint32_t numEmissives,
                     uint32_t bounce, uint32_t minBouncesForRR, uint32_t seed, float *accumulator, bool fixEmitterIndex, ShadeOut *o) {
	o->occl = o->indirect = o->emitterHit = false;
	float bxdfPdf = 1.0f, bxdfWeight = 1.0f;
	V3 bxdfTint = v3s(1.0f);
	// reference leaves these uninitialised when numEmissives == 0 (pt_integrator.cl:63-64,158-159);
	// zero is the value the host build of the reference reads too (quirk a-9(7))
	V3 emissiveOutRayDir = zero3(), emissiveSample = zero3();
	float emissivePdf = 0.0f, emissiveSelectionPdf = 0.0f, emissiveWeight = 0.0f, distToEmissive = 0.0f;

	uint32_t rnd[2] = {seed, (uint32_t)gid};
	V2 sample0 = randomGetSample2f(rnd);
	V2 sample1 = randomGetSample2f(rnd);
	V2 sample2 = randomGetSample2f(rnd);

	const Ray &ray = rays[gid];
	uint32_t rayPathIndex = (uint32_t)ray.dir[3];
	V3 inRayDir = -ld3(ray.dir);
	Path &path = paths[rayPathIndex];
	V3 curPathThroughput = ld3(path.throughput);

	Surface surface;
	surfaceInit(&surface, &isects[gid], R);

	MatNode materialNode;
	matSelectNode(&path, &surface, inRayDir, &materialNode, &bxdfTint, R.nodes, rnd, R.tm, R.td);
	float inRayDotNormal = dot(inRayDir, surface.normal);

	if (materialNode.type == POLARIS_BXDF_EMISSIVE) {
		if (inRayDotNormal > 0.0f) { // pt_integrator.cl:103-107
			V3 add = curPathThroughput * materialNode.scale * matGetSample3f(surface.uv, K(materialNode), materialNode.tex, R.tm, R.td);
			size_t idx = fixEmitterIndex ? path.pixelIndex : rayPathIndex;
			accumulator[4 * idx + 0] += add.x;
			accumulator[4 * idx + 1] += add.y;
			accumulator[4 * idx + 2] += add.z;
			o->emitterHit = true;
		}
		return;
	}
	bool rejectSample = materialNode.type == POLARIS_BXDF_INVALID;
	if (bounce >= minBouncesForRR) { // :113-125
		float rrProbability = pm_max(pm_min(0.5f, 0.2126f * curPathThroughput.x + 0.7152f * curPathThroughput.y + 0.0722f * curPathThroughput.z), 0.01f);
		if (rrProbability < sample2.x) rejectSample = true;
		else curPathThroughput = curPathThroughput / rrProbability;
	}
	if (rejectSample) return;

	V3 bxdfOutRayDir = zero3();
	V3 bxdfSample = bxdfGetSample(&surface, &materialNode, R.tm, R.td, sample0, inRayDir, &bxdfOutRayDir, &bxdfPdf);
	float displaceDir = pm_sign(dot(surface.normal, bxdfOutRayDir));
	V3 outBxdfRayOrigin = surface.point + (surface.normal * displaceDir) * INTERSECTION_EPSILON;
	V3 outEmissiveRayOrigin = surface.point + surface.normal * INTERSECTION_EPSILON;

	int emissiveIndex = numEmissives > 0 ? emissiveSelect((int)numEmissives, sample1.x, &emissiveSelectionPdf) : -1;
	if (emissiveIndex > -1) { // :142-155
		const PolarisEmissive *em = R.sc->emissives + emissiveIndex;
		emissiveSample = emissiveGetSample(&surface, em, R, sample1, &emissiveOutRayDir, &emissivePdf, &distToEmissive);
		float bxdfEmissivePdf = bxdfGetPdf(&surface, &materialNode, R.tm, R.td, inRayDir, emissiveOutRayDir);
		emissiveWeight = (emissivePdf * emissivePdf) / (emissivePdf * emissivePdf + bxdfEmissivePdf * bxdfEmissivePdf);
		float emissiveBxdfPdf = emissiveGetPdf(&surface, em, R, bxdfOutRayDir);
		bxdfWeight = (bxdfPdf * bxdfPdf) / (bxdfPdf * bxdfPdf + emissiveBxdfPdf * emissiveBxdfPdf);
	}
	float nDotEmissiveOutRay = pm_max(0.0f, dot(surface.normal, emissiveOutRayDir));
	if (maxcomp(emissiveSample) > 0.0f && emissivePdf > 0.0f && nDotEmissiveOutRay > 0.0f) { // :158-163
		V3 bxdfEmissiveSample = bxdfEval(&surface, &materialNode, R.tm, R.td, inRayDir, emissiveOutRayDir);
		emissiveSample = emissiveSample * (emissiveWeight * bxdfEmissiveSample * curPathThroughput * nDotEmissiveOutRay / (emissivePdf * emissiveSelectionPdf));
		if (maxcomp(emissiveSample) > 0.0f) {
			o->occl = true;
			o->sample[0] = emissiveSample.x; o->sample[1] = emissiveSample.y; o->sample[2] = emissiveSample.z; o->sample[3] = 0.0f;
			Ray &r = o->occlRay; // :203
			r.origin[0] = outEmissiveRayOrigin.x; r.origin[1] = outEmissiveRayOrigin.y; r.origin[2] = outEmissiveRayOrigin.z;
			r.origin[3] = distToEmissive - INTERSECTION_WITH_LIGHT_EPSILON;
			r.dir[0] = emissiveOutRayDir.x; r.dir[1] = emissiveOutRayDir.y; r.dir[2] = emissiveOutRayDir.z; r.dir[3] = (float)rayPathIndex;
		}
	}
	if ((materialNode.type & (POLARIS_BXDF_CONDUCTOR | POLARIS_BXDF_DIELECTRIC)) != 0) bxdfWeight = 1.0f; // :166-168
	V3 throughput = bxdfWeight * bxdfSample * bxdfTint * pm_fabs(dot(surface.normal, bxdfOutRayDir));
	if (maxcomp(throughput) > 0.0f && bxdfPdf > 0.0f) { // :173-177
		V3 t = curPathThroughput * throughput / bxdfPdf;
		path.throughput[0] = t.x; path.throughput[1] = t.y; path.throughput[2] = t.z;
		o->indirect = true;
		Ray &r = o->indirectRay; // :209
		r.origin[0] = outBxdfRayOrigin.x; r.origin[1] = outBxdfRayOrigin.y; r.origin[2] = outBxdfRayOrigin.z; r.origin[3] = FLT_MAX_;
		r.dir[0] = bxdfOutRayDir.x; r.dir[1] = bxdfOutRayDir.y; r.dir[2] = bxdfOutRayDir.z; r.dir[3] = (float)rayPathIndex;
	}
}

// kernels/pt_integrator.cl:214-275
inline void shadeMiss(int gid, const Ray *rays, const Path *paths, const SceneRefs &R, uint32_t bg, bool primary, float *accumulator) {
	MatNode m = R.nodes[bg];
	const Ray &ray = rays[gid];
	uint32_t rayPathIndex = (uint32_t)ray.dir[3];
	V2 uv = rayToLatLongUV(ld3(ray.dir));
	V3 kd = matGetSample3f(uv, K(m), m.tex, R.tm, R.td);
	const Path &p = paths[rayPathIndex];
	V3 add = primary ? kd : ld3(p.throughput) * kd;
	float *a = accumulator + 4 * (size_t)p.pixelIndex;
	a[0] += add.x; a[1] += add.y; a[2] += add.z;
}

struct Timer {
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// One sample of Tracer.Trace: generate -> query -> bounce loop, into `accum` (frame-sized).
// All buffers are owned by the caller so the sample-parallel mode can give each thread its own.
struct SampleBuffers {
	std::vector<Ray> rays[3];
	std::vector<Path> paths;
	std::vector<int> hitFlags;
	std::vector<Intersection> isects;
	std::vector<float> emissiveSamples;
	std::vector<ShadeOut> souts;
	explicit SampleBuffers(int N) : paths(N), hitFlags(N), isects(N), emissiveSamples((size_t)N * 4), souts(N) {
		for (auto &r : rays) r.resize(N);
	}
};

struct TraceCtx {
	const PolarisSceneView *sc;
	const PolarisBlockRequest *req;
	const float *frustum;
	V3 eye;
	V2 texel;
	SceneRefs R;
	int N;
	bool fixEmitter, serial;
};

static void trace_sample(const TraceCtx &C, SampleBuffers &Bf, uint32_t s, const uint32_t *sseed, float *accum, PolarisTraceStats &st,
                         bool &overflow, const PolarisOracleTaps *taps) {
	const PolarisSceneView *sc = C.sc;
	const PolarisBlockRequest *req = C.req;
	const uint32_t W = req->frame_w, BH = req->block_h, BY = req->block_y, B = req->num_bounces;
	const int N = C.N;
	const bool serial = C.serial;
	const int bg = sc->scene_diffuse_mat_index;
	auto &rays = Bf.rays;
	auto &paths = Bf.paths;
	auto &hitFlags = Bf.hitFlags;
	auto &isects = Bf.isects;
	auto &emissiveSamples = Bf.emissiveSamples;
	auto &souts = Bf.souts;
	int counters[3] = {0, 0, 0};
	const bool tap = taps && taps->tap_sample == s;

	auto query = [&](uint32_t buf) {
		const int n = counters[buf];
		bool ovf = false;
#pragma omp parallel for schedule(dynamic, 64) reduction(|| : ovf) if (!serial)
		for (int g = 0; g < n; g++) {
			bool o = false;
			hitFlags[g] = traverse<false>(rays[buf][g], sc, &isects[g], &o);
			ovf = ovf || o;
		}
		overflow = overflow || ovf;
	};

#pragma omp parallel for schedule(static) if (!serial)
	for (int y = 0; y < (int)BH; y++)
		for (uint32_t x = 0; x < W; x++)
			generatePrimaryRay(x, (uint32_t)y, rays[0].data(), paths.data(), C.frustum, C.eye, C.texel, BY, W, sseed[0]);
	counters[0] = N; // camera.cl:27-29
	st.primary_rays += (uint64_t)N;
	if (tap && taps->primary_rays) memcpy(taps->primary_rays, rays[0].data(), (size_t)N * sizeof(Ray));

	uint32_t cur = 0;
	query(cur); // pipeline.go:107-111, CPU branch
	if (tap)
		for (int g = 0; g < N; g++) {
			if (taps->primary_hit) taps->primary_hit[g] = hitFlags[g];
			if (hitFlags[g]) {
				if (taps->primary_wuvt) memcpy(taps->primary_wuvt + 4 * g, isects[g].wuvt, 16);
				if (taps->primary_tri) {
					taps->primary_tri[2 * g] = (int32_t)isects[g].meshInstance;
					taps->primary_tri[2 * g + 1] = (int32_t)isects[g].triIndex;
				}
			}
		}

	for (uint32_t b = 0; b < B; b++) {
		const int n = counters[cur];
		st.rays_per_bounce[b] += (uint64_t)n;
		if (b > 0) st.indirect_rays += (uint64_t)n;
		// pipeline.go:134-143
		int nhit = 0;
		for (int g = 0; g < n; g++) nhit += hitFlags[g] ? 1 : 0;
		st.shaded_hits += (uint64_t)nhit;
		if (bg != -1) {
			st.shaded_misses += (uint64_t)(n - nhit);
#pragma omp parallel for schedule(static) if (!serial)
			for (int g = 0; g < n; g++)
				if (!hitFlags[g]) shadeMiss(g, rays[cur].data(), paths.data(), C.R, (uint32_t)bg, b == 0, accum);
		}
		// ShadeHits (resources.go:226-273)
#pragma omp parallel for schedule(dynamic, 64) if (!serial)
		for (int g = 0; g < n; g++) {
			souts[g].occl = souts[g].indirect = souts[g].emitterHit = false;
			if (hitFlags[g])
				shadeHit(g, rays[cur].data(), paths.data(), isects.data(), C.R, sc->num_emissives, b, req->min_bounces_for_rr, sseed[1 + b],
				         accum, C.fixEmitter, &souts[g]);
		}
		// stable compaction == atomic compaction with work-group size 1 (pt_integrator.cl:188-210)
		int nocc = 0, nind = 0;
		for (int g = 0; g < n; g++) {
			const ShadeOut &o = souts[g];
			if (o.emitterHit) st.emitter_hits++;
			if (o.occl) {
				rays[2][nocc] = o.occlRay;
				memcpy(&emissiveSamples[4 * (size_t)nocc], o.sample, 16);
				nocc++;
			}
			if (o.indirect) rays[1 - cur][nind++] = o.indirectRay;
		}
		counters[2] = nocc;
		counters[1 - cur] = nind;
		if (tap && b == 0 && taps->throughput0)
			for (int g = 0; g < N; g++) memcpy(taps->throughput0 + 4 * g, paths[g].throughput, 16);
		if (tap && taps->num_rays) {
			taps->num_rays[2 * b] = n;
			taps->num_rays[2 * b + 1] = nocc;
		}
		st.occl_per_bounce[b] += (uint64_t)nocc;
		st.occlusion_rays += (uint64_t)nocc;
		// RayIntersectionTest + AccumulateEmissiveSamples (pipeline.go:160-168, pt_integrator.cl:278-296)
		uint64_t unocc = 0;
		bool ovf = false;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : unocc) reduction(|| : ovf) if (!serial)
		for (int g = 0; g < nocc; g++) {
			bool o = false;
			int hit = traverse<true>(rays[2][g], sc, nullptr, &o);
			ovf = ovf || o;
			if (!hit) {
				unocc++;
				uint32_t pathIndex = (uint32_t)rays[2][g].dir[3];
				float *a = accum + 4 * (size_t)paths[pathIndex].pixelIndex;
				a[0] += emissiveSamples[4 * (size_t)g];
				a[1] += emissiveSamples[4 * (size_t)g + 1];
				a[2] += emissiveSamples[4 * (size_t)g + 2];
			}
		}
		overflow = overflow || ovf;
		st.unoccluded += unocc;
		if (b + 1 < B) { // pipeline.go:203-208
			cur = 1 - cur;
			query(cur);
		}
	}
}

static void add_stats(PolarisTraceStats &a, const PolarisTraceStats &b) {
	a.primary_rays += b.primary_rays; a.indirect_rays += b.indirect_rays; a.occlusion_rays += b.occlusion_rays;
	a.shaded_hits += b.shaded_hits; a.shaded_misses += b.shaded_misses; a.emitter_hits += b.emitter_hits; a.unoccluded += b.unoccluded;
	for (int i = 0; i < POLARIS_MAX_BOUNCES; i++) { a.rays_per_bounce[i] += b.rays_per_bounce[i]; a.occl_per_bounce[i] += b.occl_per_bounce[i]; }
}

} // namespace

// =========================================================================================
// C API (oracle_api.h)
// =========================================================================================
extern "C" int polaris_oracle_trace(const PolarisSceneView *sc, const float eye[3], const float frustum[16],
                                    const PolarisBlockRequest *req, const uint32_t *seeds, size_t n_seeds,
                                    float *trace_accum, PolarisTraceStats *stats, const PolarisOracleTaps *taps,
                                    uint32_t flags) {
	// tracer/opencl/tracer.go:194-247 (Trace) + pipeline.go:94-213 (MonteCarloIntegrator)
	const uint32_t W = req->frame_w, H = req->frame_h, BH = req->block_h, BY = req->block_y;
	const uint32_t B = req->num_bounces, spp = req->samples_per_pixel;
	if (!sc || !req || !trace_accum || W == 0 || BH == 0 || BY + BH > H || B > POLARIS_MAX_BOUNCES) return 1;
	if (n_seeds < (size_t)spp * (1 + B)) return 2;
	const int N = (int)(W * BH);
	const size_t F = (size_t)W * H;
	const bool sample_parallel = (flags & POLARIS_ORACLE_PARALLEL_SAMPLES) != 0 && !taps;
	Timer timer;

	TraceCtx C{sc, req, frustum, ld3(eye), {1.0f / (float)W, 1.0f / (float)H} /* resources.go:130-133 */,
	           SceneRefs{sc, sc->vertices, sc->normals, sc->uvs, sc->material_nodes, sc->texture_meta, sc->texture_data}, N,
	           (flags & POLARIS_ORACLE_FIX_EMITTER_INDEX) != 0, (flags & POLARIS_ORACLE_SERIAL) != 0 || sample_parallel};
	bool overflow = false;
	PolarisTraceStats st;
	memset(&st, 0, sizeof st);
	memset(trace_accum, 0, F * 4 * sizeof(float)); // ClearTraceAccumulator, tracer.go:215

	if (!sample_parallel) {
		// reference order: samples one after the other, every contribution added straight into the
		// accumulator (bit-exact with the compiled reference)
		SampleBuffers Bf(N);
		for (uint32_t s = 0; s < spp; s++) trace_sample(C, Bf, s, seeds + (size_t)s * (1 + B), trace_accum, st, overflow, taps);
	} else {
		// CPU-baseline mode: samples are independent, so threads take whole samples with private
		// buffers and a private accumulator strip; strips are added in thread order at the end
		// (same paths, per-pixel sums re-associated like the GPU's batched mode).
		const size_t strip0 = (size_t)BY * W * 4, strip_n = (size_t)BH * W * 4;
#pragma omp parallel
		{
			SampleBuffers Bf(N);
			std::vector<float> acc(F * 4, 0.0f);
			PolarisTraceStats lst;
			memset(&lst, 0, sizeof lst);
			bool lovf = false;
#pragma omp for schedule(dynamic, 1)
			for (int s = 0; s < (int)spp; s++) trace_sample(C, Bf, (uint32_t)s, seeds + (size_t)s * (1 + B), acc.data(), lst, lovf, nullptr);
#pragma omp critical
			{
				for (size_t i = 0; i < strip_n; i++) trace_accum[strip0 + i] += acc[strip0 + i];
				add_stats(st, lst);
				overflow = overflow || lovf;
			}
		}
	}
	st.device_ms = timer.ms();
	if (stats) *stats = st;
	return overflow ? 3 : 0; // 3: the reference's 32-entry traversal stack would have overflowed
}


extern "C" int polaris_oracle_tonemap(const float *accum, uint32_t n_pixels, float sample_weight, float exposure,
                                      uint8_t *rgba) { // kernels/hdr.cl:5-28
	for (uint32_t g = 0; g < n_pixels; g++) {
		V3 hdr = ld3(accum + 4 * (size_t)g) * sample_weight * exposure;
		V3 mapped = hdr / (hdr + 1.0f);
		const float e = 1.0f / 2.2f;
		V3 p = {pm_pow(mapped.x, e), pm_pow(mapped.y, e), pm_pow(mapped.z, e)};
		V3 out = {pm_clamp(p.x, 0.0f, 1.0f) * 255.0f, pm_clamp(p.y, 0.0f, 1.0f) * 255.0f, pm_clamp(p.z, 0.0f, 1.0f) * 255.0f};
		rgba[4 * (size_t)g + 0] = (uint8_t)out.x; // truncation, hdr.cl:22-27
		rgba[4 * (size_t)g + 1] = (uint8_t)out.y;
		rgba[4 * (size_t)g + 2] = (uint8_t)out.z;
		rgba[4 * (size_t)g + 3] = 255;
	}
	return 0;
}

extern "C" void polaris_oracle_random(uint32_t state[2], float out[2]) {
	V2 r = randomGetSample2f(state);
	out[0] = r.x;
	out[1] = r.y;
}

extern "C" int polaris_oracle_intersect_probe(const PolarisSceneView *scene, const float *rays, uint32_t n, int any_hit,
                                              int32_t *hit, float *wuvt, int32_t *inst_tri) {
	bool overflow = false;
#pragma omp parallel for schedule(dynamic, 256) reduction(|| : overflow)
	for (long i = 0; i < (long)n; i++) {
		Ray r;
		memcpy(&r, rays + 8 * (size_t)i, sizeof r);
		Intersection is;
		memset(&is, 0, sizeof is);
		bool ov = false;
		const int h = any_hit ? traverse<true>(r, scene, nullptr, &ov) : traverse<false>(r, scene, &is, &ov);
		overflow = overflow || ov;
		hit[i] = h;
		if (!any_hit && h) {
			if (wuvt) memcpy(wuvt + 4 * (size_t)i, is.wuvt, 16);
			if (inst_tri) { inst_tri[2 * (size_t)i] = (int32_t)is.meshInstance; inst_tri[2 * (size_t)i + 1] = (int32_t)is.triIndex; }
		}
	}
	return overflow ? 1 : 0;
}

extern "C" void polaris_oracle_bxdf_probe(const PolarisMaterialNode *node, const PolarisTextureMetadata *tex_meta,
                                          const uint8_t *tex_data, const float normal[3], const float uv[2],
                                          const float in_dir[3], const float sample[2], const float eval_dir[3],
                                          float out[11]) {
	Surface sf{zero3(), ld3(normal), ld2(uv), 0};
	MatNode m = *node;
	V3 od = zero3();
	float pdf = 0.0f;
	V3 v = bxdfGetSample(&sf, &m, tex_meta, tex_data, ld2(sample), ld3(in_dir), &od, &pdf);
	out[0] = v.x; out[1] = v.y; out[2] = v.z;
	out[3] = od.x; out[4] = od.y; out[5] = od.z;
	out[6] = pdf;
	out[7] = bxdfGetPdf(&sf, &m, tex_meta, tex_data, ld3(in_dir), ld3(eval_dir));
	V3 e = bxdfEval(&sf, &m, tex_meta, tex_data, ld3(in_dir), ld3(eval_dir));
	out[8] = e.x; out[9] = e.y; out[10] = e.z;
}

extern "C" void polaris_oracle_tex_probe(const PolarisTextureMetadata *tex_meta, const uint8_t *tex_data,
                                         int32_t tex_index, const float uv[2], float out[7]) {
	V3 a = texGetSample3f(ld2(uv), tex_index, tex_meta, tex_data);
	out[0] = a.x; out[1] = a.y; out[2] = a.z;
	out[3] = texGetSample1f(ld2(uv), tex_index, tex_meta, tex_data);
	V3 b = texGetBumpSample3f(ld2(uv), tex_index, tex_meta, tex_data);
	out[4] = b.x; out[5] = b.y; out[6] = b.z;
}

extern "C" void polaris_oracle_emissive_probe(const PolarisSceneView *sc, uint32_t emissive_index, const float point[3],
                                              const float normal[3], const float sample[2], const float pdf_dir[3],
                                              float out[9]) {
	SceneRefs R{sc, sc->vertices, sc->normals, sc->uvs, sc->material_nodes, sc->texture_meta, sc->texture_data};
	Surface sf{ld3(point), ld3(normal), {0.0f, 0.0f}, 0};
	const PolarisEmissive *em = sc->emissives + emissive_index;
	V3 od = zero3();
	float pdf = 0.0f, dist = 0.0f;
	V3 v = emissiveGetSample(&sf, em, R, ld2(sample), &od, &pdf, &dist);
	out[0] = v.x; out[1] = v.y; out[2] = v.z;
	out[3] = od.x; out[4] = od.y; out[5] = od.z;
	out[6] = pdf;
	out[7] = dist;
	out[8] = emissiveGetPdf(&sf, em, R, ld3(pdf_dir));
}

extern "C" void polaris_oracle_material_probe(const PolarisSceneView *sc, uint32_t root, const float normal[3], const float uv[2],
                                              const uint32_t rng_state[2], uint32_t path_flags, float out[18]) {
	Path path;
	memset(&path, 0, sizeof path);
	path.flags = path_flags;
	Surface sf{zero3(), ld3(normal), ld2(uv), root};
	MatNode m;
	V3 tint = v3(1.0f, 1.0f, 1.0f);
	uint32_t rnd[2] = {rng_state[0], rng_state[1]};
	matSelectNode(&path, &sf, zero3(), &m, &tint, sc->material_nodes, rnd, sc->texture_meta, sc->texture_data);
	memcpy(&out[0], &m.type, 4);
	out[1] = m.int_ior; out[2] = m.ext_ior;
	out[3] = sf.normal.x; out[4] = sf.normal.y; out[5] = sf.normal.z;
	out[6] = tint.x; out[7] = tint.y; out[8] = tint.z;
	memcpy(&out[9], &path.flags, 4);
	memcpy(&out[10], &rnd[0], 4); memcpy(&out[11], &rnd[1], 4);
	for (int k = 0; k < 3; k++) { out[12 + k] = m.k[k]; out[15 + k] = m.t[k]; }
}

extern "C" const char *polaris_oracle_describe(void) {
	return "CPU restatement of tracer/opencl (oracle/polaris_oracle.cpp); built-ins: polaris_math.h; OpenMP";
}
