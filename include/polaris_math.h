/*
 * polaris_math.h -- deterministic IEEE-754 binary32 math for the polaris tracer path.
 *
 * Why this exists
 * ---------------
 * The reference device code (tracer/opencl/CL) calls OpenCL
 * built-ins whose results are implementation defined: native_cos / native_sin /
 * native_sqrt / native_recip (kernels/camera.cl:41-42, samplers/distribution_sampler.cl:66-69,
 * :111, samplers/emissive_sampler.cl:67,99,147,234, kernels/intersect.cl:104,136), atan / atan2 /
 * acos (util/transform.cl:29-36, distribution_sampler.cl:63), pow (kernels/hdr.cl:22) and the
 * geometric built-ins dot / cross / normalize / length / mix / clamp / sign / fmin / fmax.
 * A path tracer's branch decisions (Russian roulette, Fresnel choice, hit / miss at an edge)
 * depend on the last bit of those results, so "the reference result" is only defined once
 * one concrete implementation of the built-ins is fixed.
 *
 * This header fixes one: every function below is written with +, -, *, /, sqrt, floor and
 * integer bit manipulation only, evaluated in the written order, and must be compiled with
 * floating-point contraction OFF (-ffp-contract=off) and without fast-math on every
 * target.  IEEE-754 then guarantees bit-identical results on x86-64 and on gfx950, which
 * is what lets (a) the reference's own OpenCL C, compiled for the host with these
 * functions bound to its built-ins (oracle/refbuild), (b) the CPU restatement (oracle/)
 * and (c) the HIP kernels (polaris_amd/csrc) agree bit for bit.
 *
 * The polynomial kernels are the classic single-precision minimax approximations
 * published in Moshier's Cephes library (sinf/cosf/atanf/asinf/logf/expf); they are
 * restated here, not copied from any file of the reference (which contains no such code).
 * Accuracy is <= 2 ulp on the ranges the tracer uses, i.e. inside what OpenCL allows for
 * the full-precision built-ins and far inside what it allows for native_*.
 *
 * This is a public header of the product (the HIP kernels include it); the test oracle
 * includes it too so that both sides share ONE definition of the built-ins.
 */
#ifndef POLARIS_MATH_H
#define POLARIS_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PM_HD __host__ __device__ __forceinline__
#else
#define PM_HD static inline
#endif

#define PM_FLT_MAX 3.402823466e+38f

PM_HD uint32_t pm_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PM_HD float pm_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- exact / correctly rounded primitives ------------------------------------------- */

PM_HD float pm_fabs(float x) { return pm_u2f(pm_f2u(x) & 0x7fffffffu); }
PM_HD float pm_sqrt(float x) { return __builtin_sqrtf(x); }      /* IEEE correctly rounded */
PM_HD float pm_rcp(float x) { return 1.0f / x; }                 /* IEEE correctly rounded */
PM_HD float pm_floor(float x) { return __builtin_floorf(x); }    /* exact */

/* OpenCL min/max(float,float): "y < x ? y : x" / "x < y ? y : x" (OpenCL 1.2 s6.12.4). */
PM_HD float pm_min(float x, float y) { return y < x ? y : x; }
PM_HD float pm_max(float x, float y) { return x < y ? y : x; }
/* OpenCL fmin/fmax (s6.12.2): "returns y if y < x, otherwise x; if one argument is a NaN, returns the other" --
 * written so that the zero of fmin(+0, -0) / fmax(-0, +0) is the one that text selects (x). */
PM_HD float pm_fmin(float x, float y) { return (y < x || x != x) ? y : x; }
PM_HD float pm_fmax(float x, float y) { return (x < y || x != x) ? y : x; }
/* clamp(x,lo,hi) = min(max(x,lo),hi) (s6.12.4). */
PM_HD float pm_clamp(float x, float lo, float hi) { return pm_min(pm_max(x, lo), hi); }
PM_HD int32_t pm_clampi(int32_t x, int32_t lo, int32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }
PM_HD uint32_t pm_clampu(uint32_t x, uint32_t lo, uint32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* sign(x): 1, -1, +-0 -> +-0, NaN -> 0 (s6.12.4). */
PM_HD float pm_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == x ? x : 0.0f)); }
/* mix(a,b,t) = a + (b - a) * t (s6.12.4). */
PM_HD float pm_mix(float a, float b, float t) { return a + (b - a) * t; }

/* ---- sin / cos ---------------------------------------------------------------------- */

/* Shared range reduction: j = round-to-even-octant index, r = x - j*pi/4 (3-part Cody-Waite).
 * Valid for |x| < 8192. */
PM_HD float pm__reduce_pio4(float ax, uint32_t *octant) {
	uint32_t j = (uint32_t)(ax * 1.27323954473516f); /* 4/pi */
	j = (j + 1u) & ~1u;                               /* map to even octant */
	float y = (float)j;
	float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
	*octant = j;
	return r;
}

PM_HD float pm__sin_poly(float r) {
	float z = r * r;
	return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
}

PM_HD float pm__cos_poly(float r) {
	float z = r * r;
	return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}

PM_HD float pm_sin(float x) {
	uint32_t j;
	float ax = pm_fabs(x);
	float r = pm__reduce_pio4(ax, &j);
	uint32_t q = (j >> 1) & 3u;                       /* quadrant */
	float v = (q & 1u) ? pm__cos_poly(r) : pm__sin_poly(r);
	if (q & 2u) v = -v;
	return x < 0.0f ? -v : v;
}

PM_HD float pm_cos(float x) {
	uint32_t j;
	float ax = pm_fabs(x);
	float r = pm__reduce_pio4(ax, &j);
	uint32_t q = (j >> 1) & 3u;
	float v = (q & 1u) ? pm__sin_poly(r) : pm__cos_poly(r);
	if (q == 1u || q == 2u) v = -v;
	return v;
}

/* ---- atan / atan2 / acos ------------------------------------------------------------ */

#define PM_PI     3.14159265358979323846f
#define PM_PI_2   1.57079632679489661923f
#define PM_PI_4   0.78539816339744830962f

PM_HD float pm_atan(float xx) {
	float x = pm_fabs(xx);
	float y;
	if (x > 2.414213562373095f) {          /* tan(3pi/8) */
		y = PM_PI_2;
		x = -(1.0f / x);
	} else if (x > 0.4142135623730950f) {  /* tan(pi/8) */
		y = PM_PI_4;
		x = (x - 1.0f) / (x + 1.0f);
	} else {
		y = 0.0f;
	}
	float z = x * x;
	y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
	return xx < 0.0f ? -y : y;
}

PM_HD float pm_atan2(float y, float x) {
	if (x == 0.0f) {
		if (y > 0.0f) return PM_PI_2;
		if (y < 0.0f) return -PM_PI_2;
		return 0.0f;
	}
	float w = 0.0f;
	if (x < 0.0f) w = (pm_f2u(y) >> 31) ? -PM_PI : PM_PI; /* sign BIT of y: atan2(-0, x < 0) = -pi (OpenCL 1.2 s7.5.1) */
	return w + pm_atan(y / x);
}

PM_HD float pm__asin_core(float a /* |x| <= 1 */) {
	/* returns asin(a) for a in [0,1] */
	float z, x;
	int big = a > 0.5f;
	if (big) {
		z = 0.5f * (1.0f - a);
		x = pm_sqrt(z);
	} else {
		x = a;
		z = x * x;
	}
	float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
	if (big) {
		p = p + p;
		p = PM_PI_2 - p;
	}
	return p;
}

PM_HD float pm_acos(float x) {
	if (x != x) return x;
	if (x >= 1.0f) return 0.0f;
	if (x <= -1.0f) return PM_PI;
	if (x < -0.5f) return PM_PI - 2.0f * pm__asin_core(pm_sqrt(0.5f * (1.0f + x)));
	if (x > 0.5f) return 2.0f * pm__asin_core(pm_sqrt(0.5f * (1.0f - x)));
	float a = pm_fabs(x);
	float s = pm__asin_core(a);
	return PM_PI_2 - (x < 0.0f ? -s : s);
}

/* ---- log / exp / pow ---------------------------------------------------------------- */

PM_HD float pm_log(float x) {
	/* x > 0, finite */
	int32_t e = 0;
	if (x < 1.17549435e-38f) { x *= 16777216.0f; e = -24; }
	uint32_t u = pm_f2u(x);
	e += (int32_t)((u >> 23) & 0xffu) - 126;
	float m = pm_u2f((u & 0x007fffffu) | 0x3f000000u);   /* m in [0.5, 1) */
	if (m < 0.707106781186547524f) {
		e -= 1;
		m = m + m - 1.0f;
	} else {
		m = m - 1.0f;
	}
	float z = m * m;
	float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
	               + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
	           + 3.3333331174e-1f) * m * z;
	float fe = (float)e;
	y += -2.12194440e-4f * fe;
	y += -0.5f * z;
	z = m + y;
	z += 0.693359375f * fe;
	return z;
}

PM_HD float pm_exp(float x) {
	if (x > 88.72283905206835f) return PM_FLT_MAX;
	if (x < -87.0f) return 0.0f;
	float fn = pm_floor(1.44269504088896341f * x + 0.5f);
	int32_t n = (int32_t)fn;
	x -= fn * 0.693359375f;
	x -= fn * -2.12194440e-4f;
	float z = x * x;
	z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x
	      + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
	/* ldexp(z, n), n in [-126, 128] here */
	if (n > 127) { z *= 2.0f; n -= 1; }
	return z * pm_u2f((uint32_t)(n + 127) << 23);
}

/* pow for the tone-mapper (kernels/hdr.cl:22): base >= 0. */
PM_HD float pm_pow(float x, float y) {
	if (!(x > 0.0f)) return (x == 0.0f) ? 0.0f : (x - x) / (x - x);
	return pm_exp(y * pm_log(x));
}

#endif /* POLARIS_MATH_H */
