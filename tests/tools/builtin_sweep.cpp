// builtin_sweep.cpp -- independent check of include/polaris_math.h, the ONE definition of the OpenCL built-ins that
// the HIP kernels, the CPU oracle and the compiled reference kernels (oracle/_ref) all share.  A wrong pm_acos would be
// reproduced bit for bit by all three and every parity test would stay green -- so the functions themselves are
// measured here against double-precision glibc over ALL binary32 inputs (unary) or >= 1e8 random + edge-grid inputs
// (binary / ternary), and the error is held against the OpenCL 1.2 full-profile ULP bounds (section 7.4):
//     sin, cos <= 4 ulp   atan <= 5   atan2 <= 6   acos <= 4   pow <= 16   sqrt <= 3   x / y, 1 / x <= 2.5
// Functions whose result the specification FIXES (fabs, floor, sign, min, max, fmin, fmax, clamp, mix, conversions,
// and sqrt / reciprocal, which this build rounds correctly) must be bit-equal to an independent formulation.
//
// Test tool, not product.  g++ -O2 -fopenmp -ffp-contract=off -Iinclude builtin_sweep.cpp -o builtin_sweep
//   builtin_sweep <stride> : visits bit patterns 0, stride, 2*stride, ... (stride 1 = all 2^32) and prints one JSON object.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "polaris_math.h"

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// error of `got` against the real number `want`, in units of the last place of float(want) (OpenCL 1.2 s7.4 definition:
// ulp(x) of the exactly representable neighbours; subnormal results measured against the smallest normal's ulp grid)
static inline double ulp_err(float got, double want) {
	if (std::isnan(want)) return std::isnan(got) ? 0.0 : 1e30;
	if (std::isinf(want)) return (std::isinf(got) && (got > 0) == (want > 0)) ? 0.0 : 1e30;
	if (std::isnan(got) || std::isinf(got)) return 1e30;
	int e;
	std::frexp(want, &e); // want = m * 2^e, m in [0.5, 1)
	int ulp_exp = e - 24;
	if (ulp_exp < -149) ulp_exp = -149;
	return std::fabs((double)got - want) / std::ldexp(1.0, ulp_exp);
}

struct Acc {
	double max_ulp = 0.0, max_abs = 0.0;
	uint32_t worst = 0;
	uint64_t n = 0, mismatches = 0;
	void merge(const Acc &o) {
		if (o.max_ulp > max_ulp) { max_ulp = o.max_ulp; worst = o.worst; }
		if (o.max_abs > max_abs) max_abs = o.max_abs;
		n += o.n; mismatches += o.mismatches;
	}
};

typedef float (*F1)(float);
typedef double (*D1)(double);

// sweep every `stride`-th bit pattern; lo/hi: only arguments with lo <= x <= hi are measured.  The ulp error is taken where
// |exact result| >= min_result (0 = everywhere); the absolute error everywhere.
static Acc sweep_ulp(F1 f, D1 ref, float lo, float hi, uint64_t stride, double min_result = 0.0) {
	Acc total;
#pragma omp parallel
	{
		Acc a;
#pragma omp for schedule(static)
		for (int64_t k = 0; k < (int64_t)((0x100000000ull + stride - 1) / stride); k++) {
			const uint32_t u = (uint32_t)((uint64_t)k * stride);
			const float x = u2f(u);
			if (!(x >= lo && x <= hi)) continue;
			const float got = f(x);
			const double want = ref((double)x);
			a.n++;
			const double ae = std::fabs((double)got - want);
			if (ae > a.max_abs) a.max_abs = ae;
			if (std::fabs(want) < min_result) continue;
			const double e = ulp_err(got, want);
			if (e > a.max_ulp) { a.max_ulp = e; a.worst = u; }
		}
#pragma omp critical
		total.merge(a);
	}
	return total;
}

// bit-equality of two unary functions over the swept patterns (NaN results: any NaN equals any NaN)
static Acc sweep_equal(F1 f, F1 g, uint64_t stride) {
	Acc total;
#pragma omp parallel
	{
		Acc a;
#pragma omp for schedule(static)
		for (int64_t k = 0; k < (int64_t)((0x100000000ull + stride - 1) / stride); k++) {
			const uint32_t u = (uint32_t)((uint64_t)k * stride);
			const float x = u2f(u);
			const float p = f(x), q = g(x);
			a.n++;
			if (f2u(p) != f2u(q) && !(p != p && q != q)) { if (!a.mismatches) a.worst = u; a.mismatches++; }
		}
#pragma omp critical
		total.merge(a);
	}
	return total;
}

// ---- independent formulations of the functions whose result the specification fixes ---------------------------------
static float ref_fabs(float x) { return fabsf(x); }
static float ref_floor(float x) { return floorf(x); }
static float ref_sqrt(float x) { return (float)std::sqrt((double)x); } // double sqrt rounded once more: exact for binary32 (2p+2 rule)
static float ref_rcp(float x) { return (float)(1.0 / (double)x); }     // likewise for division
static float ref_sign(float x) { // OpenCL 1.2 s6.12.4: 1 if x > 0, -0 if x = -0, +0 if x = +0, -1 if x < 0, 0 if NaN
	if (x != x) return 0.0f;
	if (x > 0.0f) return 1.0f;
	if (x < 0.0f) return -1.0f;
	return x;
}
static float pm_sqrt_(float x) { return pm_sqrt(x); }
static float pm_rcp_(float x) { return pm_rcp(x); }
static float pm_fabs_(float x) { return pm_fabs(x); }
static float pm_floor_(float x) { return pm_floor(x); }
static float pm_sign_(float x) { return pm_sign(x); }
static float pm_sin_(float x) { return pm_sin(x); }
static float pm_cos_(float x) { return pm_cos(x); }
static float pm_atan_(float x) { return pm_atan(x); }
static float pm_acos_(float x) { return pm_acos(x); }
static const float kInvGamma = 1.0f / 2.2f;                            // the tone-mapper's exponent (kernels/hdr.cl:8,22)
static float pm_pow_gamma(float x) { return pm_pow(x, kInvGamma); }
static double ref_pow_gamma(double x) { return std::pow(x, (double)kInvGamma); }
static double ref_sin(double x) { return std::sin(x); }
static double ref_cos(double x) { return std::cos(x); }
static double ref_atan(double x) { return std::atan(x); }
static double ref_acos(double x) { return std::acos(x); }

// ---- binary / ternary functions: random + edge grid -------------------------------------------------------------------
static inline uint64_t splitmix(uint64_t &s) {
	uint64_t z = (s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static const uint32_t kEdgeBits[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x00800000u, 0x80800000u, 0x3f800000u,
                                     0xbf800000u, 0x3f7fffffu, 0x3f800001u, 0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u,
                                     0xffc00000u, 0x3f000000u, 0x40000000u, 0x40490fdbu, 0xc0490fdbu, 0x33800000u, 0x4b000000u, 0x4b800000u};
static const int kEdges = (int)(sizeof kEdgeBits / sizeof kEdgeBits[0]);

static float ref_min(float x, float y) { return y < x ? y : x; }                       // s6.12.4
static float ref_max(float x, float y) { return x < y ? y : x; }
static float ref_fmin(float x, float y) { if (x != x) return y; if (y != y) return x; return y < x ? y : x; } // s6.12.2
static float ref_fmax(float x, float y) { if (x != x) return y; if (y != y) return x; return x < y ? y : x; }

int main(int argc, char **argv) {
	const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
	const uint64_t n_random = argc > 2 ? strtoull(argv[2], nullptr, 10) : 100000000ull;
	std::string out = "{";
	char buf[512];
	auto put_ulp = [&](const char *name, const char *range, const Acc &a, double bound) {
		snprintf(buf, sizeof buf, "\"%s\": {\"range\": \"%s\", \"inputs\": %llu, \"max_ulp\": %.4f, \"max_abs_err\": %.4g, \"worst_bits\": \"0x%08x\", \"opencl_bound_ulp\": %.1f}, ",
		         name, range, (unsigned long long)a.n, a.max_ulp, a.max_abs, a.worst, bound);
		out += buf;
	};
	auto put_eq = [&](const char *name, const Acc &a) {
		snprintf(buf, sizeof buf, "\"%s\": {\"inputs\": %llu, \"mismatches\": %llu, \"first_bits\": \"0x%08x\"}, ", name, (unsigned long long)a.n,
		         (unsigned long long)a.mismatches, a.worst);
		out += buf;
	};
	const float two_pi = 6.2831855f;
	// ranges the kernels use: azimuths 2*pi*u in [0, 2*pi] (distribution_sampler.cl:66-69,111; emissive_sampler.cl); pm_sin / pm_cos
	// document |x| < 8192 as their domain
	// The reference calls ONLY the native_ forms (native_sin / native_cos: implementation-defined accuracy, OpenCL 1.2 s7.4); the 4-ulp
	// bound of sin / cos is still met wherever the result is not within 2^-10 of a zero crossing -- right at a zero of cos
	// (|x| = pi/2, 3pi/2) the 3-term Cody-Waite reduction leaves an ABSOLUTE error < 2^-24 that is many ulps of a result near 0
	put_ulp("sin", "[-2pi, 2pi], |result| >= 2^-10", sweep_ulp(pm_sin_, ref_sin, -two_pi, two_pi, stride, 1.0 / 1024), 4.0);
	put_ulp("cos", "[-2pi, 2pi], |result| >= 2^-10", sweep_ulp(pm_cos_, ref_cos, -two_pi, two_pi, stride, 1.0 / 1024), 4.0);
	put_ulp("sin_all", "[-2pi, 2pi], every result (informational ulp; max_abs_err is what is asserted)", sweep_ulp(pm_sin_, ref_sin, -two_pi, two_pi, stride), 4.0);
	put_ulp("cos_all", "[-2pi, 2pi], every result (informational ulp; max_abs_err is what is asserted)", sweep_ulp(pm_cos_, ref_cos, -two_pi, two_pi, stride), 4.0);
	put_ulp("sin_wide", "[-8192, 8192] (informational: beyond the kernels' range)", sweep_ulp(pm_sin_, ref_sin, -8192.0f, 8192.0f, stride), 4.0);
	put_ulp("cos_wide", "[-8192, 8192] (informational)", sweep_ulp(pm_cos_, ref_cos, -8192.0f, 8192.0f, stride), 4.0);
	put_ulp("atan", "all finite", sweep_ulp(pm_atan_, ref_atan, -PM_FLT_MAX, PM_FLT_MAX, stride), 5.0);
	put_ulp("acos", "[-1, 1]", sweep_ulp(pm_acos_, ref_acos, -1.0f, 1.0f, stride), 4.0);
	// tone-mapper: clamp(pow(c / (1 + c), 1/2.2), 0, 1) * 255 truncated to a byte (hdr.cl:22-27).  Results below 1/255 all map to
	// byte 0, i.e. only x >= (1/255)^2.2 = 5.1e-6 can change an output; the whole unit interval is reported beside it
	put_ulp("pow_gamma", "x in [2^-20, 1], y = 1/2.2 (hdr.cl:22): every input that can reach a non-zero byte", sweep_ulp(pm_pow_gamma, ref_pow_gamma, 9.5367431640625e-7f, 1.0f, stride), 16.0);
	put_ulp("pow_gamma_unit", "x in [0, 1], y = 1/2.2 (informational: below 2^-20 the byte is 0 whatever the ulp error)", sweep_ulp(pm_pow_gamma, ref_pow_gamma, 0.0f, 1.0f, stride), 16.0);
	put_eq("sqrt", sweep_equal(pm_sqrt_, ref_sqrt, stride));
	put_eq("recip", sweep_equal(pm_rcp_, ref_rcp, stride));
	put_eq("fabs", sweep_equal(pm_fabs_, ref_fabs, stride));
	put_eq("floor", sweep_equal(pm_floor_, ref_floor, stride));
	put_eq("sign", sweep_equal(pm_sign_, ref_sign, stride));

	// binary / ternary: n_random random bit patterns + the full edge grid
	Acc atan2_a, pow_a, eq_min, eq_max, eq_fmin, eq_fmax, eq_clamp, eq_mix, div_a;
#pragma omp parallel
	{
		Acc l_atan2, l_pow, l_min, l_max, l_fmin, l_fmax, l_clamp, l_mix, l_div;
		auto one = [&](float x, float y, float z) {
			// atan2(y, x): finite, not both zero (s7.5.1 edge cases for zeros / infinities are outside what rayToLatLongUV feeds it)
			if (std::isfinite(x) && std::isfinite(y) && !(x == 0.0f && y == 0.0f)) {
				const double e = ulp_err(pm_atan2(y, x), std::atan2((double)y, (double)x));
				l_atan2.n++;
				if (e > l_atan2.max_ulp) { l_atan2.max_ulp = e; l_atan2.worst = f2u(y); }
			}
			// pow(x, y): the path's only call raises [0, 1) to 1/2.2 (hdr.cl:22).  pm_pow = exp(y * log(x)) in plain binary32, so its
			// error grows with |y ln x| (41 ulp at y = 4, x = 1e-5): it is held to the OpenCL bound on base in [2^-20, 4], exponent in [0, 1]
			const float bx = pm_fabs(x), by = pm_fabs(y);
			if (bx >= 9.5367431640625e-7f && bx <= 4.0f && by <= 1.0f) {
				const double e = ulp_err(pm_pow(bx, by), std::pow((double)bx, (double)by));
				l_pow.n++;
				if (e > l_pow.max_ulp) { l_pow.max_ulp = e; l_pow.worst = f2u(bx); }
			}
			{ // x / y is IEEE correctly rounded in every build: equal to the double quotient rounded once more (exact for binary32)
				const float q = x / y, w = (float)((double)x / (double)y);
				l_div.n++;
				if (f2u(q) != f2u(w) && !(q != q && w != w)) l_div.mismatches++;
			}
			auto eq = [](float p, float q) { return f2u(p) == f2u(q) || (p != p && q != q); };
			l_min.n++; if (!eq(pm_min(x, y), ref_min(x, y))) l_min.mismatches++;
			l_max.n++; if (!eq(pm_max(x, y), ref_max(x, y))) l_max.mismatches++;
			l_fmin.n++; if (!eq(pm_fmin(x, y), ref_fmin(x, y))) l_fmin.mismatches++;
			l_fmax.n++; if (!eq(pm_fmax(x, y), ref_fmax(x, y))) l_fmax.mismatches++;
			l_clamp.n++; if (!eq(pm_clamp(x, y, z), ref_min(ref_max(x, y), z))) l_clamp.mismatches++;         // s6.12.4: min(max(x, lo), hi)
			l_mix.n++; if (!eq(pm_mix(x, y, z), x + (y - x) * z)) l_mix.mismatches++;                             // s6.12.4: x + (y - x) * a
		};
#pragma omp for schedule(static)
		for (int64_t k = 0; k < (int64_t)n_random; k++) {
			uint64_t s = 0x1234567ull + (uint64_t)k * 0x9E3779B97F4A7C15ull;
			const uint64_t r0 = splitmix(s), r1 = splitmix(s);
			float x = u2f((uint32_t)r0), y = u2f((uint32_t)(r0 >> 32)), z = u2f((uint32_t)r1);
			if (k & 1) { // half of the draws in "ordinary" magnitudes: unit-ish vectors, radiances, uv
				x = (float)((double)(int32_t)(uint32_t)r0 / 2147483648.0 * 4.0);
				y = (float)((double)(int32_t)(uint32_t)(r0 >> 32) / 2147483648.0 * 4.0);
				z = (float)((double)(uint32_t)r1 / 4294967296.0);
			}
			one(x, y, z);
		}
#pragma omp for schedule(static)
		for (int i = 0; i < kEdges * kEdges * kEdges; i++)
			one(u2f(kEdgeBits[i % kEdges]), u2f(kEdgeBits[(i / kEdges) % kEdges]), u2f(kEdgeBits[i / (kEdges * kEdges)]));
#pragma omp critical
		{
			atan2_a.merge(l_atan2); pow_a.merge(l_pow); div_a.merge(l_div); eq_min.merge(l_min); eq_max.merge(l_max); eq_fmin.merge(l_fmin);
			eq_fmax.merge(l_fmax); eq_clamp.merge(l_clamp); eq_mix.merge(l_mix);
		}
	}
	put_ulp("atan2", "finite (y, x), not both zero: random + edge grid", atan2_a, 6.0);
	put_ulp("pow", "x in [2^-20, 4], y in [0, 1]: random + edge grid", pow_a, 16.0);
	put_eq("divide", div_a);
	put_eq("min", eq_min); put_eq("max", eq_max); put_eq("fmin", eq_fmin); put_eq("fmax", eq_fmax); put_eq("clamp", eq_clamp); put_eq("mix", eq_mix);

	// conversions (s6.2.3, round to nearest even): uint -> float of the PRNG (random_sampler.cl:15) against an integer-only rounding
	Acc conv;
#pragma omp parallel
	{
		Acc a;
#pragma omp for schedule(static)
		for (int64_t k = 0; k < (int64_t)((0x100000000ull + stride - 1) / stride); k++) {
			const uint32_t u = (uint32_t)((uint64_t)k * stride);
			uint32_t want = 0;
			if (u) {
				const int msb = 31 - __builtin_clz(u);
				uint64_t m = (uint64_t)u << (63 - msb);          // normalised: bit 63 set
				uint32_t frac = (uint32_t)(m >> 40);              // 24 significant bits
				const uint64_t rest = m & ((1ull << 40) - 1), half = 1ull << 39;
				if (rest > half || (rest == half && (frac & 1u))) frac++;
				int e = msb;
				if (frac == (1u << 24)) { frac >>= 1; e++; }
				want = ((uint32_t)(e + 127) << 23) | (frac & 0x7fffffu);
			}
			a.n++;
			if (f2u((float)u) != want) { if (!a.mismatches) a.worst = u; a.mismatches++; }
		}
#pragma omp critical
		conv.merge(a);
	}
	put_eq("convert_float_uint", conv);
	snprintf(buf, sizeof buf, "\"stride\": %llu, \"random_inputs\": %llu}", (unsigned long long)stride, (unsigned long long)n_random);
	out += buf;
	puts(out.c_str());
	return 0;
}
