// cl_builtins.cpp -- host definitions of the 38 OpenCL C built-ins that the reference's
// device code (compiled IN PLACE from /root/reference/tracer/opencl/CL/main.cl, never
// copied) leaves undefined when it is built for x86-64 instead of an OpenCL device.
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref).  Nothing in the product links this.
//
// The image has no OpenCL CPU device (clinfo: 0 devices), so the run-time half of OpenCL C
// -- work-item functions and the math/geometric built-in library -- has to come from
// somewhere.  Each built-in below is implemented exactly as the OpenCL 1.2 specification
// (section 6.12) describes it; where the specification leaves the result implementation
// defined (native_*, and the ulp slack of sqrt / atan / acos / pow / normalize) one of two
// concrete choices is compiled in:
//
//   default            include/polaris_math.h  ("pm" build)  -> bit-reproducible everywhere;
//                      this is the definition the CPU restatement and the HIP kernels use,
//                      so reference == restatement == HIP can be checked bit for bit.
//   -DPOLARIS_REF_LIBM glibc libm                ("libm" build) -> an independent second
//                      opinion; used for the statistical (RMSE) cross-check only.
//
// Execution model: work-items run one at a time in ascending
// global id with work-group size 1 -- what the reference itself does on CPU devices
// (tracer/opencl/pipeline.go:105-111) -- so barrier() is a no-op, get_local_id() is 0 and
// the local/global atomics are plain read-modify-writes.
#include <cmath>
#include <cstdint>
#include <cstdio>

#include "polaris_math.h"

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint2 __attribute__((ext_vector_type(2)));
typedef unsigned char uchar4 __attribute__((ext_vector_type(4)));

// current work-item id, set by the driver (ref_driver.cpp) before each kernel call
thread_local size_t polaris_ref_gid[3] = {0, 0, 0};

#define CLFN(mangled) __asm__(mangled)

#ifdef POLARIS_REF_LIBM
static inline float x_sqrt(float x) { return sqrtf(x); }
static inline float x_cos(float x) { return cosf(x); }
static inline float x_sin(float x) { return sinf(x); }
static inline float x_atan(float x) { return atanf(x); }
static inline float x_atan2(float y, float x) { return atan2f(y, x); }
static inline float x_acos(float x) { return acosf(x); }
static inline float x_pow(float x, float y) { return powf(x, y); }
#else
static inline float x_sqrt(float x) { return pm_sqrt(x); }
static inline float x_cos(float x) { return pm_cos(x); }
static inline float x_sin(float x) { return pm_sin(x); }
static inline float x_atan(float x) { return pm_atan(x); }
static inline float x_atan2(float y, float x) { return pm_atan2(y, x); }
static inline float x_acos(float x) { return pm_acos(x); }
static inline float x_pow(float x, float y) { return pm_pow(x, y); }
#endif

// ---- work-item functions (s6.12.1) and synchronisation (s6.12.8) -----------------------
size_t cl_get_global_id(unsigned d) CLFN("_Z13get_global_idj");
size_t cl_get_global_id(unsigned d) { return d < 3 ? polaris_ref_gid[d] : 0; }
size_t cl_get_local_id(unsigned d) CLFN("_Z12get_local_idj");
size_t cl_get_local_id(unsigned) { return 0; }
void cl_barrier(unsigned f) CLFN("_Z7barrierj");
void cl_barrier(unsigned) {}

// ---- atomics (s6.12.11): return the old value ------------------------------------------
int cl_atomic_inc(volatile int *p) CLFN("_Z10atomic_incPU7CLlocalVi");
int cl_atomic_inc(volatile int *p) { int o = *p; *p = o + 1; return o; }
int cl_atomic_add(volatile int *p, int v) CLFN("_Z10atomic_addPU8CLglobalVii");
int cl_atomic_add(volatile int *p, int v) { int o = *p; *p = o + v; return o; }

// ---- math (s6.12.2) ---------------------------------------------------------------------
float cl_native_cos(float x) CLFN("_Z10native_cosf");
float cl_native_cos(float x) { return x_cos(x); }
float cl_native_sin(float x) CLFN("_Z10native_sinf");
float cl_native_sin(float x) { return x_sin(x); }
float cl_native_sqrt(float x) CLFN("_Z11native_sqrtf");
float cl_native_sqrt(float x) { return x_sqrt(x); }
float cl_native_recip(float x) CLFN("_Z12native_recipf");
float cl_native_recip(float x) { return 1.0f / x; }
float3 cl_native_recip3(float3 v) CLFN("_Z12native_recipDv3_f");
float3 cl_native_recip3(float3 v) { float3 r; r.x = 1.0f / v.x; r.y = 1.0f / v.y; r.z = 1.0f / v.z; return r; }
float cl_sqrt(float x) CLFN("_Z4sqrtf");
float cl_sqrt(float x) { return x_sqrt(x); }
float cl_acos(float x) CLFN("_Z4acosf");
float cl_acos(float x) { return x_acos(x); }
float cl_atan(float x) CLFN("_Z4atanf");
float cl_atan(float x) { return x_atan(x); }
float cl_atan2(float y, float x) CLFN("_Z5atan2ff");
float cl_atan2(float y, float x) { return x_atan2(y, x); }
float cl_fabs(float x) CLFN("_Z4fabsf");
float cl_fabs(float x) { return pm_fabs(x); }
float3 cl_pow3(float3 x, float3 y) CLFN("_Z3powDv3_fS_");
float3 cl_pow3(float3 x, float3 y) { float3 r; r.x = x_pow(x.x, y.x); r.y = x_pow(x.y, y.y); r.z = x_pow(x.z, y.z); return r; }
float2 cl_floor2(float2 v) CLFN("_Z5floorDv2_f");
float2 cl_floor2(float2 v) { float2 r; r.x = pm_floor(v.x); r.y = pm_floor(v.y); return r; }
float cl_fmax(float a, float b) CLFN("_Z4fmaxff");
float cl_fmax(float a, float b) { return pm_fmax(a, b); }
float cl_fmin(float a, float b) CLFN("_Z4fminff");
float cl_fmin(float a, float b) { return pm_fmin(a, b); }
float3 cl_fmax3(float3 a, float3 b) CLFN("_Z4fmaxDv3_fS_");
float3 cl_fmax3(float3 a, float3 b) { float3 r; r.x = pm_fmax(a.x, b.x); r.y = pm_fmax(a.y, b.y); r.z = pm_fmax(a.z, b.z); return r; }
float3 cl_fmin3(float3 a, float3 b) CLFN("_Z4fminDv3_fS_");
float3 cl_fmin3(float3 a, float3 b) { float3 r; r.x = pm_fmin(a.x, b.x); r.y = pm_fmin(a.y, b.y); r.z = pm_fmin(a.z, b.z); return r; }

// ---- common (s6.12.4) -------------------------------------------------------------------
float cl_max(float a, float b) CLFN("_Z3maxff");
float cl_max(float a, float b) { return pm_max(a, b); }
float cl_min(float a, float b) CLFN("_Z3minff");
float cl_min(float a, float b) { return pm_min(a, b); }
float cl_mix(float a, float b, float t) CLFN("_Z3mixfff");
float cl_mix(float a, float b, float t) { return pm_mix(a, b, t); }
float4 cl_mix4(float4 a, float4 b, float t) CLFN("_Z3mixDv4_fS_f");
float4 cl_mix4(float4 a, float4 b, float t) {
	float4 r;
	r.x = pm_mix(a.x, b.x, t); r.y = pm_mix(a.y, b.y, t); r.z = pm_mix(a.z, b.z, t); r.w = pm_mix(a.w, b.w, t);
	return r;
}
float cl_sign(float x) CLFN("_Z4signf");
float cl_sign(float x) { return pm_sign(x); }
float cl_clamp(float x, float lo, float hi) CLFN("_Z5clampfff");
float cl_clamp(float x, float lo, float hi) { return pm_clamp(x, lo, hi); }
float3 cl_clamp3(float3 v, float lo, float hi) CLFN("_Z5clampDv3_fff");
float3 cl_clamp3(float3 v, float lo, float hi) { float3 r; r.x = pm_clamp(v.x, lo, hi); r.y = pm_clamp(v.y, lo, hi); r.z = pm_clamp(v.z, lo, hi); return r; }
int cl_clampi(int x, int lo, int hi) CLFN("_Z5clampiii");
int cl_clampi(int x, int lo, int hi) { return pm_clampi(x, lo, hi); }
unsigned cl_clampu(unsigned x, unsigned lo, unsigned hi) CLFN("_Z5clampjjj");
unsigned cl_clampu(unsigned x, unsigned lo, unsigned hi) { return pm_clampu(x, lo, hi); }

// ---- geometric (s6.12.5) ----------------------------------------------------------------
// dot = x*x' + y*y' + z*z' summed left to right; length = sqrt(dot); normalize = v * (1/length).
float cl_dot3(float3 a, float3 b) CLFN("_Z3dotDv3_fS_");
float cl_dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float3 cl_cross3(float3 a, float3 b) CLFN("_Z5crossDv3_fS_");
float3 cl_cross3(float3 a, float3 b) {
	float3 r;
	r.x = a.y * b.z - a.z * b.y;
	r.y = a.z * b.x - a.x * b.z;
	r.z = a.x * b.y - a.y * b.x;
	return r;
}
float cl_length3(float3 v) CLFN("_Z6lengthDv3_f");
float cl_length3(float3 v) { return x_sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
float3 cl_normalize3(float3 v) CLFN("_Z9normalizeDv3_f");
float3 cl_normalize3(float3 v) {
	float inv = 1.0f / x_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
	float3 r; r.x = v.x * inv; r.y = v.y * inv; r.z = v.z * inv;
	return r;
}
float4 cl_normalize4(float4 v) CLFN("_Z9normalizeDv4_f");
float4 cl_normalize4(float4 v) {
	float inv = 1.0f / x_sqrt(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
	float4 r; r.x = v.x * inv; r.y = v.y * inv; r.z = v.z * inv; r.w = v.w * inv;
	return r;
}

// ---- conversions (s6.2.3): default rounding = round to nearest even ---------------------
float2 cl_convert_float2(uint2 v) CLFN("_Z14convert_float2Dv2_j");
float2 cl_convert_float2(uint2 v) { float2 r; r.x = (float)v.x; r.y = (float)v.y; return r; }
float4 cl_convert_float4(uchar4 v) CLFN("_Z14convert_float4Dv4_h");
float4 cl_convert_float4(uchar4 v) { float4 r; r.x = (float)v.x; r.y = (float)v.y; r.z = (float)v.z; r.w = (float)v.w; return r; }
