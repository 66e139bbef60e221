// rcp_sweep.hip -- test tool (not product): over ALL 2^32 float bit patterns x, is a reciprocal built from v_rcp_f32 and
// Newton steps equal, bit for bit, to the correctly rounded 1.0f / x the kernels are built with
// (-fhip-fp32-correctly-rounded-divide-sqrt)?  Prints, per candidate, the number of mismatches and the range of |x| that
// holds them, split into "inside [lo, hi]" (the interval a caller would use the candidate on) and outside.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
//         tests/tools/rcp_sweep.hip -o /tmp/rcp_sweep && /tmp/rcp_sweep
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float cand(int which, float x) {
	float r = __builtin_amdgcn_rcpf(x);
	float e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	if (which == 0) return r;
	e = __builtin_fmaf(-x, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	return r;
}

struct Res { unsigned long long bad_in, bad_out; uint32_t min_bad_in, max_bad_in, first_bad_in; };

__global__ void sweep(int which, float lo, float hi, Res *res) {
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	unsigned long long bad_in = 0, bad_out = 0;
	uint32_t mn = 0xFFFFFFFFu, mx = 0, first = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
		const uint32_t b = (uint32_t)i;
		const float x = __uint_as_float(b);
		const float want = 1.0f / x;
		const float got = cand(which, x);
		const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
		if (!same) {
			const float ax = __builtin_fabsf(x);
			if (ax >= lo && ax <= hi) {
				bad_in++;
				const uint32_t ab = b & 0x7FFFFFFFu;
				if (ab < mn) mn = ab;
				if (ab > mx) mx = ab;
				first = b;
			} else bad_out++;
		}
	}
	if (bad_in) { atomicAdd(&res->bad_in, bad_in); atomicMin(&res->min_bad_in, mn); atomicMax(&res->max_bad_in, mx); res->first_bad_in = first; }
	if (bad_out) atomicAdd(&res->bad_out, bad_out);
}

int main(int argc, char **argv) {
	const float lo = argc > 1 ? (float)atof(argv[1]) : 1e-30f, hi = argc > 2 ? (float)atof(argv[2]) : 1e30f;
	Res *d, h;
	hipMalloc(&d, sizeof(Res));
	int fails = 0;
	for (int which = 0; which < 2; which++) {
		h = Res{0, 0, 0xFFFFFFFFu, 0, 0};
		hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(sweep, dim3(256 * 32), dim3(256), 0, 0, which, lo, hi, d);
		hipDeviceSynchronize();
		hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
		float fmn, fmx, ff;
		memcpy(&fmn, &h.min_bad_in, 4); memcpy(&fmx, &h.max_bad_in, 4); memcpy(&ff, &h.first_bad_in, 4);
		printf("candidate %d (v_rcp_f32 + %d Newton step%s): mismatches with |x| in [%g, %g]: %llu", which, which + 1, which ? "s" : "", lo, hi, h.bad_in);
		if (h.bad_in) printf(" (|x| from %g to %g, e.g. %a)", fmn, fmx, ff);
		printf("; outside: %llu\n", h.bad_out);
		if (which == 1 && h.bad_in) fails = 1;
	}
	return fails;
}
