// valu_rate.hip -- calibration microbenchmark (not product): how many cycles does a wave64 VALU instruction cost a
// SIMD of gfx950 when 1 / 2 / 4 / 8 waves share it?  Decides how to read SQ_INSTS_VALU against kernel time.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

// KIND 3 / 4: the same 16 floats per lane as 8 pairs through v_pk_mul_f32 / v_pk_add_f32 -- does a packed f32 instruction
// cost a SIMD what a plain one costs (twice the work per issue slot), or twice that?
template <int KIND>
__global__ void spin_pk(float *out, int iters, unsigned long long *ticks) {
	v2f a[8];
	for (int i = 0; i < 8; i++) a[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
	const v2f b = {1.0001f, 1.0002f};
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 8; r++)
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
			}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
	for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int KIND>
__global__ void spin(float *out, int iters, unsigned long long *ticks) {
	float a[16];
	for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
	const float b = 1.0001f, c = 0.0001f;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 4; r++)
#pragma unroll
			for (int i = 0; i < 16; i++) {
				if (KIND == 0) a[i] = __builtin_fmaf(a[i], b, c);       // v_fma_f32
				else if (KIND == 1) a[i] = a[i] * b;                     // v_mul_f32
				else a[i] = a[i] > 1.0f ? a[i] - b : a[i] + c;           // cmp + 2 ops + cndmask
			}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
	for (int i = 0; i < 16; i++) s += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

int main() {
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	float *out;
	unsigned long long *ticks, h;
	hipMalloc(&out, sizeof(float) * cus * 64 * 1024);
	hipMalloc(&ticks, 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 20000;
	for (int kind = 0; kind < 5; kind++)
		for (int wps : {1, 2, 4, 8}) { // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD
			const int blocks = cus * wps;
			for (int rep = 0; rep < 2; rep++) {
				hipEventRecord(e0);
				if (kind == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, out, iters, ticks);
				else if (kind == 1) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, out, iters, ticks);
				else if (kind == 2) hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), 0, 0, out, iters, ticks);
				else if (kind == 3) hipLaunchKernelGGL(spin_pk<3>, dim3(blocks), dim3(256), 0, 0, out, iters, ticks);
				else hipLaunchKernelGGL(spin_pk<4>, dim3(blocks), dim3(256), 0, 0, out, iters, ticks);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
			}
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
			const double instr_per_simd = (double)iters * 64 * wps * (kind == 2 ? 4 : 1); // wave-instructions issued on one SIMD (kind 2: cmp, sub, add, cndmask per element; kinds 3-4: 64 packed instructions per iteration)
			printf("kind=%d waves/SIMD=%d  %.3f ms  %.2f ns per wave64 VALU instr per SIMD  s_memtime ticks=%llu (%.1f MHz)\n", kind, wps, ms,
			       ms * 1e6 / instr_per_simd, h, h / (ms * 1e3));
		}
	return 0;
}
