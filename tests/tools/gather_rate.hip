// gather_rate.hip -- calibration microbenchmark (not product): what does the memory side of a BVH traversal step cost on
// gfx950?  Every lane (or every quad of lanes) walks a chain of DEPENDENT record fetches at uniformly random places of a
// table -- the access pattern of k_trace's node fetch (kernels.h) -- and the program reports ns per chain step and the
// chip-wide rate of gathered bytes, by record shape, table size (L2 / Infinity Cache / HBM resident) and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate
//
// shapes:  lane64   one lane = one chain, 64-B records  (4 x global_load_dwordx4 per lane and step: today's pair node)
//          lane128  one lane = one chain, 128-B records (8 loads: a four-wide node per lane)
//          quad64   four lanes = one chain, 64-B records (1 load per lane: the quad reads one contiguous line segment)
//          quad128  four lanes = one chain, 128-B records (2 loads per lane: a four-wide node, one box per lane)
//          pair128  two lanes = one chain, 128-B records (4 loads per lane)
//          lane32   one lane = one chain, 32-B records  (2 loads: a pair node with both boxes quantised to 16 bits per coordinate)
//          lane48   one lane = one chain, 48-B records  (3 loads)
//          lane16   one lane = one chain, 16-B records  (1 load: the floor of a per-lane gather)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

enum Shape { kLane64 = 0, kLane128, kQuad64, kQuad128, kPair128, kLane32, kLane48, kLane16, kShapes };
static const char *kShapeName[kShapes] = {"lane64", "lane128", "quad64", "quad128", "pair128", "lane32", "lane48", "lane16"};
static const int kShapeRecBytes[kShapes] = {64, 128, 64, 128, 128, 32, 48, 16};

// record r starts with the index of the next record of the chain that passes through it (a random permutation cycle), the rest
// is payload that is summed so the loads cannot be dropped
template <int SHAPE>
__global__ __launch_bounds__(256) void walk(const float4 *table, uint32_t num_records, int steps, uint32_t live_mask, float *out) {
	constexpr int REC4 = SHAPE == kLane32 ? 2 : SHAPE == kLane48 ? 3 : SHAPE == kLane16 ? 1 : (SHAPE == kLane64 || SHAPE == kQuad64) ? 4 : 8; // float4s per record
	constexpr int TEAM = (SHAPE == kQuad64 || SHAPE == kQuad128) ? 4 : (SHAPE == kPair128 ? 2 : 1); // lanes per chain
	constexpr int PER = REC4 / TEAM;                                               // float4 loads per lane and step
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63;
	// dead lanes (divergence): do they cost the L1?  One bit per lane (mod 32) for the one-lane shapes, one per TEAM otherwise
	if (((live_mask >> ((TEAM == 1 ? lane : lane / TEAM) & 31)) & 1u) == 0u) return;
	const uint32_t chain = tid / TEAM, part = tid % TEAM;
	uint32_t cur = (uint32_t)(((uint64_t)chain * 2654435761ull) % num_records);
	float acc = 0.0f;
	for (int s = 0; s < steps; s++) {
		const float4 *rec = table + (size_t)cur * REC4 + part * PER;
		float4 v[PER];
#pragma unroll
		for (int k = 0; k < PER; k++) v[k] = rec[k];
#pragma unroll
		for (int k = 0; k < PER; k++) acc += v[k].y + v[k].z + v[k].w + (k ? v[k].x : 0.0f);
		uint32_t next = (uint32_t)__float_as_int(v[0].x);                           // (lane `part == 0` of the team holds the link)
		if (TEAM > 1) next = (uint32_t)__shfl((int)next, (int)(lane & ~(uint32_t)(TEAM - 1)));
		cur = next;
	}
	out[tid] = acc + (float)cur;
}

int main(int argc, char **argv) {
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
	const int cus = p.multiProcessorCount;
	const int steps = argc > 1 ? atoi(argv[1]) : 2000;
	float *out;
	hipMalloc(&out, sizeof(float) * (size_t)cus * 8 * 256 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const size_t table_bytes[] = {1u << 20, 3u << 20, 24u << 20, 96u << 20};
	std::mt19937 rng(7);
	for (size_t tb : table_bytes) {
		for (int rec_bytes : {16, 32, 48, 64, 128}) {
			const uint32_t n = (uint32_t)(tb / rec_bytes);
			// one random cycle through all records (Sattolo): every chain step lands on a fresh, uniformly distributed record
			std::vector<uint32_t> perm(n);
			for (uint32_t i = 0; i < n; i++) perm[i] = i;
			for (uint32_t i = n - 1; i > 0; i--) { std::uniform_int_distribution<uint32_t> d(0, i - 1); std::swap(perm[i], perm[d(rng)]); }
			std::vector<float> host((size_t)n * rec_bytes / 4, 1.0f);
			for (uint32_t i = 0; i < n; i++) memcpy(&host[(size_t)i * rec_bytes / 4], &perm[i], 4);
			float4 *table;
			hipMalloc(&table, (size_t)n * rec_bytes);
			hipMemcpy(table, host.data(), (size_t)n * rec_bytes, hipMemcpyHostToDevice);
			for (int shape = 0; shape < kShapes; shape++) {
				if (kShapeRecBytes[shape] != rec_bytes) continue;
				const int team = (shape == kQuad64 || shape == kQuad128) ? 4 : (shape == kPair128 ? 2 : 1);
				for (uint32_t live : {0xFFFFFFFFu, 0x55555555u, 0x11111111u}) {
					for (int wps : {2, 4, 8}) {
						const int blocks = cus * wps;
						float ms = 0;
						for (int rep = 0; rep < 2; rep++) {
							hipEventRecord(e0);
							void (*fn)(const float4 *, uint32_t, int, uint32_t, float *) = shape == kLane64 ? walk<kLane64> : shape == kLane128 ? walk<kLane128> : shape == kQuad64 ? walk<kQuad64> : shape == kQuad128 ? walk<kQuad128> :
								shape == kPair128 ? walk<kPair128> : shape == kLane32 ? walk<kLane32> : shape == kLane48 ? walk<kLane48> : walk<kLane16>;
							hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, table, n, steps, live, out);
							hipEventRecord(e1);
							hipEventSynchronize(e1);
							hipEventElapsedTime(&ms, e0, e1);
						}
						const double lanes_live = live == 0xFFFFFFFFu ? 1.0 : (live == 0x55555555u ? 0.5 : 0.25);
						const double chains = (double)blocks * 256 / team * lanes_live;
						const double fetches = chains * steps;
						printf("table %5zu MB  %-8s live=%.2f waves/SIMD=%d  %8.3f ms  %7.1f ns per dependent step  %7.2f G records/s  %6.2f TB/s gathered  %6.1f ns of CU time per wave-step\n",
						       tb >> 20, kShapeName[shape], lanes_live, wps, ms, ms * 1e6 / steps, fetches / (ms * 1e6), fetches * rec_bytes / (ms * 1e9),
						       ms * 1e6 / steps / (wps * 4));
					}
				}
			}
			hipFree(table);
		}
	}
	return 0;
}
