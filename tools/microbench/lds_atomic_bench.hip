// Microbenchmark (not part of the product): ds_add_u32 throughput on gfx950 for different lane->address patterns.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench lds_atomic_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kWords = 24576; // 96 KB histogram
constexpr int kIters = 4096;

template <int PATTERN, bool RET>
__global__ __launch_bounds__(1024) void bench(unsigned* out, const unsigned* rnd)
{
	__shared__ unsigned hist[kWords];
	for (int i = threadIdx.x; i < kWords; i += 1024) hist[i] = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	unsigned s = rnd[threadIdx.x + blockIdx.x * 1024];
	unsigned acc = 0;
	for (int it = 0; it < kIters; ++it) {
		unsigned a;
		s = s * 1664525u + 1013904223u;
		const unsigned r = s >> 8;
		if (PATTERN == 0) a = (r & ~63u) % (kWords - 64) + lane;             // 64 consecutive words
		else if (PATTERN == 1) a = r % kWords;                                // random
		else if (PATTERN == 2) a = __shfl(r % kWords, 0);                     // one address per wave
		else if (PATTERN == 3) a = (r & ~127u) % (kWords - 128) + lane * 2;   // stride 2
		else if (PATTERN == 4) a = (r & ~63u) % (kWords - 64) + (lane & 31);  // 2 lanes per address
		else if (PATTERN == 5) a = ((r % (kWords / 64)) * 64) + lane;         // random row, bank == lane (conflict-free if 64 banks)
		else if (PATTERN == 6) a = ((r % (kWords / 32)) * 32) + (lane & 31);  // random row per lane, bank == lane&31
		else a = (r % (kWords / 32)) * 32;                                    // all lanes bank 0, random rows
		if (RET) acc += atomicAdd(&hist[a], 1u);
		else atomicAdd(&hist[a], 1u);
	}
	__syncthreads();
	unsigned v = acc;
	for (int i = threadIdx.x; i < kWords; i += 1024) v += hist[i];
	if (v == 0xdeadbeefu) out[0] = v;
}

template <int P, bool RET> void run(const char* name, unsigned* out, unsigned* rnd)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * 4;
	hipLaunchKernelGGL((bench<P, RET>), dim3(blocks), dim3(1024), 0, 0, out, rnd);
	hipEventRecord(e0);
	hipLaunchKernelGGL((bench<P, RET>), dim3(blocks), dim3(1024), 0, 0, out, rnd);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	const double votes = (double)blocks * 1024 * kIters;
	printf("%-44s ret=%d  %8.3f ms  %7.1f Gvotes/s  %5.2f lanes/clk/CU (2.4 GHz, 256 CUs)\n", name, (int)RET, ms, votes / ms * 1e-6, votes / (ms * 1e-3) / 256 / 2.4e9);
}

int main()
{
	unsigned *out, *rnd;
	hipMalloc(&out, 4);
	std::vector<unsigned> h(1024 * 1024);
	for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)rand() * 2654435761u + (unsigned)i;
	hipMalloc(&rnd, h.size() * 4);
	hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	run<0, false>("consecutive 64 words", out, rnd);
	run<1, false>("random", out, rnd);
	run<2, false>("single address per wave", out, rnd);
	run<3, false>("stride 2", out, rnd);
	run<4, false>("2 lanes per address (32 addresses)", out, rnd);
	run<5, false>("random rows, bank==lane (64)", out, rnd);
	run<6, false>("random rows, bank==lane&31", out, rnd);
	run<7, false>("all lanes same bank, random rows", out, rnd);
	run<0, true>("consecutive 64 words", out, rnd);
	run<1, true>("random", out, rnd);
	return 0;
}
