// Microbenchmark (not part of the product): pure ds_add_u32 issue rate on gfx950 -- addresses are computed once,
// the timed loop holds only LDS atomics (8 per iteration, addresses in registers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kWords = 24576; // 96 KB
constexpr int kIters = 2048;
constexpr int kA = 8;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bench(unsigned* out, const unsigned* addr, int pattern)
{
	__shared__ unsigned hist[kWords];
	for (int i = threadIdx.x; i < kWords; i += WAVES * 64) hist[i] = 0;
	__syncthreads();
	unsigned a[kA];
	for (int k = 0; k < kA; ++k) a[k] = addr[(size_t)pattern * 1024 * kA + (threadIdx.x % 1024) * kA + k];
	for (int it = 0; it < kIters; ++it) {
#pragma unroll
		for (int k = 0; k < kA; ++k) atomicAdd(&hist[a[k]], 1u);
	}
	__syncthreads();
	unsigned v = 0;
	for (int i = threadIdx.x; i < kWords; i += WAVES * 64) v += hist[i];
	if (v == 0xdeadbeefu) out[0] = v;
}

template <int WAVES> void run(const char* name, unsigned* out, unsigned* addr, int pattern)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int blocks = 256 * 4;
	hipLaunchKernelGGL((bench<WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, addr, pattern);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((bench<WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, addr, pattern);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	const double votes = (double)blocks * WAVES * 64 * kIters * kA;
	printf("%-40s waves/CU=%2d %8.3f ms %8.1f Gvotes/s %6.2f lanes/clk/CU  %6.1f clk/instr\n", name, WAVES, ms, votes / ms * 1e-6,
	       votes / (ms * 1e-3) / 256 / 2.4e9, 64.0 / (votes / (ms * 1e-3) / 256 / 2.4e9));
}

int main()
{
	unsigned *out, *addr;
	(void)hipMalloc(&out, 4);
	const int NP = 8;
	std::vector<unsigned> h((size_t)NP * 1024 * kA);
	srand(1);
	for (int p = 0; p < NP; ++p)
		for (int t = 0; t < 1024; ++t)
			for (int k = 0; k < kA; ++k) {
				const int lane = t & 63;
				unsigned r = ((unsigned)rand() * 2654435761u) >> 7;
				unsigned a;
				switch (p) {
				case 0: a = (r % (kWords / 64)) * 64 + lane; break;                       // bank == lane (64 distinct banks if 64 banks)
				case 1: a = (r % (kWords / 32)) * 32 + (lane & 31); break;                // bank == lane&31, lanes l and l+32 different rows
				case 2: a = r % kWords; break;                                             // random
				case 3: a = (r % (kWords / 64)) * 64 + (lane & 15); break;                // 16 banks, 4 lanes each, different rows
				case 4: a = (r % (kWords / 64)) * 64; break;                              // one bank, 64 rows
				case 5: { static unsigned rowk[16 * kA]; if (lane == 0) rowk[(t >> 6) * kA + k] = r % (kWords / 64); a = rowk[(t >> 6) * kA + k] * 64 + (lane & 31); break; } // 2 lanes per address, 32 addresses consecutive
				case 6: { static unsigned rk[16 * kA]; if (lane == 0) rk[(t >> 6) * kA + k] = r % kWords; a = rk[(t >> 6) * kA + k]; break; } // one address per wave
				default: a = (r % (kWords / 64)) * 64 + (lane & 31) * 2; break;           // even banks only, 2 lanes per bank, different rows
				}
				h[((size_t)p * 1024 + t) * kA + k] = a;
			}
	(void)hipMalloc(&addr, h.size() * 4);
	(void)hipMemcpy(addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	const char* names[NP] = { "bank==lane (64 rows random)", "bank==lane&31", "random", "16 banks x 4 lanes", "1 bank x 64 rows", "32 consecutive addrs x 2 lanes", "1 address per wave", "even banks, 2 lanes each" };
	for (int p = 0; p < NP; ++p) run<16>(names[p], out, addr, p);
	for (int p = 0; p < 3; ++p) run<8>(names[p], out, addr, p);
	for (int p = 0; p < 3; ++p) run<4>(names[p], out, addr, p);
	return 0;
}
