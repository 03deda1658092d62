// Microbenchmark (not part of the product): does a ds_add_u32 wave-instruction slow down when some of its lanes hit the SAME address as the previous
// instruction of the wave (adjacent edge pixels vote for the same (rho, theta) cell for every theta whose |cos| or |sin| is small)?
// Addresses: bank == lane & 31 always (the voting kernel's layout); with probability q a lane repeats the row of its previous instruction, otherwise a random row.
// Also: the same repetition but between DIFFERENT waves of the workgroup (wave w + 1 repeats wave w's rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRows = 1216;           // window rows of the 4K plan
constexpr int kIters = 1024;
constexpr int kA = 16;

__global__ __launch_bounds__(1024) void bench(unsigned* out, const unsigned* addr, int pattern)
{
	extern __shared__ unsigned hist[];
	for (int i = threadIdx.x; i < kRows * 32; i += 1024) hist[i] = 0;
	__syncthreads();
	unsigned a[kA];
	for (int k = 0; k < kA; ++k) a[k] = addr[((size_t)pattern * 1024 + threadIdx.x) * kA + k];
	const unsigned inc = (threadIdx.x & 32) ? 0x10000u : 1u;
	for (int it = 0; it < kIters; ++it) {
#pragma unroll
		for (int k = 0; k < kA; ++k) asm volatile("ds_add_u32 %0, %1" : : "v"(a[k]), "v"(inc) : "memory");
	}
	__builtin_amdgcn_s_waitcnt(0xc07f);
	__syncthreads();
	unsigned v = 0;
	for (int i = threadIdx.x; i < kRows * 32; i += 1024) v += hist[i];
	if (v == 0xdeadbeefu) out[0] = v;
}

int main()
{
	const float qs[] = { 0.f, 0.25f, 0.5f, 0.75f, 1.f };
	const int NQ = 5, NP = 2 * NQ;
	std::vector<unsigned> h((size_t)NP * 1024 * kA);
	srand(7);
	auto rnd = [] { return ((unsigned)rand() * 2654435761u) >> 7; };
	for (int p = 0; p < NP; ++p) {
		const float q = qs[p % NQ];
		const bool cross = p >= NQ;
		for (int t = 0; t < 1024; ++t)
			for (int k = 0; k < kA; ++k) {
				const int lane = t & 63;
				unsigned row = rnd() % kRows;
				const bool rep = (float)(rnd() % 1000) < q * 1000.f;
				if (rep) {
					if (!cross && k > 0) row = h[((size_t)p * 1024 + t) * kA + k - 1] / 128;            // same row as this lane's previous instruction
					if (cross && t >= 64) row = h[((size_t)p * 1024 + t - 64) * kA + k] / 128;           // same row as the lane of the previous wave, same instruction slot
				}
				h[((size_t)p * 1024 + t) * kA + k] = row * 128 + (lane & 31) * 4;                       // byte address
			}
	}
	unsigned *out, *addr;
	(void)hipMalloc(&out, 4);
	(void)hipMalloc(&addr, h.size() * 4);
	(void)hipMemcpy(addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	const size_t lds = (size_t)kRows * 128;
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(bench), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for (int p = 0; p < NP; ++p) {
		const int blocks = 256 * 2;
		hipLaunchKernelGGL(bench, dim3(blocks), dim3(1024), lds, 0, out, addr, p);
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(bench, dim3(blocks), dim3(1024), lds, 0, out, addr, p);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
		const double instr = (double)blocks * 16 * kIters * kA;
		printf("%s  q = %.2f: %8.3f ms  %6.2f clk per ds_add_u32 wave-instruction per CU (2.4 GHz)\n", p >= NQ ? "repeat the previous WAVE's rows      " : "repeat the lane's previous instruction", qs[p % NQ], ms,
		       ms * 1e-3 * 2.4e9 / (instr / 256));
	}
	return 0;
}
