// sort_bench -- which rocPRIM onesweep configuration sorts the line keys fastest?  (experiment harness, not part of the product)
// Keys shaped like the SHT line keys of the benchmark: 5 frame bits | 13 strength bits, strengths mostly just above the threshold.
//   hipcc --offload-arch=gfx950 -O3 -o sort_bench sort_bench.hip ; ./sort_bench [n] [keyBits]
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int HT, int HI, int ST, int SI, int BITS, size_t MERGE = 1024 * 1024>
static float run(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int keyBits, const char* name, std::vector<uint32_t>& ref, bool check)
{
	using Onesweep = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<HT, HI>, rocprim::kernel_config<ST, SI>, BITS, rocprim::block_radix_rank_algorithm::match>;
	using Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, Onesweep, MERGE>;
	size_t tb = 0;
	CK(rocprim::radix_sort_pairs_desc<Config>(nullptr, tb, kin, kout, vin, vout, n, 0u, (unsigned)keyBits, 0));
	void* tmp; CK(hipMalloc(&tmp, tb));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int i = 0; i < 3; ++i) CK(rocprim::radix_sort_pairs_desc<Config>(tmp, tb, kin, kout, vin, vout, n, 0u, (unsigned)keyBits, 0));
	std::vector<float> ts;
	for (int i = 0; i < 20; ++i) {
		CK(hipEventRecord(e0, 0));
		CK(rocprim::radix_sort_pairs_desc<Config>(tmp, tb, kin, kout, vin, vout, n, 0u, (unsigned)keyBits, 0));
		CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
	}
	std::sort(ts.begin(), ts.end());
	std::vector<uint32_t> out(n);
	CK(hipMemcpy(out.data(), vout, n * 4, hipMemcpyDeviceToHost));
	const char* verdict = "";
	if (check) { if (ref.empty()) { ref = out; verdict = "reference"; } else verdict = (out == ref) ? "MATCH" : "DIFFERS"; }
	printf("%-34s n=%zu  median %.4f ms  min %.4f  %s\n", name, n, ts[ts.size() / 2], ts.front(), verdict);
	CK(hipFree(tmp));
	return ts[ts.size() / 2];
}

int main(int argc, char** argv)
{
	const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1960000;
	const int keyBits = argc > 2 ? atoi(argv[2]) : 18;
	std::vector<uint32_t> k(n), v(n);
	uint32_t s = 12345;
	const size_t per = (n + 31) / 32;
	for (size_t i = 0; i < n; ++i) {
		s = s * 1664525u + 1013904223u;
		const uint32_t r = s >> 8;
		const uint32_t strength = 100 + ((r & 7) ? (r >> 3) % 60 : (r >> 3) % 3000);   // most lines just above the threshold
		k[i] = ((31u - (uint32_t)(i / per)) << 13) | strength;
		v[i] = (uint32_t)i;
	}
	uint32_t *kin, *kout, *vin, *vout;
	CK(hipMalloc(&kin, n * 4)); CK(hipMalloc(&kout, n * 4)); CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
	CK(hipMemcpy(kin, k.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vin, v.data(), n * 4, hipMemcpyHostToDevice));
	std::vector<uint32_t> ref;
	run<1024, 8, 1024, 10, 10>(kin, kout, vin, vout, n, keyBits, "shipped  h1024x8 s1024x10 b10 (merge sort <= 1M)", ref, true);
	run<1024, 8, 1024, 10, 10, 0>(kin, kout, vin, vout, n, keyBits, "h1024x8 s1024x10 b10 onesweep", ref, true);
	run<1024, 8, 1024, 16, 10, 0>(kin, kout, vin, vout, n, keyBits, "h1024x8 s1024x16 b10 onesweep", ref, true);
	run<1024, 16, 1024, 12, 10, 0>(kin, kout, vin, vout, n, keyBits, "h1024x16 s1024x12 b10 onesweep", ref, true);
	run<1024, 8, 1024, 4, 10, 0>(kin, kout, vin, vout, n, keyBits, "h1024x8 s1024x4 b10 onesweep", ref, true);
	run<512, 8, 512, 8, 10, 0>(kin, kout, vin, vout, n, keyBits, "h512x8 s512x8 b10 onesweep", ref, true);
	run<512, 4, 512, 4, 10, 0>(kin, kout, vin, vout, n, keyBits, "h512x4 s512x4 b10 onesweep", ref, true);
	run<256, 8, 256, 8, 9, 0>(kin, kout, vin, vout, n, keyBits, "h256x8 s256x8 b9 onesweep", ref, true);
	run<256, 4, 256, 4, 9, 0>(kin, kout, vin, vout, n, keyBits, "h256x4 s256x4 b9 onesweep", ref, true);
	run<256, 4, 256, 4, 8, 0>(kin, kout, vin, vout, n, keyBits, "h256x4 s256x4 b8 onesweep", ref, true);
	return 0;
}
