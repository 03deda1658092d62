// sht_sort_kernels.hip -- the line sort of the device-resident SHT, sized by the lines that exist (gfx950, hand-written HIP).
//
// Replaces the tail of CompVHoughSht::process: std::sort(lines, strength descending) + the maxLines cut, core/features/hough/compv_core_feature_houghsht.cxx:241-249
// (canonical order of the plan API: strength descending, ties in nms_apply's emission order = accumulator (row, col) ascending, :546-562).
//
// Why not the library sort: rocPRIM's device radix sort takes its size from the HOST and costs 35-50 us whatever the size (histogram + two onesweep passes
// + their fills: 0.050 ms for 75 000 keys, 0.076 ms for 1.96 M; tools/microbench/sort_bench) -- at 1080p the sort of 75 000 real keys was 22 % of a step.
// The line keys have structure a general sort cannot use: a strength has at most 13 bits (a cell never exceeds 2 max(W, H)), the emission order inside a frame
// is the wanted tie order, frames are independent.  So the sort is a counting sort on the strength, per frame, whose stable ranks come from sorting CHUNKS of
// 4096 consecutive lines in the LDS -- a stable LSD radix sort of the 13 strength bits, ranks from wave ballots -- and every size is read on the
// device: the three launches cover the capacity, a chunk without lines returns at once.
//   1. sht_chunk_sort_kernel   one workgroup per chunk: stable sort of the chunk's keys in the LDS; per line its rank inside its run of equal strengths
//                              (lower bound by binary search), the chunk's strength histogram (u16 [8192], written whole: nothing to clear)
//   2. sht_strength_scan_kernel  one workgroup per frame: lines per strength over the frame's chunks, exclusive scan in descending strength order
//   3. sht_place_lines_kernel  one thread per line: slot = start[strength] + lines of that strength in the frame's earlier chunks + rank; the slot's
//                              compvhip_line is written directly (rho, theta, strength, row, col: what sht_decode_kernel did behind the library sort)
// Larger strengths (max(W, H) > 4095) or line capacities beyond 32 chunks per frame keep the library sort (sht_kernels.hip).
#include "kernels.hpp"

#include <type_traits>

namespace compvhip {

namespace {

constexpr int kChunk = kShtSortChunk;        // lines per chunk
constexpr int kSortThreads = 1024;
constexpr int kBins = 1 << kShtSortMaxStrengthBits;   // 8192 strengths
static_assert(kChunk == 4096 && kBins == 8192, "key layout: 13 bits of inverted strength above 12 bits of chunk position");

struct LineOut { float rho; float theta; int32_t strength; int32_t row; int32_t col; };

// lines of the earlier frames (clamped to lineCap): where frame `frame` starts in the dense key array.  Every thread of the block gets the sum.
template <int THREADS>
__device__ __forceinline__ size_t frame_base(const int* __restrict__ counts, int frame, size_t lineCap, unsigned long long* s_part)
{
	unsigned long long before = 0;
	for (int g = threadIdx.x; g < frame; g += THREADS) {
		const size_t cg = (size_t)max(counts[g], 0);
		before += cg < lineCap ? cg : lineCap;
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = before;
	__syncthreads();
	unsigned long long sum = 0;
#pragma unroll
	for (int w = 0; w < THREADS / 64; ++w) sum += s_part[w];
	return (size_t)sum;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void sht_chunk_sort_kernel(ShtArgs a, ShtSortArgs q)
{
	__shared__ uint32_t s_key[kChunk];
	__shared__ uint32_t s_hist[kBins];
	__shared__ uint32_t s_cnt[kSortThreads], s_wsum[kSortThreads / 64];
	__shared__ unsigned long long s_part[kSortThreads / 64];
	const int frame = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
	const size_t nf = min((size_t)max(a.lineCounts[frame], 0), a.lineCap);
	const size_t c0 = (size_t)chunk * kChunk;
	if (c0 >= nf) return;   // uniform: a chunk without lines
	const int m = (int)min((size_t)kChunk, nf - c0);
	const size_t base = frame_base<kSortThreads>(a.lineCounts, frame, a.lineCap, s_part) + c0;
	const uint32_t mask = (1u << a.strengthBits) - 1u;
	const uint32_t* __restrict__ kin = a.lineKeys + base;
	const uint32_t* __restrict__ vin = a.lineVals + base;
	// thread t holds the chunk's elements u * 1024 + t, u = 0 .. 3 (a wave = 64 consecutive elements of each quarter)
	uint32_t e[4];
	{
		uint32_t kraw[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) kraw[u] = kin[min(u * kSortThreads + t, m - 1)];   // clamped, and all four in flight: a load written under `i < m` is compiled into
		asm volatile("" : "+v"(kraw[0]), "+v"(kraw[1]), "+v"(kraw[2]), "+v"(kraw[3]));   // branch -> load -> s_waitcnt, one memory latency each
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int i = u * kSortThreads + t;
			e[u] = i < m ? (((mask - (kraw[u] & mask)) << 12) | (uint32_t)i) : 0xffffffffu;
		}
	}
#pragma unroll
	for (int u = 0; u < kBins / kSortThreads; ++u) s_hist[t + u * kSortThreads] = 0u;
	// Stable LSD radix sort of the 13 inverted-strength bits in the LDS, digits of 4 + 3 + 3 + 3 bits (the position bits below them only ride along: equal
	// strengths keep the emission order because every pass is stable).  Per pass and element: the lanes of its wave with the same digit (one ballot per
	// digit bit) give its rank inside (quarter, wave); the counts of every (digit, quarter, wave) -- 1024 of them at most, one per thread -- are scanned
	// over the block in exactly that order, which is digit-major and, inside a digit, the elements' order.  A bitonic network on the unique key
	// (strength, position) does the same job with 45 of its 78 steps as wave shuffles: ds_bpermute issues at ~16 cycles on the CU's one LDS pipe, 28 us of
	// a 38 us workgroup (rocprofv3, profiles/r05); this version needs five barriers and a dozen plain LDS accesses per pass.
	const int lane = t & 63, wave = t >> 6;
	auto radix_pass = [&](auto nbits, int shift) {
		constexpr int NB = decltype(nbits)::value, BINS = 1 << NB;
		s_cnt[t] = 0u;
		__syncthreads();
		uint32_t rank[4], slot[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t d = (e[u] >> shift) & (uint32_t)(BINS - 1);
			uint64_t same = ~0ull;
#pragma unroll
			for (int b = 0; b < NB; ++b) {
				const bool bit = (d >> b) & 1u;
				const uint64_t bal = __ballot(bit);
				same &= bit ? bal : ~bal;
			}
			rank[u] = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
			slot[u] = (d * 4u + (uint32_t)u) * (uint32_t)(kSortThreads / 64) + (uint32_t)wave;
			if (rank[u] == 0u) s_cnt[slot[u]] = (uint32_t)__popcll(same);   // the group's first lane
		}
		__syncthreads();
		// exclusive scan of the 1024 counts, one per thread (unused entries of a pass with fewer digits are zero)
		const uint32_t c = s_cnt[t];
		uint32_t incl = c;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t n = __shfl_up(incl, o);
			if (lane >= o) incl += n;
		}
		if (lane == 63) s_wsum[wave] = incl;
		__syncthreads();
		uint32_t before = 0;
#pragma unroll
		for (int w = 0; w < kSortThreads / 64; ++w) before += (w < wave) ? s_wsum[w] : 0u;
		s_cnt[t] = before + incl - c;
		__syncthreads();
#pragma unroll
		for (int u = 0; u < 4; ++u) s_key[s_cnt[slot[u]] + rank[u]] = e[u];
		__syncthreads();
#pragma unroll
		for (int u = 0; u < 4; ++u) e[u] = s_key[u * kSortThreads + t];
	};
	radix_pass(std::integral_constant<int, 4>{}, 12);
	radix_pass(std::integral_constant<int, 3>{}, 16);
	radix_pass(std::integral_constant<int, 3>{}, 19);
	radix_pass(std::integral_constant<int, 3>{}, 22);
	// s_key holds the sorted chunk (the last pass wrote it, and every thread has passed the barrier behind that)
	// rank of a line inside its run of equal strengths = its position - the run's first position (lower bound of (inv << 12))
	uint32_t* __restrict__ kout = q.sortedKeys + base;
	uint32_t* __restrict__ vout = q.sortedVals + base;
	{
		// four lines per thread (i = t + 1024 u: coalesced stores), their searches and gathers interleaved: 12 dependent LDS reads and one global
		// gather per line, one after the other, were 8 us of this kernel
		uint32_t kk[4]; int lo[4], hi[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { const int i = min(t + u * kSortThreads, m - 1); kk[u] = s_key[i]; lo[u] = 0; hi[u] = i; }
#pragma unroll
		for (int s = 0; s < 12; ++s) {
#pragma unroll
			for (int u = 0; u < 4; ++u) {   // the first position whose key >= (inv << 12), in [0, i]
				const int mid = (lo[u] + hi[u]) >> 1;
				const bool ge = s_key[mid] >= (kk[u] & ~4095u);
				hi[u] = ge ? mid : hi[u];
				lo[u] = ge ? lo[u] : mid + 1;
			}
		}
		uint32_t cell[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) cell[u] = vin[kk[u] & 4095u];
		asm volatile("" : "+v"(cell[0]), "+v"(cell[1]), "+v"(cell[2]), "+v"(cell[3]));
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int i = t + u * kSortThreads;
			if (i < m) {
				kout[i] = (kk[u] & ~4095u) | (uint32_t)(i - lo[u]);
				vout[i] = cell[u];
				atomicAdd(&s_hist[kk[u] >> 12], 1u);
			}
		}
	}
	__syncthreads();
	// the chunk's histogram, whole (u16: a chunk has 4096 lines)
	uint4* __restrict__ hout = reinterpret_cast<uint4*>(q.chunkHist + ((size_t)frame * q.chunks + chunk) * kBins);
	{
		const uint32_t* h = s_hist + 8 * t;
		hout[t] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
	}
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void sht_strength_scan_kernel(ShtArgs a, ShtSortArgs q)
{
	__shared__ uint32_t s_wave[kSortThreads / 64];
	const int frame = blockIdx.x, t = threadIdx.x;
	const size_t nf = min((size_t)max(a.lineCounts[frame], 0), a.lineCap);
	if (nf == 0) return;
	const int nch = (int)((nf + kChunk - 1) / kChunk);
	// lines per inverted strength 8 t .. 8 t + 7 over the frame's chunks
	uint32_t tot[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	const uint4* __restrict__ h = reinterpret_cast<const uint4*>(q.chunkHist + (size_t)frame * q.chunks * kBins) + t;
	auto add8 = [](uint32_t (&d)[8], const uint4& v) {
		d[0] += v.x & 0xffffu; d[1] += v.x >> 16; d[2] += v.y & 0xffffu; d[3] += v.y >> 16;
		d[4] += v.z & 0xffffu; d[5] += v.z >> 16; d[6] += v.w & 0xffffu; d[7] += v.w >> 16;
	};
	const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
	for (int c0 = 0; c0 < nch; c0 += 4) {   // four chunks' histograms in flight (clamped, unconditional loads)
		uint4 v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) v[u] = h[(size_t)min(c0 + u, nch - 1) * (kBins / 8)];
#pragma unroll
		for (int u = 0; u < 4; ++u) add8(tot, (c0 + u < nch) ? v[u] : zero4);
	}
	uint32_t sum = 0;
#pragma unroll
	for (int u = 0; u < 8; ++u) { const uint32_t c = tot[u]; tot[u] = sum; sum += c; }   // exclusive inside the thread
	// exclusive scan of the thread sums over the block (ascending inverted strength = descending strength)
	uint32_t incl = sum;
	const int lane = t & 63, wave = t >> 6;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t n = __shfl_up(incl, o);
		if (lane >= o) incl += n;
	}
	if (lane == 63) s_wave[wave] = incl;
	__syncthreads();
	uint32_t before = 0;
#pragma unroll
	for (int w = 0; w < kSortThreads / 64; ++w) before += (w < wave) ? s_wave[w] : 0u;
	const uint32_t excl = before + incl - sum;
	uint4* __restrict__ out = reinterpret_cast<uint4*>(q.strengthStart + (size_t)frame * kBins) + 2 * t;
	out[0] = make_uint4(excl + tot[0], excl + tot[1], excl + tot[2], excl + tot[3]);
	out[1] = make_uint4(excl + tot[4], excl + tot[5], excl + tot[6], excl + tot[7]);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sht_place_lines_kernel(ShtArgs a, ShtSortArgs q, float thetaStep, int maxLines, LineOut* __restrict__ lines, size_t outCap)
{
	__shared__ unsigned long long s_part[4];
	const int frame = blockIdx.y;
	const size_t nf = min((size_t)max(a.lineCounts[frame], 0), a.lineCap);
	const size_t i0 = (size_t)blockIdx.x * 256;
	if (i0 >= nf) return;   // uniform
	const size_t base = frame_base<256>(a.lineCounts, frame, a.lineCap, s_part);
	const size_t i = i0 + threadIdx.x;
	if (i >= nf) return;
	size_t limit = nf;      // slots of the frame that are written: min(lines, lineCap, maxLines, outCap)
	if (maxLines > 0 && limit > (size_t)maxLines) limit = (size_t)maxLines;
	if (limit > outCap) limit = outCap;
	const uint32_t kv = q.sortedKeys[base + i];
	const uint32_t inv = kv >> 12, rank = kv & 4095u;
	const int c = (int)(i >> 12);
	// slot = first slot of the strength in the frame + its lines in the frame's earlier chunks + rank inside the chunk.  (Per-chunk starts written by the scan
	// kernel instead: 15 MB more traffic from 32 workgroups, 20 + 16 us against 8 + 18 for this loop at 4K.)  Four histograms in flight, clamped loads.
	size_t pos = (size_t)q.strengthStart[(size_t)frame * kBins + inv] + rank;
	const uint16_t* __restrict__ h = q.chunkHist + (size_t)frame * q.chunks * kBins + inv;
	for (int c0 = 0; c0 < c; c0 += 4) {
		uint32_t v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) v[u] = h[(size_t)min(c0 + u, c) * kBins];   // chunk c itself is a valid address (its histogram exists): masked below
#pragma unroll
		for (int u = 0; u < 4; ++u) pos += (c0 + u < c) ? v[u] : 0u;
	}
	if (pos >= limit) return;
	const uint32_t cell = q.sortedVals[base + i];
	const int row = (int)(cell / (uint32_t)a.T), col = (int)(cell - (uint32_t)row * (uint32_t)a.T);
	LineOut o;
	o.rho = (float)(a.barrier - row);            // static_cast<float>(barrier - row), houghsht.cxx:661
	o.theta = __fmul_rn((float)col, thetaStep);  // col * theta (f32), houghsht.cxx:662
	o.strength = (int32_t)(((1u << a.strengthBits) - 1u) - inv);
	o.row = row; o.col = col;
	lines[(size_t)frame * outCap + pos] = o;
}

// ---------------------------------------------------------------------------------------------------------------
hipError_t launch_sht_sort_lines(const ShtArgs& a, const ShtSortArgs& q, int frames, float thetaStep, int maxLines, void* lines, size_t outCap, hipStream_t stream)
{
	if (!lines || !outCap) return hipSuccess;
	hipLaunchKernelGGL(sht_chunk_sort_kernel, dim3((unsigned)q.chunks, (unsigned)frames), dim3(kSortThreads), 0, stream, a, q);
	hipLaunchKernelGGL(sht_strength_scan_kernel, dim3((unsigned)frames), dim3(kSortThreads), 0, stream, a, q);
	hipLaunchKernelGGL(sht_place_lines_kernel, dim3((unsigned)(q.chunks * (kChunk / 256)), (unsigned)frames), dim3(256), 0, stream, a, q, thetaStep, maxLines,
	                   reinterpret_cast<LineOut*>(lines), outCap);
	return hipGetLastError();
}

} // namespace compvhip
