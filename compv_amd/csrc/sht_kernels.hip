// sht_kernels.hip -- standard Hough transform (SHT) line accumulation for gfx950, hand-written HIP.
//
// Replaces, behind compvhip_houghsht_u8 / compvhip_plan_houghsht (include/compv_hip.h):
//   CompVHoughSht::process                  core/features/hough/compv_core_feature_houghsht.cxx:96-262
//   acc_gather + AccGatherRow/RowTimesSinRho ...houghsht.cxx:350-481,607-627 (+ intrin_avx2.cxx:41-98)
//   nms_gather / nms_apply                  ...houghsht.cxx:483-564,629-668 (+ intrin_sse2.cxx:16-101)
//
// Data layout in HBM (per frame):
//   edge bit mask   u32 [H][wb]            produced by the Canny kernels (or by bytes_to_bits for foreign edge maps)
//   edge list       u32 [E]   (y<<16)|x    compacted with wave ballot/popcount prefix sums
//   accumulator     i32 [T][accPitch]      THETA-major (the reference is rho-major [R][192]); each vote workgroup
//                                          owns kShtThetaPerGroup whole theta columns, so the accumulator is
//                                          written exactly once, coalesced, with no global atomics
//   line keys       u64 [lines]            strength<<32 | ~(row*T+col): unique keys, sorted descending on device
//
// Voting: rho = (x*cosQ[t] + y*sinQ[t]) >> 16 (int32, arithmetic shift), acc[barrier - rho][t]++ for every edge and
// every t -- E*T scattered increments, the whole cost of the reference's SHT.  Each workgroup privatises the (rho)
// histogram of 4 theta bins in LDS (two u16 counters per dword: a cell can never exceed the number of pixels in a
// 1-px-wide band < 65536 for W,H <= 32767) and votes with ds_add_u32.  Lanes walk far-apart contiguous chunks of the
// raster-ordered edge list so that a wave's 64 simultaneous votes land on different image rows: raster neighbours
// share rho around theta = 90 deg and would otherwise serialise on one LDS address.
#include "kernels.hpp"

#include <cstring>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace compvhip {

// ---------------------------------------------------------------------------------------------------------------
// foreign edge maps: bytes -> bit mask (any non-zero byte is an edge: houghsht.cxx:159-165)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bytes_to_bits_kernel(const uint8_t* __restrict__ edges, int W, int H, int S, size_t frameStride,
                                                            uint32_t* __restrict__ ebits, int wb, size_t bitsFrameStride)
{
	const int frame = blockIdx.z;
	const int y = blockIdx.y;
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= wb) return;
	const uint8_t* row = edges + (size_t)frame * frameStride + (size_t)y * S;
	const int x = k * 32;
	uint32_t bits = 0;
	if (x + 32 <= W && ((S & 15) == 0)) {
		const uint4 a = reinterpret_cast<const uint4*>(row + x)[0];
		const uint4 b = reinterpret_cast<const uint4*>(row + x)[1];
		const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
		for (int i = 0; i < 8; ++i) {
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if ((w[i] >> (8 * j)) & 0xffu) bits |= 1u << (4 * i + j);
			}
		}
	}
	else {
		for (int i = 0; i < 32; ++i) {
			if (x + i < W && row[x + i]) bits |= 1u << i;
		}
	}
	ebits[(size_t)frame * bitsFrameStride + (size_t)y * wb + k] = bits;
}

// ---------------------------------------------------------------------------------------------------------------
// bit mask -> edge list.  One thread per 32-px word; wave prefix sum of popcounts, one global atomic per wave.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sht_compact_kernel(ShtArgs a)
{
	const int frame = blockIdx.y;
	const size_t nwords = (size_t)a.H * a.wb;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t bits = 0;
	if (i < nwords) bits = a.ebits[(size_t)frame * a.bitsFrameStride + i];
	const int cnt = __popc(bits);
	// inclusive wave scan
	int incl = cnt;
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int n = __shfl_up(incl, o);
		if (lane >= o) incl += n;
	}
	const int total = __shfl(incl, 63);
	if (total == 0) return; // wave-uniform
	int base = 0;
	if (lane == 63) base = atomicAdd(&a.edgeCounts[frame], total);
	base = __shfl(base, 63);
	size_t pos = (size_t)base + (incl - cnt);
	if (cnt) {
		const int y = (int)(i / a.wb);
		const int x0 = (int)(i - (size_t)y * a.wb) * 32;
		uint32_t* __restrict__ dst = a.edges + (size_t)frame * a.edgeCap;
		while (bits) {
			const int b = __ffs(bits) - 1;
			bits &= bits - 1;
			if (pos < a.edgeCap) dst[pos] = ((uint32_t)y << 16) | (uint32_t)(x0 + b);
			++pos;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// voting
// ---------------------------------------------------------------------------------------------------------------
size_t sht_vote_lds_bytes(int R) { return (size_t)R * 2 * sizeof(uint32_t); }

__global__ __launch_bounds__(kShtVoteThreads) void sht_vote_kernel(ShtArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t hist[]; // [R][2]: (t0,t0+1) and (t0+2,t0+3) u16 pairs
	const int frame = blockIdx.z;
	const int shard = blockIdx.y;
	const int t0 = blockIdx.x * kShtThetaPerGroup;
	const int tid = threadIdx.x;
	const int R = a.R;

	for (int i = tid; i < 2 * R; i += kShtVoteThreads) hist[i] = 0u;

	int cq[kShtThetaPerGroup], sq[kShtThetaPerGroup];
#pragma unroll
	for (int k = 0; k < kShtThetaPerGroup; ++k) {
		const int t = min(t0 + k, a.T - 1);
		cq[k] = a.cosQ[t];
		sq[k] = a.sinQ[t];
	}
	const int nvalid = min(kShtThetaPerGroup, a.T - t0);

	const int n = min(a.edgeCounts[frame], (int)a.edgeCap);
	const int sBeg = (int)(((long long)n * shard) / a.shards);
	const int sEnd = (int)(((long long)n * (shard + 1)) / a.shards);
	const int cnt = sEnd - sBeg;
	const int chunk = (cnt + kShtVoteThreads - 1) / kShtVoteThreads;
	const uint32_t* __restrict__ edges = a.edges + (size_t)frame * a.edgeCap + sBeg;
	const int eBeg = tid * chunk;
	const int eEnd = min(eBeg + chunk, cnt);
	__syncthreads();

	const int barrier = a.barrier;
	for (int e = eBeg; e < eEnd; ++e) {
		const uint32_t xy = edges[e];
		const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
#pragma unroll
		for (int k = 0; k < kShtThetaPerGroup; ++k) {
			if (k < nvalid) {
				const int rho = (__mul24(x, cq[k]) + __mul24(y, sq[k])) >> 16;
				const int idx = barrier - rho;
				atomicAdd(&hist[idx * 2 + (k >> 1)], (k & 1) ? 0x10000u : 1u);
			}
		}
	}
	__syncthreads();

	int32_t* __restrict__ acc = a.acc + (size_t)frame * a.accFrameStride + (size_t)t0 * a.accPitch;
	for (int r = tid; r < R; r += kShtVoteThreads) {
		const uint32_t v0 = hist[2 * r], v1 = hist[2 * r + 1];
		const int c[4] = { (int)(v0 & 0xffffu), (int)(v0 >> 16), (int)(v1 & 0xffffu), (int)(v1 >> 16) };
#pragma unroll
		for (int k = 0; k < kShtThetaPerGroup; ++k) {
			if (k < nvalid) {
				if (a.shards == 1) acc[(size_t)k * a.accPitch + r] = c[k];
				else if (c[k]) atomicAdd(&acc[(size_t)k * a.accPitch + r], c[k]);
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// NMS + threshold -> line keys
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sht_nms_kernel(ShtArgs a)
{
	const int frame = blockIdx.z;
	const int c = blockIdx.y;
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= a.R) return;
	const int32_t* __restrict__ acc = a.acc + (size_t)frame * a.accFrameStride;
	const size_t p = a.accPitch;
	const int v = acc[(size_t)c * p + r];
	if (v <= a.threshold) return;
	if (r >= 1 && r <= a.R - 2 && c >= 1 && c <= a.nmsLastCol) {
		const int32_t* l = acc + (size_t)(c - 1) * p + r;
		const int32_t* m = acc + (size_t)c * p + r;
		const int32_t* h = acc + (size_t)(c + 1) * p + r;
		if (l[-1] > v || l[0] > v || l[1] > v || m[-1] > v || m[1] > v || h[-1] > v || h[0] > v || h[1] > v) return;
	}
	const int idx = atomicAdd(&a.lineCounts[frame], 1);
	if ((size_t)idx < a.lineCap) {
		const uint32_t cell = (uint32_t)r * (uint32_t)a.T + (uint32_t)c;
		a.lineKeys[(size_t)frame * a.lineCap + idx] = ((uint64_t)(uint32_t)v << 32) | (uint64_t)(0xffffffffu - cell);
	}
}

__global__ void sht_segments_kernel(const int* __restrict__ counts, size_t lineCap, int frames, unsigned int* __restrict__ beg, unsigned int* __restrict__ end)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= frames) return;
	const size_t c = (size_t)max(counts[f], 0);
	beg[f] = (unsigned int)(f * lineCap);
	end[f] = (unsigned int)(f * lineCap + (c < lineCap ? c : lineCap));
}

struct LineOut { float rho; float theta; int32_t strength; int32_t row; int32_t col; };

__global__ __launch_bounds__(256) void sht_decode_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ counts, size_t lineCap, int T, int barrier,
                                                         float thetaStep, int maxLines, LineOut* __restrict__ lines, size_t outCap)
{
	const int frame = blockIdx.y;
	size_t n = (size_t)max(counts[frame], 0);
	if (n > lineCap) n = lineCap;
	if (maxLines > 0 && n > (size_t)maxLines) n = (size_t)maxLines;
	if (n > outCap) n = outCap;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t k = keys[(size_t)frame * lineCap + i];
	const uint32_t cell = 0xffffffffu - (uint32_t)k;
	const int row = (int)(cell / (uint32_t)T), col = (int)(cell - (uint32_t)row * (uint32_t)T);
	LineOut o;
	o.rho = (float)(barrier - row);              // static_cast<float>(barrier - row), houghsht.cxx:661
	o.theta = __fmul_rn((float)col, thetaStep);  // col * theta (f32), houghsht.cxx:662
	o.strength = (int32_t)(k >> 32);
	o.row = row; o.col = col;
	lines[(size_t)frame * outCap + i] = o;
}

__global__ __launch_bounds__(256) void sht_acc_transpose_kernel(const int32_t* __restrict__ accT, int R, int T, int accPitch, int32_t* __restrict__ out, size_t outStride)
{
	__shared__ int32_t tile[32][33];
	const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
	for (int j = ty; j < 32; j += 8) {
		const int c = c0 + j, r = r0 + tx;
		tile[j][tx] = (c < T && r < R) ? accT[(size_t)c * accPitch + r] : 0;
	}
	__syncthreads();
	for (int j = ty; j < 32; j += 8) {
		const int r = r0 + j, c = c0 + tx;
		if (r < R && c < T) out[(size_t)r * outStride + c] = tile[tx][j];
	}
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
hipError_t launch_bytes_to_bits(const uint8_t* edges, int W, int H, int S, size_t frameStride, uint32_t* ebits, int wb, size_t bitsFrameStride,
                                int frames, hipStream_t stream)
{
	dim3 grid((wb + 255) / 256, H, frames);
	hipLaunchKernelGGL(bytes_to_bits_kernel, grid, dim3(256), 0, stream, edges, W, H, S, frameStride, ebits, wb, bitsFrameStride);
	return hipGetLastError();
}

hipError_t launch_sht_compact(const ShtArgs& a, int frames, hipStream_t stream)
{
	hipError_t e = hipMemsetAsync(a.edgeCounts, 0, sizeof(int) * frames, stream);
	if (e != hipSuccess) return e;
	const size_t nwords = (size_t)a.H * a.wb;
	dim3 grid((unsigned)((nwords + 255) / 256), frames);
	hipLaunchKernelGGL(sht_compact_kernel, grid, dim3(256), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_sht_vote(const ShtArgs& a, int frames, hipStream_t stream)
{
	static size_t attr_lds = 0;
	const size_t lds = sht_vote_lds_bytes(a.R);
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	if (lds > attr_lds) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sht_vote_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
		attr_lds = lds;
	}
	if (a.shards > 1) {
		hipError_t e = hipMemsetAsync(a.acc, 0, sizeof(int32_t) * a.accFrameStride * frames, stream);
		if (e != hipSuccess) return e;
	}
	const int groups = (a.T + kShtThetaPerGroup - 1) / kShtThetaPerGroup;
	dim3 grid(groups, a.shards, frames);
	hipLaunchKernelGGL(sht_vote_kernel, grid, dim3(kShtVoteThreads), lds, stream, a);
	return hipGetLastError();
}

hipError_t launch_sht_nms(const ShtArgs& a, int frames, hipStream_t stream)
{
	hipError_t e = hipMemsetAsync(a.lineCounts, 0, sizeof(int) * frames, stream);
	if (e != hipSuccess) return e;
	dim3 grid((a.R + 255) / 256, a.T, frames);
	hipLaunchKernelGGL(sht_nms_kernel, grid, dim3(256), 0, stream, a);
	return hipGetLastError();
}

// descending segmented sort of the unique 64-bit keys (one segment per frame) -- rocPRIM device primitive
hipError_t sht_sort_keys(void* temp, size_t& tempBytes, const uint64_t* keysIn, uint64_t* keysOut, size_t lineCap, int frames,
                         const int* counts, unsigned int* segBeg, unsigned int* segEnd, hipStream_t stream)
{
	if (temp) {
		hipLaunchKernelGGL(sht_segments_kernel, dim3((frames + 63) / 64), dim3(64), 0, stream, counts, lineCap, frames, segBeg, segEnd);
	}
	return rocprim::segmented_radix_sort_keys_desc(temp, tempBytes, keysIn, keysOut, (unsigned int)(lineCap * frames), (unsigned int)frames,
	                                               segBeg, segEnd, 0, 64, stream);
}

hipError_t launch_sht_decode(const uint64_t* keys, const int* counts, size_t lineCap, int frames, int T, int barrier, float thetaStep,
                             int maxLines, void* lines, size_t outCap, hipStream_t stream)
{
	size_t n = lineCap < outCap ? lineCap : outCap;
	if (maxLines > 0 && (size_t)maxLines < n) n = (size_t)maxLines;
	if (n == 0) return hipSuccess;
	dim3 grid((unsigned)((n + 255) / 256), frames);
	hipLaunchKernelGGL(sht_decode_kernel, grid, dim3(256), 0, stream, keys, counts, lineCap, T, barrier, thetaStep, maxLines,
	                   reinterpret_cast<LineOut*>(lines), outCap);
	return hipGetLastError();
}

hipError_t launch_sht_acc_transpose(const int32_t* accT, int R, int T, int accPitch, int32_t* out, size_t outStride, hipStream_t stream)
{
	dim3 grid((R + 31) / 32, (T + 31) / 32);
	hipLaunchKernelGGL(sht_acc_transpose_kernel, grid, dim3(256), 0, stream, accT, R, T, accPitch, out, outStride);
	return hipGetLastError();
}

} // namespace compvhip
