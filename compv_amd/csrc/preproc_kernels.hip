// preproc_kernels.hip -- the samples' caller-side pre-processing, hand-written HIP for gfx950.
//
// Replaces, behind compvhip_grayscale_u8 / compvhip_plan_grayscale and compvhip_otsu_u8 / compvhip_plan_otsu:
//   CompVImage::convertGrayscale   base/image/compv_image_conv_to_grayscale.cxx:35-282 with the leaves
//       rgb24family_to_y / rgb32family_to_y / rgb565family_to_y   base/image/compv_image_conv_rgbfamily.cxx:93-117,243-268,403-436
//       (Y = ((33 R + 65 G + 13 B) >> 7) + 16, coefficient tables base/image/compv_image_conv_common.cxx:29-135)
//       yuyv422_to_y / uyvy422_to_y                                base/image/compv_image_conv_to_grayscale.cxx:233-282
//   CompVImage::thresholdOtsu      base/image/compv_image_threshold.cxx:52-114 (CompVMathHistogram::build + the f32 scan)
// as called by samples/hough_lines/main.cxx:102-105.
//
// Both are pure streaming kernels (HBM-bound): grayscale reads bpp B/px and writes 1 B/px; the histogram reads 1 B/px.
#include "kernels.hpp"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace compvhip {

// ---------------------------------------------------------------------------------------------------------------
// packed pixel formats -> luma.  One thread = 8 adjacent pixels = one 8-byte store; rows are S samples long with
// S % 8 == 0, so a group never leaves its row (columns >= W are padding the reference's SIMD leaves also write).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t luma(int c0, int c1, int c2, uint32_t a, uint32_t b, uint32_t c)
{
	return ((uint32_t)(c0 * (int)a + c1 * (int)b + c2 * (int)c) >> 7) + 16u; // <= 237: clampPixel8 never clamps
}

template <int FMT>
__device__ __forceinline__ uint32_t luma565(uint32_t k)
{
	constexpr bool be = (FMT == COMPVHIP_FMT_RGB565BE || FMT == COMPVHIP_FMT_BGR565BE);
	constexpr bool bgr = (FMT == COMPVHIP_FMT_BGR565LE || FMT == COMPVHIP_FMT_BGR565BE);
	if (be) k = ((k << 8) | (k >> 8)) & 0xffffu;
	uint32_t r = (k & 0xF800u) >> 8; r |= r >> 5;
	uint32_t g = (k & 0x07E0u) >> 3; g |= g >> 6;
	uint32_t b = (k & 0x001Fu) << 3; b |= b >> 5;
	return bgr ? luma(13, 65, 33, r, g, b) : luma(33, 65, 13, r, g, b);
}

template <int FMT>
__global__ __launch_bounds__(256) void gray_kernel(GrayArgs a)
{
	const int g8 = blockIdx.x * blockDim.x + threadIdx.x; // group of 8 pixels
	const int y = blockIdx.y, frame = blockIdx.z;
	if (g8 * 8 >= a.W) return;
	constexpr int bpp = (FMT <= COMPVHIP_FMT_BGRA32) ? 4 : (FMT <= COMPVHIP_FMT_BGR24 ? 3 : (FMT == COMPVHIP_FMT_Y ? 1 : 2));
	const uint8_t* __restrict__ src = a.in + ((size_t)frame * a.H + y) * (size_t)a.S * bpp + (size_t)g8 * 8 * bpp;
	uint32_t yv[8];
	if constexpr (bpp == 4) {
		const uint4 q0 = reinterpret_cast<const uint4*>(src)[0], q1 = reinterpret_cast<const uint4*>(src)[1];
		const uint32_t px[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint32_t b0 = px[i] & 0xffu, b1 = (px[i] >> 8) & 0xffu, b2 = (px[i] >> 16) & 0xffu, b3 = px[i] >> 24;
			if (FMT == COMPVHIP_FMT_RGBA32) yv[i] = luma(33, 65, 13, b0, b1, b2);
			else if (FMT == COMPVHIP_FMT_ARGB32) yv[i] = luma(33, 65, 13, b1, b2, b3);
			else yv[i] = luma(13, 65, 33, b0, b1, b2); // BGRA
		}
	}
	else if constexpr (bpp == 3) {
		const uint2 q0 = reinterpret_cast<const uint2*>(src)[0], q1 = reinterpret_cast<const uint2*>(src)[1], q2 = reinterpret_cast<const uint2*>(src)[2];
		const uint32_t w[6] = { q0.x, q0.y, q1.x, q1.y, q2.x, q2.y };
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			uint32_t c[3];
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				const int byte = 3 * i + k;
				c[k] = (w[byte >> 2] >> (8 * (byte & 3))) & 0xffu;
			}
			yv[i] = (FMT == COMPVHIP_FMT_RGB24) ? luma(33, 65, 13, c[0], c[1], c[2]) : luma(13, 65, 33, c[0], c[1], c[2]);
		}
	}
	else if constexpr (bpp == 2) {
		const uint4 q = reinterpret_cast<const uint4*>(src)[0];
		const uint32_t w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint32_t k = (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
			if (FMT == COMPVHIP_FMT_YUYV422) yv[i] = k & 0xffu;
			else if (FMT == COMPVHIP_FMT_UYVY422) yv[i] = k >> 8;
			else yv[i] = luma565<FMT>(k);
		}
	}
	else {
		const uint2 q = reinterpret_cast<const uint2*>(src)[0];
#pragma unroll
		for (int i = 0; i < 8; ++i) yv[i] = ((i < 4 ? q.x : q.y) >> (8 * (i & 3))) & 0xffu;
	}
	uint2 o;
	o.x = yv[0] | (yv[1] << 8) | (yv[2] << 16) | (yv[3] << 24);
	o.y = yv[4] | (yv[5] << 8) | (yv[6] << 16) | (yv[7] << 24);
	*reinterpret_cast<uint2*>(a.out + ((size_t)frame * a.H + y) * (size_t)a.So + (size_t)g8 * 8) = o;
}

hipError_t launch_gray(const GrayArgs& a, int fmt, int frames, hipStream_t stream)
{
	const int groups = (a.W + 7) / 8;
	dim3 grid((groups + 255) / 256, a.H, frames), block(256);
	switch (fmt) {
#define COMPV_GRAY_CASE(F) case F: hipLaunchKernelGGL((gray_kernel<F>), grid, block, 0, stream, a); break
	COMPV_GRAY_CASE(COMPVHIP_FMT_RGBA32); COMPV_GRAY_CASE(COMPVHIP_FMT_ARGB32); COMPV_GRAY_CASE(COMPVHIP_FMT_BGRA32);
	COMPV_GRAY_CASE(COMPVHIP_FMT_RGB24); COMPV_GRAY_CASE(COMPVHIP_FMT_BGR24);
	COMPV_GRAY_CASE(COMPVHIP_FMT_RGB565LE); COMPV_GRAY_CASE(COMPVHIP_FMT_RGB565BE); COMPV_GRAY_CASE(COMPVHIP_FMT_BGR565LE); COMPV_GRAY_CASE(COMPVHIP_FMT_BGR565BE);
	COMPV_GRAY_CASE(COMPVHIP_FMT_YUYV422); COMPV_GRAY_CASE(COMPVHIP_FMT_UYVY422); COMPV_GRAY_CASE(COMPVHIP_FMT_Y);
#undef COMPV_GRAY_CASE
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// 256-bin histogram (CompVMathHistogram::build on the `cols` bytes of every row).
// Natural images put most pixels in a few bins, and same-address LDS atomics serialise (2 cycles per lane, see
// tools/microbench): the workgroup's histogram is therefore replicated 32 ways, [256 bins][32 columns] u32 = 32 KB, and a
// lane always uses column lane % 32.  The hardware resolves a wave64 ds_add as two half-waves, so within one pass every
// lane hits its own bank: conflict-free whatever the image.  Columns are summed at the end (rotated reads).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kHistThreads = 256;

__global__ __launch_bounds__(kHistThreads) void hist256_kernel(const uint8_t* __restrict__ in, int W, int H, int S, size_t frameStride,
                                                              uint32_t* __restrict__ hist)
{
	__shared__ uint32_t s_hist[256 * 32];
	const int tid = threadIdx.x, frame = blockIdx.y;
	for (int i = tid; i < 256 * 32; i += kHistThreads) s_hist[i] = 0u;
	__syncthreads();
	const int rows = (H + (int)gridDim.x - 1) / (int)gridDim.x; // gridDim.x row chunks per frame
	const int r0 = blockIdx.x * rows, r1 = min(H, r0 + rows);
	const int groups = (W + 7) >> 3;              // 8-byte groups per row (the last one may be partial)
	const int fullGroups = W >> 3;
	const uint8_t* __restrict__ base = in + (size_t)frame * frameStride;
	uint32_t* col = s_hist + (tid & 31);
	const int lane = tid & 63, wave = tid >> 6;
	auto vote8 = [&](uint2 q) {
#pragma unroll
		for (int k = 0; k < 8; ++k) atomicAdd(col + (((k < 4 ? q.x : q.y) >> (8 * (k & 3))) & 0xffu) * 32, 1u);
	};
	// one wave per row, lanes stride over the row's 8-byte groups; no per-byte bounds tests
	for (int r = r0 + wave; r < r1; r += kHistThreads / 64) {
		const uint2* __restrict__ row = reinterpret_cast<const uint2*>(base + (size_t)r * S);
		int g = lane;
		for (; g + 192 < fullGroups; g += 256) { // four 8-byte loads in flight per lane: the kernel is latency-bound otherwise
			const uint2 q0 = row[g], q1 = row[g + 64], q2 = row[g + 128], q3 = row[g + 192];
			vote8(q0); vote8(q1); vote8(q2); vote8(q3);
		}
		{
			const bool v0 = g < fullGroups, v1 = g + 64 < fullGroups, v2 = g + 128 < fullGroups;
			uint2 q0 = make_uint2(0, 0), q1 = q0, q2 = q0;
			if (v0) q0 = row[g];
			if (v1) q1 = row[g + 64];
			if (v2) q2 = row[g + 128];
			if (v0) vote8(q0);
			if (v1) vote8(q1);
			if (v2) vote8(q2);
		}
		if ((W & 7) && lane == 0) { // ragged tail of the row: W % 8 bytes of the last group
			const uint2 q = row[fullGroups];
			for (int k = 0; k < (W & 7); ++k) atomicAdd(col + (((k < 4 ? q.x : q.y) >> (8 * (k & 3))) & 0xffu) * 32, 1u);
		}
	}
	(void)groups;
	__syncthreads();
	// thread t sums bin t over the 32 columns, starting at column t so that the 64 lanes of a wave read 32 different banks
	uint32_t sum = 0;
#pragma unroll 8
	for (int j = 0; j < 32; ++j) sum += s_hist[tid * 32 + ((j + tid) & 31)];
	hist[((size_t)frame * gridDim.x + blockIdx.x) * 256 + tid] = sum; // per-chunk partial histogram: no global atomics, no memset
}

// CompVImageThreshold::otsu scan (compv_image_threshold.cxx:83-105): u32 sums, f32 arithmetic in source order with single
// correctly-rounded operations (no FMA contraction); then the sample's Canny thresholds (samples/hough_lines/main.cxx:104-105).
// One workgroup of 256 threads per frame: thread 0 walks the two running sums (q1: int, sumB: an f32 accumulation whose
// rounding depends on the order, so it stays serial -- 256 dependent adds), every thread then evaluates "its" level's
// between-class variance (the two divisions), and the first level reaching the maximum wins (the reference updates on '>').
__global__ __launch_bounds__(256) void otsu_kernel(const uint32_t* __restrict__ hist, int chunks, int N, float fLowFactor, float fHighFactor, int32_t* __restrict__ otsu,
                                                   int2* __restrict__ thr)
{
	__shared__ int s_q1[256];
	__shared__ float s_sumB[256];
	__shared__ uint32_t s_sumA[256];
	__shared__ float s_sumf;
	__shared__ int s_last; // first level at which q2 == 0 (the reference breaks there), 256 if none
	__shared__ float s_var[256];
	__shared__ int s_idx[256];
	const int f = blockIdx.x, i = threadIdx.x;
	uint32_t hi_ = 0; // CompVMathHistogram::build: sum of the row-chunk partial histograms
	{
		const uint32_t* hp = hist + (size_t)f * chunks * 256 + i;
		int c = 0;
		for (; c + 8 <= chunks; c += 8) { // eight independent loads in flight (a plain loop serialises on the load latency)
			uint32_t v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = hp[(size_t)(c + u) * 256];
#pragma unroll
			for (int u = 0; u < 8; ++u) hi_ += v[u];
		}
		for (; c < chunks; ++c) hi_ += hp[(size_t)c * 256];
	}
	s_sumA[i] = (uint32_t)i * hi_;
	s_q1[i] = (int)hi_;
	__syncthreads();
	if (i == 0) {
		uint32_t sum32 = 0;
		int q1 = 0, last = 256;
		float sumB = 0.f;
		for (int k0 = 0; k0 < 256; k0 += 16) { // 16 levels per trip: the LDS reads are issued together, only the adds are serial
			uint32_t sa[16]; int hq[16], oq[16]; float ob[16];
#pragma unroll
			for (int j = 0; j < 16; ++j) { sa[j] = s_sumA[k0 + j]; hq[j] = s_q1[k0 + j]; }
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				sum32 += sa[j];
				q1 += hq[j];
				if (q1) {
					if (N - q1 == 0) { if (last == 256) last = k0 + j; }
					else if (last == 256) sumB = __fadd_rn(sumB, (float)sa[j]);
				}
				oq[j] = q1; ob[j] = sumB;
			}
#pragma unroll
			for (int j = 0; j < 16; ++j) { s_q1[k0 + j] = oq[j]; s_sumB[k0 + j] = ob[j]; }
		}
		s_sumf = (float)sum32;
		s_last = last;
	}
	__syncthreads();
	float varB = -1.f; // levels that the reference never evaluates lose against varMax = 0
	const int q1 = s_q1[i];
	if (q1 && i < s_last) {
		const float q1f = (float)q1, q2f = (float)(N - q1), sumB = s_sumB[i];
		const float mf = __fsub_rn(__fdiv_rn(sumB, q1f), __fdiv_rn(__fsub_rn(s_sumf, sumB), q2f));
		varB = __fmul_rn(__fmul_rn(__fmul_rn(q1f, q2f), mf), mf);
	}
	// varB > varMax with varMax starting at 0: only strictly positive values can win; NaN never does
	s_var[i] = (varB > 0.f) ? varB : 0.f;
	s_idx[i] = (varB > 0.f) ? i : 0;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if (i < o) {
			const float a = s_var[i], b = s_var[i + o];
			const int ia = s_idx[i], ib = s_idx[i + o];
			if (b > a || (b == a && b > 0.f && ib < ia)) { s_var[i] = b; s_idx[i] = ib; }
		}
		__syncthreads();
	}
	if (i != 0) return;
	const int t = s_idx[0];
	if (otsu) otsu[f] = t;
	if (thr) {
		// LOW = (float)(t * (double)factor), HIGH likewise; COMPARE_TO_GRADIENT clamp (canny_dete.cxx:251-266); degenerate -> (1,3)
		const float fLow = (float)((double)t * (double)fLowFactor), fHigh = (float)((double)t * (double)fHighFactor);
		int lo = 1, hi = 3;
		if (fLow > 0.f && fHigh > 0.f && fLow < fHigh) {
			const float l = fLow < 1.f ? 1.f : (fLow > 65535.f ? 65535.f : fLow);
			const float hh = fHigh < 1.f ? 1.f : (fHigh > 65535.f ? 65535.f : fHigh);
			lo = (int)l; hi = (int)hh;
			lo = lo < 1 ? 1 : lo;
			hi = (lo + 2 > hi) ? lo + 2 : hi;
		}
		thr[f] = make_int2(lo, hi);
	}
}

int otsu_hist_chunks(int H, int frames)
{
	int chunks = 1024 / (frames > 0 ? frames : 1); // ~4 workgroups per CU in flight, at least 16 rows each
	chunks = chunks < 16 ? 16 : (chunks > kOtsuMaxChunks ? kOtsuMaxChunks : chunks);
	if (chunks > (H + 15) / 16) chunks = (H + 15) / 16;
	return chunks;
}

hipError_t launch_otsu(const uint8_t* in, int W, int H, int S, size_t frameStride, int frames, float fLowFactor, float fHighFactor, uint32_t* hist,
                       int32_t* otsu, void* thr, hipStream_t stream)
{
	const int chunks = otsu_hist_chunks(H, frames);
	hipLaunchKernelGGL(hist256_kernel, dim3(chunks, frames), dim3(kHistThreads), 0, stream, in, W, H, S, frameStride, hist);
	hipLaunchKernelGGL(otsu_kernel, dim3(frames), dim3(256), 0, stream, hist, chunks, W * H, fLowFactor, fHighFactor, otsu, reinterpret_cast<int2*>(thr));
	return hipGetLastError();
}

} // namespace compvhip
