// convlt_kernels.hip -- separable fixed-point (Q16) convolution u8 -> u8, the optional Gaussian pre-blur of SURVEY.md 8f row 2.
//
// Replaces, behind compvhip_convlt1_fixedpoint_u8 / compvhip_plan_convlt1_fixedpoint:
//   CompVMathConvlt::convlt1FixedPoint       base/include/compv/base/math/compv_math_convlt.h:31-33 (entry), 98-173 (hz pass, then vt
//                                            pass through a temporary, zero OUTPUT border of kernSize/2), 386-405 (the leaf:
//                                            sum_k ((in[k] * kern[k]) >> 16), saturated to u8)
// with kernels from CompVMathGauss::kernelDim1FixedPoint (base/math/compv_math_gauss.cxx:11-17), computed on the HOST (libm exp).
//
// Data flow of the reference: horizontal pass, u8 rounding, vertical pass (the intermediate rounding is part of the result).
// Out of place: ONE fused kernel (no intermediate in memory, one v_mul_hi_u32_u24 per tap); in place: two streaming passes through
// a scratch plane.  VALU-bound (2K taps per pixel), not HBM-bound.
#include "kernels.hpp"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace compvhip {

__device__ __forceinline__ uint32_t byteOf(const uint32_t (&w)[6], int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; }

// horizontal pass: one thread = 8 adjacent pixels of one row; bytes x0-8 .. x0+15 come from three 8-byte loads whose addresses
// are clamped into the row (anything a clamp falsifies only feeds columns < r or >= W-r, which are forced to zero)
template <int K>
__global__ __launch_bounds__(256) void convlt_fxp_hz_kernel(FxpArgs a)
{
	constexpr int r = K / 2;
	const int x0 = (blockIdx.x * 256 + threadIdx.x) * 8, y = blockIdx.y, frame = blockIdx.z;
	if (x0 >= a.W) return;
	const uint8_t* __restrict__ row = a.in + (size_t)frame * a.inFrameStride + (size_t)y * a.S;
	const int xl = max(x0 - 8, 0), xr = min(x0 + 8, a.S - 8);
	const uint2 L = *reinterpret_cast<const uint2*>(row + xl), M = *reinterpret_cast<const uint2*>(row + x0), Rr = *reinterpret_cast<const uint2*>(row + xr);
	const uint32_t w[6] = { L.x, L.y, M.x, M.y, Rr.x, Rr.y };
	uint32_t o[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		uint32_t sum = 0;
#pragma unroll
		for (int t = 0; t < K; ++t) sum += __umul24(byteOf(w, 8 + j - r + t), a.kern[t]) >> 16;
		const int x = x0 + j;
		o[j] = (x < r || x >= a.W - r) ? 0u : min(sum, 255u);
	}
	uint2 q;
	q.x = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
	q.y = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
	*reinterpret_cast<uint2*>(a.out + (size_t)frame * a.outFrameStride + (size_t)y * a.So + x0) = q;
}

// vertical pass: one thread = 8 adjacent columns, marching down kFxpRows output rows with the last K input rows in registers
constexpr int kFxpRows = 64;

template <int K>
__global__ __launch_bounds__(256) void convlt_fxp_vt_kernel(FxpArgs a)
{
	constexpr int r = K / 2;
	const int x0 = (blockIdx.x * 256 + threadIdx.x) * 8, y0 = blockIdx.y * kFxpRows, frame = blockIdx.z;
	if (x0 >= a.W) return;
	const uint8_t* __restrict__ src = a.in + (size_t)frame * a.inFrameStride + x0;
	uint8_t* __restrict__ dst = a.out + (size_t)frame * a.outFrameStride + x0;
	uint2 ring[K]; // ring[t] = input row yo - r + t of the output row yo being produced
#pragma unroll
	for (int t = 0; t < K; ++t) ring[t] = make_uint2(0u, 0u);
	for (int i = 0; i < kFxpRows + K - 1; ++i) {
		const int yin = y0 - r + i;
#pragma unroll
		for (int t = 0; t + 1 < K; ++t) ring[t] = ring[t + 1];
		ring[K - 1] = *reinterpret_cast<const uint2*>(src + (size_t)min(max(yin, 0), a.H - 1) * a.S); // clamped rows only feed border outputs
		const int yo = yin - r;
		if (i < K - 1 || yo >= a.H) continue;
		uint2 q = make_uint2(0u, 0u);
		if (yo >= r && yo < a.H - r) {
			uint32_t o[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				uint32_t sum = 0;
#pragma unroll
				for (int t = 0; t < K; ++t) sum += __umul24(((j < 4 ? ring[t].x : ring[t].y) >> (8 * (j & 3))) & 0xffu, a.kern[t]) >> 16;
				o[j] = min(sum, 255u);
			}
			q.x = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
			q.y = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
		}
		*reinterpret_cast<uint2*>(dst + (size_t)yo * a.So) = q;
	}
}

// floor((b * k) / 65536) in ONE instruction: the pixel byte is kept pre-shifted in bits 8..15 of a 16-bit half (b << 8) and the Q16
// weight as k << 8; v_mul_hi_u32_u24 returns bits 47..32 of the 24x24-bit product (b << 8) * (k << 8) = b * k * 65536.
template <int HALF>
__device__ __forceinline__ uint32_t tapHi(uint32_t halves, uint32_t k8)
{
	uint32_t d;
	if (HALF == 0) asm("v_mul_hi_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(d) : "v"(halves), "v"(k8));
	else asm("v_mul_hi_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(d) : "v"(halves), "v"(k8));
	return d;
}
// bytes b0..b3 of a dword -> E = (b0 << 8) | (b2 << 24), O = (b1 << 8) | (b3 << 24)
__device__ __forceinline__ void splitShifted(uint32_t w, uint32_t& E, uint32_t& O) { E = (w << 8) & 0xff00ff00u; O = w & 0xff00ff00u; }

// Fused version (used when in and out do not alias): the thread filters each incoming row horizontally on the fly and pushes the
// u8 result into the vertical ring, so the intermediate never goes to memory: 1 B/px read (+ the 2r/64 row halo and the 8-byte
// column halos, which hit in L1/L2) and 1 B/px written, instead of 2 + 2.  Same arithmetic, same u8 rounding between the passes.
template <int K>
__global__ __launch_bounds__(256) void convlt_fxp_fused_kernel(FxpArgs a, FxpArgs v)
{
	constexpr int r = K / 2;
	const int x0 = (blockIdx.x * 256 + threadIdx.x) * 8, y0 = blockIdx.y * kFxpRows, frame = blockIdx.z;
	if (x0 >= a.W) return;
	const uint8_t* __restrict__ src = a.in + (size_t)frame * a.inFrameStride;
	uint8_t* __restrict__ dst = a.out + (size_t)frame * a.outFrameStride + x0;
	const int xl = max(x0 - 8, 0), xr = min(x0 + 8, a.S - 8);
	auto loadRow = [&](int yin, uint2& L, uint2& M, uint2& Rr) {
		const uint8_t* row = src + (size_t)min(max(yin, 0), a.H - 1) * a.S;
		L = *reinterpret_cast<const uint2*>(row + xl); M = *reinterpret_cast<const uint2*>(row + x0); Rr = *reinterpret_cast<const uint2*>(row + xr);
	};
	uint32_t colok = 0; // bit j: column x0+j is inside [r, W-r)
#pragma unroll
	for (int j = 0; j < 8; ++j) colok |= ((x0 + j >= r && x0 + j < a.W - r) ? 1u : 0u) << j;
	uint32_t hk[K], vk[K]; // Q16 weights << 8, in VGPRs (SDWA operands)
#pragma unroll
	for (int t = 0; t < K; ++t) { hk[t] = a.kern[t] << 8; vk[t] = v.kern[t] << 8; }
	// ring[t]: horizontally filtered row (8 pixels) in split pre-shifted form: [0] = px 0,2  [1] = px 1,3  [2] = px 4,6  [3] = px 5,7
	uint32_t ring[K][4];
#pragma unroll
	for (int t = 0; t < K; ++t) { ring[t][0] = ring[t][1] = ring[t][2] = ring[t][3] = 0u; }
	uint2 nL, nM, nR;
	loadRow(y0 - r, nL, nM, nR);
	for (int i = 0; i < kFxpRows + 2 * r; ++i) {
		const int yin = y0 - r + i; // input row filtered and pushed now; with K rows in the ring it completes output row yin - r
		// the 24 bytes x0-8 .. x0+15 as 12 registers of two pre-shifted bytes each: byte i -> sp[(i >> 2) * 2 + (i & 1)], half (i >> 1) & 1
		uint32_t sp[12];
		splitShifted(nL.x, sp[0], sp[1]); splitShifted(nL.y, sp[2], sp[3]); splitShifted(nM.x, sp[4], sp[5]);
		splitShifted(nM.y, sp[6], sp[7]); splitShifted(nR.x, sp[8], sp[9]); splitShifted(nR.y, sp[10], sp[11]);
		loadRow(yin + 1, nL, nM, nR); // prefetch: in flight while this row is filtered
		uint32_t o[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			uint32_t sum = 0;
#pragma unroll
			for (int t = 0; t < K; ++t) {
				const int bi = 8 + j - r + t; // byte index 0..23
				const uint32_t reg = sp[(bi >> 2) * 2 + (bi & 1)];
				sum += ((bi >> 1) & 1) ? tapHi<1>(reg, hk[t]) : tapHi<0>(reg, hk[t]);
			}
			o[j] = ((colok >> j) & 1u) ? min(sum, 255u) : 0u;
		}
#pragma unroll
		for (int t = 0; t + 1 < K; ++t) { ring[t][0] = ring[t + 1][0]; ring[t][1] = ring[t + 1][1]; ring[t][2] = ring[t + 1][2]; ring[t][3] = ring[t + 1][3]; }
		ring[K - 1][0] = (o[0] << 8) | (o[2] << 24); ring[K - 1][1] = (o[1] << 8) | (o[3] << 24);
		ring[K - 1][2] = (o[4] << 8) | (o[6] << 24); ring[K - 1][3] = (o[5] << 8) | (o[7] << 24);
		const int yo = yin - r;
		if (i < 2 * r || yo >= a.H) continue;
		uint2 q = make_uint2(0u, 0u);
		if (yo >= r && yo < a.H - r) {
			uint32_t p[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				uint32_t sum = 0;
#pragma unroll
				for (int t = 0; t < K; ++t) {
					const uint32_t reg = ring[t][(j >> 2) * 2 + (j & 1)];
					sum += ((j >> 1) & 1) ? tapHi<1>(reg, vk[t]) : tapHi<0>(reg, vk[t]);
				}
				p[j] = min(sum, 255u);
			}
			q.x = p[0] | (p[1] << 8) | (p[2] << 16) | (p[3] << 24);
			q.y = p[4] | (p[5] << 8) | (p[6] << 16) | (p[7] << 24);
		}
		*reinterpret_cast<uint2*>(dst + (size_t)yo * a.So) = q;
	}
}

template <int K>
static hipError_t launchK(const FxpArgs& hz, const FxpArgs& vt, int frames, bool fused, hipStream_t stream)
{
	const int groups = (hz.W + 7) / 8;
	if (fused) {
		FxpArgs f = hz;
		f.out = vt.out; f.outFrameStride = vt.outFrameStride; f.So = vt.So;
		hipLaunchKernelGGL((convlt_fxp_fused_kernel<K>), dim3((groups + 255) / 256, (hz.H + kFxpRows - 1) / kFxpRows, frames), dim3(256), 0, stream, f, vt);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((convlt_fxp_hz_kernel<K>), dim3((groups + 255) / 256, hz.H, frames), dim3(256), 0, stream, hz);
	hipLaunchKernelGGL((convlt_fxp_vt_kernel<K>), dim3((groups + 255) / 256, (vt.H + kFxpRows - 1) / kFxpRows, frames), dim3(256), 0, stream, vt);
	return hipGetLastError();
}

// in -> tmp (horizontal, hzKern) -> out (vertical, vtKern); in may alias out
hipError_t launch_convlt_fxp(const uint8_t* in, uint8_t* tmp, uint8_t* out, int W, int H, int S, size_t frameStride, int frames, const uint16_t* vtKern,
                             const uint16_t* hzKern, int K, hipStream_t stream)
{
	if (K < 3 || K > kFxpMaxTaps || !(K & 1)) return hipErrorInvalidValue;
	FxpArgs hz, vt;
	hz.in = in; hz.out = tmp; vt.in = tmp; vt.out = out;
	hz.W = vt.W = W; hz.H = vt.H = H; hz.S = vt.S = hz.So = vt.So = S;
	hz.inFrameStride = hz.outFrameStride = vt.inFrameStride = vt.outFrameStride = frameStride;
	for (int t = 0; t < kFxpMaxTaps; ++t) { hz.kern[t] = t < K ? hzKern[t] : 0u; vt.kern[t] = t < K ? vtKern[t] : 0u; }
	// the fused kernel reads input rows that other workgroups may already have overwritten when in == out: two passes then
	const size_t span = frameStride * (size_t)frames;
	const bool fused = !((in < out + span) && (out < in + span));
	switch (K) {
	case 3: return launchK<3>(hz, vt, frames, fused, stream);
	case 5: return launchK<5>(hz, vt, frames, fused, stream);
	case 7: return launchK<7>(hz, vt, frames, fused, stream);
	case 9: return launchK<9>(hz, vt, frames, fused, stream);
	case 11: return launchK<11>(hz, vt, frames, fused, stream);
	case 13: return launchK<13>(hz, vt, frames, fused, stream);
	default: return launchK<15>(hz, vt, frames, fused, stream);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Integer separable correlation, int16 out: CompVMathConvlt::convlt1<uint8_t, int16_t, int16_t> and <int16_t, int16_t, int16_t>
// (base/include/compv/base/math/compv_math_convlt.h:26-28,37-39; driver :98-173, hz :176-229, vt :232-292; AVX2 leaf
// base/math/intrin/x86/compv_math_convlt_intrin_avx2.cxx:314-430: int32 sums, packs = signed saturation).  This is the operator the
// gradient of the path is made of (canny_dete.cxx:237-241) -- there it is fused into the tile kernels and never materialised; these two
// kernels are its stand-alone form (compvhip_convlt1_8u16s16s / _16s16s16s), generic in the (odd) kernel size.  Horizontal pass into an
// int16 temporary (zero columns [0, r) and [W - r, W)), vertical pass over it (zero rows [0, r) and [H - r, H)), both saturating.
// One thread = 4 adjacent outputs of one row; the taps are wave-uniform kernel arguments.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sat16i(int v) { return min(max(v, -32768), 32767); }

template <typename TIn>
__global__ __launch_bounds__(256) void convlt_i16_hz_kernel(I16Args a)
{
	const int y = blockIdx.y;
	const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
	if (x0 >= a.W) return;
	const int r = a.K >> 1;
	const TIn* __restrict__ row = reinterpret_cast<const TIn*>(a.in) + (size_t)y * a.S;
	int16_t* __restrict__ dst = a.out + (size_t)y * a.So;
	int v[4 + kFxpMaxTaps - 1];
#pragma unroll
	for (int j = 0; j < 4 + kFxpMaxTaps - 1; ++j) {
		const int x = x0 - r + j;
		v[j] = (j < 4 + a.K - 1 && x >= 0 && x < a.W) ? (int)row[x] : 0;
	}
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = x0 + j;
		if (x >= a.W) break;
		int sum = 0;
#pragma unroll
		for (int t = 0; t < kFxpMaxTaps; ++t) if (t < a.K) sum += v[j + t] * a.kern[t];
		dst[x] = (x < r || x >= a.W - r) ? (int16_t)0 : (int16_t)sat16i(sum);
	}
}

__global__ __launch_bounds__(256) void convlt_i16_vt_kernel(I16Args a)
{
	const int y = blockIdx.y;
	const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
	if (x0 >= a.W) return;
	const int r = a.K >> 1;
	const int16_t* __restrict__ src = reinterpret_cast<const int16_t*>(a.in);
	int16_t* __restrict__ dst = a.out + (size_t)y * a.So;
	const bool border = (y < r || y >= a.H - r);
	int sum[4] = { 0, 0, 0, 0 };
	if (!border) {
		for (int t = 0; t < a.K; ++t) {
			const int16_t* __restrict__ row = src + (size_t)(y - r + t) * a.S;
			const int k = a.kern[t];
#pragma unroll
			for (int j = 0; j < 4; ++j) if (x0 + j < a.W) sum[j] += (int)row[x0 + j] * k;
		}
	}
#pragma unroll
	for (int j = 0; j < 4; ++j) if (x0 + j < a.W) dst[x0 + j] = border ? (int16_t)0 : (int16_t)sat16i(sum[j]);
}

// in (u8 when inIsU8, else s16; stride S elements) -> tmp (s16, stride W rounded up) -> out (s16, stride So)
hipError_t launch_convlt_i16(const void* in, bool inIsU8, int16_t* tmp, int16_t* out, int W, int H, int S, int So, const int16_t* vtKern, const int16_t* hzKern,
                             int K, hipStream_t stream)
{
	if (K < 1 || K > kFxpMaxTaps || !(K & 1)) return hipErrorInvalidValue;
	I16Args hz, vt;
	hz.in = in; hz.out = tmp; hz.W = W; hz.H = H; hz.S = S; hz.So = So; hz.K = K;
	vt.in = tmp; vt.out = out; vt.W = W; vt.H = H; vt.S = So; vt.So = So; vt.K = K;
	for (int t = 0; t < kFxpMaxTaps; ++t) { hz.kern[t] = t < K ? hzKern[t] : 0; vt.kern[t] = t < K ? vtKern[t] : 0; }
	const dim3 grid((unsigned)((W + 1023) / 1024), (unsigned)H);
	if (inIsU8) hipLaunchKernelGGL((convlt_i16_hz_kernel<uint8_t>), grid, dim3(256), 0, stream, hz);
	else hipLaunchKernelGGL((convlt_i16_hz_kernel<int16_t>), grid, dim3(256), 0, stream, hz);
	hipLaunchKernelGGL(convlt_i16_vt_kernel, grid, dim3(256), 0, stream, vt);
	return hipGetLastError();
}

} // namespace compvhip
