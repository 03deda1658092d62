// sobel_kernels.hip -- Sobel / Scharr / Prewitt edge detector (normalised L1 gradient magnitude) for gfx950.
//
// Replaces, behind compvhip_edge_dete_u8 / compvhip_plan_edge_dete: CompVCornerDeteEdgeBase::process
// (core/features/edges/compv_core_feature_edge_dete.cxx:55-206) = convlt1 x2 + sumAbs + max + scaleAndClipPixel8.
//
// The reference materialises gx, gy (int16) and g (uint16) -- 6 B/px of intermediates.  Here the image is streamed twice through
// registers instead: pass 1 reduces gmax, pass 2 recomputes g and writes the scaled byte.  HBM traffic: 2 x 1 B/px read + 1 B/px write,
// no intermediate buffer.  Both passes use the packed gradient of the Canny tile kernel (canny_swar_kernels.hip): two pixels per
// 32-bit register in biased u16 halves, four pixels per lane, three-operand forms (v_xad / v_lshl_add / v_add3) -- 31 VALU instructions
// per 256-pixel row for Sobel instead of the ~110 of the 32-bit stencil of rounds 1-3 (round 4: 0.274 -> see DESIGN.md section 4.4).  There is no
// NMS here, so a wave owns all 256 columns of its tile (no halo lanes) and every output row leaves as one 256-byte store per wave.
//
// Bit-exactness notes:
//  * gmax only folds columns with (x & 7) in {0,1,2,4} -- quirk Q1 of CompVMathUtilsMax_16u_Intrin_SSE41
//    (base/math/intrin/x86/compv_math_utils_intrin_sse41.cxx:55-63).
//  * scale = 255.f / float(gmax) as one correctly rounded f32 division, out = min(255, trunc(float(g) * scale)) with
//    one correctly rounded f32 multiply (base/math/intrin/x86/compv_math_utils_intrin_sse2.cxx:165-..., cvttps): here v_floor_f32 of the
//    (non-negative) product, then v_cvt_pk_u8_f32, which saturates at 255 and rounds to nearest -- exact on the integer it is given
//    (tools/microbench/cvt_pk_u8_test.hip: the conversion alone would round 0.999 up to 1).
//  * gmax == 0 (single-thread branch, edge_dete.cxx:199): scale = inf -> NaN/INT_MIN -> every output byte 0.
//  * |gx| + |gy| <= 2 * 16 * 255 = 8160 for the widest operator (Scharr): the reference's saturating add never saturates.
#include "stencil.hpp"
#include "kernels.hpp"

#include <type_traits>

namespace compvhip {

namespace {

constexpr int kEdPx = 4;          // pixels per lane
constexpr int kEdCols = 256;      // columns per wave tile
constexpr int kEdRows = 62;       // rows per wave tile (+ 2 = a multiple of kEdAhead)
constexpr int kEdAhead = 4;       // input rows in flight per wave

__device__ __forceinline__ uint32_t ed_pk_max_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t ed_lshl_add(uint32_t a, uint32_t b) { uint32_t d; asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }    // 2a + b
__device__ __forceinline__ uint32_t ed_lshl3_add(uint32_t a, uint32_t b) { uint32_t d; asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }   // 8a + b
__device__ __forceinline__ uint32_t ed_xad(uint32_t a, uint32_t sk, uint32_t c) { uint32_t d; asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sk), "v"(c)); return d; }   // (a ^ k) + c
__device__ __forceinline__ uint32_t ed_add3s(uint32_t a, uint32_t b, uint32_t sk) { uint32_t d; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(sk)); return d; }
// float(half h of x) * scale, floored -- the integer part of the reference's product
template <int HALF> __device__ __forceinline__ float ed_scaled(uint32_t x, float scale)
{
	float f;
	if (HALF == 0) asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f) : "v"(x));
	else asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(x));
	return __builtin_floorf(__fmul_rn(f, scale));
}
template <int BYTE> __device__ __forceinline__ uint32_t ed_put_u8(float v, uint32_t acc)
{
	asm("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(v), "n"(BYTE));
	return acc;
}

} // namespace

// smoothing weights (A, B, A), derivative (-1, 0, 1); SCALE = false: gmax pass, true: output pass
template <int A, int B, bool SCALE>
__global__ __launch_bounds__(64, 8) void edge_dete_kernel(EdgeDeteArgs a)
{
	static_assert((A == 1 && (B == 2 || B == 1)) || (A == 3 && B == 10), "Sobel, Prewitt, Scharr");
	const int lane = threadIdx.x & 63;
	int tileX, group;
	if (!xcd_tile_map(blockIdx.x, a.tilesX, a.groups, tileX, group)) return;
	const int frame = group / a.blockRows;
	const int tileY = group - frame * a.blockRows;

	const int W = a.W, H = a.H, S = a.S;
	const int x0 = tileX * kEdCols + lane * kEdPx;
	const int y0 = tileY * kEdRows;
	const uint8_t* __restrict__ in = a.in + (size_t)frame * a.inFrameStride;
	uint8_t* __restrict__ out = a.out + (size_t)frame * a.outFrameStride;

	// biases of the packed halves: d = R - L + 255; gx carries (2A + B) * 255 + fix = BX; gy = hy[y+1] - hy[y-1] + KY with KY = 2^n - 1 >= max hy
	constexpr uint32_t kSum = 2 * A + B;                                  // 4 | 3 | 16
	constexpr uint32_t BX = (A == 3) ? 4096u : 1024u;
	constexpr uint32_t KY = (A == 3) ? 4095u : 1023u;
	constexpr uint32_t kFix = BX - kSum * 255u;                           // 4 | 259 | 16
	uint32_t k255 = 0x00ff00ffu, kFixP = kFix * 0x00010001u, kKY = KY * 0x00010001u, kNegBias = 0u - (BX + KY) * 0x00010001u;
	asm volatile("" : "+s"(k255), "+s"(kFixP), "+s"(kKY), "+s"(kNegBias));

	// zero OUTPUT border of the convolution (compv_math_convlt.h:181-209): g = 0 outside columns [1, W-2] and rows [1, H-2]
	const bool edgeTile = (tileX == 0) || (tileX * kEdCols + kEdCols > W - 1);
	uint32_t okm[2] = { 0xffffffffu, 0xffffffffu };
	if (edgeTile) {
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const int xa = x0 + 2 * k, xb = xa + 1;
			okm[k] = ((xa >= 1 && xa <= W - 2) ? 0x0000ffffu : 0u) | ((xb >= 1 && xb <= W - 2) ? 0xffff0000u : 0u);
		}
	}
	const bool borderTile = edgeTile || (tileY == 0) || (y0 + kEdRows >= H - 1);

	// row loads through a buffer descriptor (row offset in an SGPR, the lane's column offsets loop-invariant); columns clamped into the row: clamped
	// lanes only feed columns whose g is forced to 0
	const uint32_t xm = (uint32_t)min(max(x0, 0), S - 4);
	const uint32_t xl = (uint32_t)min(max(x0 - 4, 0), S - 4);
	const uint32_t xr = (uint32_t)min(max(x0 + 4, 0), S - 4);
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(in), 0, (int)((size_t)H * S), 0x00020000);
	auto load = [&](int y, uint32_t (&v)[3]) {
		const int so = min(max(y, 0), H - 1) * S;
		v[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xm, so, 0);
		v[1] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xl, so, 0);
		v[2] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xr, so, 0);
	};

	float scale = 0.f;
	bool allZero = false;
	if (SCALE) {
		const unsigned int gmax = a.gmax[frame * kFrameSlot];
		allZero = (gmax == 0);
		scale = __fdiv_rn(255.f, (float)gmax);
	}

	uint32_t P[2] = { 0, 0 }, dprev[2] = { 0, 0 }, hy[2][2] = { { 0, 0 }, { 0, 0 } };
	uint32_t vmax[2] = { 0, 0 };
	// input rows are fetched kEdAhead steps ahead (a ring of register sets indexed by the step's phase; the row loop is unrolled by kEdAhead): a step
	// of this kernel is ~35 instructions, eight waves per SIMD finish one every ~0.5 us, and a load that misses takes 1 - 2 us under load -- two rows
	// ahead (what the Canny kernel with its 3x longer steps needs) left the gmax pass at 2.5 TB/s
	uint32_t nb[kEdAhead][3];
#pragma unroll
	for (int k = 0; k < kEdAhead; ++k) load(y0 - 1 + k, nb[k]);

	// step `it` pushes input row y0 - 1 + it and yields the gradient of row yc = y0 + it - 2
	auto step = [&](auto phase, int it) {
		constexpr int PH = decltype(phase)::value;
		const uint32_t m = nb[PH % kEdAhead][0], l = nb[PH % kEdAhead][1], r = nb[PH % kEdAhead][2];
		load(y0 - 1 + it + kEdAhead, nb[PH % kEdAhead]);
		uint32_t Ak[2], L[3];
		Ak[0] = __builtin_amdgcn_perm(0u, m, 0x0c010c00u);     // (p0, p1)
		Ak[1] = __builtin_amdgcn_perm(0u, m, 0x0c030c02u);     // (p2, p3)
		L[0] = __builtin_amdgcn_perm(m, l, 0x0c040c03u);       // (p-1, p0)
		L[1] = __builtin_amdgcn_perm(0u, m, 0x0c020c01u);      // (p1, p2)
		L[2] = __builtin_amdgcn_perm(r, m, 0x0c040c03u);       // (p3, p4)
		uint32_t gq[2];
		uint32_t (&hyTop)[2] = hy[PH & 1];
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const uint32_t Lk = L[k], Rk = L[k + 1], Ck = Ak[k];
			const uint32_t s = Lk + Rk;
			uint32_t hyN, gxb;
			const uint32_t d = ed_xad(Lk, k255, Rk);                   // R - L + 255
			if (A == 1 && B == 2) {                                    // Sobel
				hyN = ed_lshl_add(Ck, s);
				gxb = ed_add3s(P[k], d, kFixP);
				P[k] = ed_lshl_add(d, dprev[k]);
				dprev[k] = d;
			}
			else if (A == 1) {                                         // Prewitt
				hyN = s + Ck;
				gxb = ed_add3s(P[k], d, kFixP);
				P[k] = dprev[k] + d;
				dprev[k] = d;
			}
			else {                                                     // Scharr: 3, 10, 3 (dprev holds 3 d[y-1])
				hyN = ed_lshl_add(s, s) + ed_lshl3_add(Ck, Ck + Ck);
				const uint32_t d3 = ed_lshl_add(d, d);
				gxb = ed_add3s(P[k], d3, kFixP);
				P[k] = dprev[k] + ed_lshl3_add(d, d + d);
				dprev[k] = d3;
			}
			const uint32_t gyb = ed_xad(hyTop[k], kKY, hyN);           // hy[y+1] - hy[y-1] + KY
			hyTop[k] = hyN;
			const uint32_t mx = ed_pk_max_u16(gxb, 2u * BX * 0x00010001u - gxb);     // |gx| + BX
			const uint32_t my = ed_pk_max_u16(gyb, 2u * KY * 0x00010001u - gyb);     // |gy| + KY
			gq[k] = ed_add3s(mx, my, kNegBias);                        // |gx| + |gy| (each half >= its bias: no borrow between the halves)
		}
		if (it < 2 || it >= kEdRows + 2) return;   // (the unrolled loop runs a few steps past the tile)
		const int yc = y0 + it - 2;
		if (borderTile) {
			asm volatile("" : "+v"(gq[0]), "+v"(gq[1]));
			const uint32_t rowm = (yc >= 1 && yc <= H - 2) ? 0xffffffffu : 0u;
			gq[0] &= okm[0] & rowm; gq[1] &= okm[1] & rowm;
		}
		if (!SCALE) {
			vmax[0] = ed_pk_max_u16(vmax[0], gq[0]);
			vmax[1] = ed_pk_max_u16(vmax[1], gq[1]);
		}
		else {
			if (yc >= H) return;   // uniform
			uint32_t o = 0u;
			if (!allZero) {
				o = ed_put_u8<0>(ed_scaled<0>(gq[0], scale), o);
				o = ed_put_u8<1>(ed_scaled<1>(gq[0], scale), o);
				o = ed_put_u8<2>(ed_scaled<0>(gq[1], scale), o);
				o = ed_put_u8<3>(ed_scaled<1>(gq[1], scale), o);
			}
			if (x0 + 4 <= a.So) *reinterpret_cast<uint32_t*>(out + (size_t)yc * a.So + x0) = o;
		}
	};
	static_assert(kEdAhead == 4, "row loop is unrolled by kEdAhead");
	for (int it = 0; it < kEdRows + 2; it += 4) {
		step(std::integral_constant<int, 0>{}, it);
		step(std::integral_constant<int, 1>{}, it + 1);
		step(std::integral_constant<int, 2>{}, it + 2);
		step(std::integral_constant<int, 3>{}, it + 3);
	}
	if (!SCALE) {
		// quirk Q1: only columns with (x & 7) in {0, 1, 2, 4} take part in the maximum: an even lane (x0 & 7 == 0) contributes p0, p1, p2, an odd one p0
		unsigned int v;
		if (lane & 1) v = vmax[0] & 0xffffu;
		else v = max(max(vmax[0] & 0xffffu, vmax[0] >> 16), vmax[1] & 0xffffu);
		for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned int)__shfl_down(v, o));
		if (lane == 0 && v) atomicMax(&a.gmax[frame * kFrameSlot], v);
	}
}

template <int A, int B>
static hipError_t launch_op(const EdgeDeteArgs& a0, int frames, hipStream_t stream)
{
	EdgeDeteArgs a = a0;
	a.tilesX = (a.W + kEdCols - 1) / kEdCols;
	a.tilesY = (a.H + kEdRows - 1) / kEdRows;
	a.blockRows = a.tilesY;
	a.groups = a.blockRows * frames;
	hipError_t e = hipMemsetAsync(a.gmax, 0, sizeof(unsigned int) * frames * kFrameSlot, stream);
	if (e != hipSuccess) return e;
	dim3 grid(8 * ((a.groups + 7) / 8) * a.tilesX);
	dim3 block(64);
	hipLaunchKernelGGL((edge_dete_kernel<A, B, false>), grid, block, 0, stream, a);
	hipLaunchKernelGGL((edge_dete_kernel<A, B, true>), grid, block, 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_edge_dete(const EdgeDeteArgs& a, int op, int frames, hipStream_t stream)
{
	switch (op) {
	case 0: return launch_op<1, 2>(a, frames, stream);   // Sobel   {1,2,1}
	case 2: return launch_op<3, 10>(a, frames, stream);  // Scharr  {3,10,3}
	case 3: return launch_op<1, 1>(a, frames, stream);   // Prewitt {1,1,1}
	default: return hipErrorInvalidValue;
	}
}

} // namespace compvhip
