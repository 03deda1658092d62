// sobel_kernels.hip -- Sobel / Scharr / Prewitt edge detector (normalised L1 gradient magnitude) for gfx950.
//
// Replaces, behind compvhip_edge_dete_u8: CompVCornerDeteEdgeBase::process
// (core/features/edges/compv_core_feature_edge_dete.cxx:55-206) = convlt1 x2 + sumAbs + max + scaleAndClipPixel8.
//
// The reference materialises gx, gy (int16) and g (uint16) -- 6 B/px of intermediates.  Here the image is streamed
// twice through registers instead (stencil.hpp): pass 1 reduces gmax, pass 2 recomputes g and writes the scaled
// byte.  HBM traffic: 2 x 1 B/px read + 1 B/px write, no intermediate buffer.
//
// Bit-exactness notes:
//  * gmax only folds columns with (x & 7) in {0,1,2,4} -- quirk Q1 of CompVMathUtilsMax_16u_Intrin_SSE41
//    (base/math/intrin/x86/compv_math_utils_intrin_sse41.cxx:55-63).
//  * scale = 255.f / float(gmax) as one correctly rounded f32 division, out = min(255, trunc(float(g) * scale)) with
//    one correctly rounded f32 multiply (base/math/intrin/x86/compv_math_utils_intrin_sse2.cxx:165-..., cvttps).
//  * gmax == 0 (single-thread branch, edge_dete.cxx:199): scale = inf -> NaN/INT_MIN -> every output byte 0.
#include "stencil.hpp"
#include "kernels.hpp"

#include <type_traits>

namespace compvhip {

constexpr int kEdgeWaves = 4;

template <int A, int B, bool SCALE>
__global__ __launch_bounds__(kEdgeWaves * 64) void edge_dete_kernel(EdgeDeteArgs a)
{
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int tileX, group;
	if (!xcd_tile_map(blockIdx.x, a.tilesX, a.groups, tileX, group)) return;
	const int frame = group / a.blockRows;
	const int tileY = (group - frame * a.blockRows) * kEdgeWaves + wave;
	if (tileY >= a.tilesY) return;

	const int W = a.W, H = a.H, S = a.S;
	const int x0 = tileX * kTileW + lane * kLanePx;
	const int y0 = tileY * kTileH;
	const uint8_t* __restrict__ in = a.in + (size_t)frame * a.inFrameStride;
	uint8_t* __restrict__ out = a.out + (size_t)frame * a.outFrameStride;

	uint32_t colok = 0;
#pragma unroll
	for (int p = 0; p < 8; ++p) {
		const int x = x0 + p;
		if (x >= 1 && x <= W - 2) colok |= 1u << p;
	}

	float scale = 0.f;
	bool allZero = false;
	if (SCALE) {
		const unsigned int gmax = a.gmax[frame * kFrameSlot];
		allZero = (gmax == 0);
		scale = __fdiv_rn(255.f, (float)gmax);
	}

	Grad3Ring<A, B> st;
	st.reset();
	unsigned int vmax = 0;
	auto step = [&](auto phase, int it) {
		constexpr int PH = decltype(phase)::value;
		const int yin = y0 - 1 + it;
		const int yl = min(max(yin, 0), H - 1);
		const RowBytes rb = load_row(in + (size_t)yl * S, x0, S);
		int gg[10], ax[8];
		bool ng[8];
		st.template push<PH>(rb, gg, ax, ng);
		const int yc = yin - 1;
		if (it < 2 || yc >= H) return;
		const bool rowok = (yc >= 1) && (yc <= H - 2);
		uint32_t o0 = 0, o1 = 0;
#pragma unroll
		for (int p = 0; p < 8; ++p) {
			int g = gg[p + 1];
			g = min(g, 65535); // adds_epu16
			g = (rowok && ((colok >> p) & 1u)) ? g : 0;
			if (!SCALE) {
				if (p == 0 || p == 1 || p == 2 || p == 4) vmax = max(vmax, (unsigned int)g);
			}
			else {
				int q = (int)__fmul_rn((float)g, scale);
				q = min(q, 255);
				q = allZero ? 0 : q;
				if (p < 4) o0 |= (uint32_t)q << (8 * p); else o1 |= (uint32_t)q << (8 * (p - 4));
			}
		}
		if (SCALE) {
			if (x0 + 8 <= a.So) *reinterpret_cast<uint2*>(out + (size_t)yc * a.So + x0) = make_uint2(o0, o1);
		}
	};
	static_assert((kTileH + 2) % 2 == 0, "row loop is unrolled by 2");
	for (int it = 0; it < kTileH + 2; it += 2) {
		step(std::integral_constant<int, 0>{}, it);
		step(std::integral_constant<int, 1>{}, it + 1);
	}
	if (!SCALE) {
		for (int o = 32; o > 0; o >>= 1) vmax = max(vmax, (unsigned int)__shfl_down(vmax, o));
		if (lane == 0 && vmax) atomicMax(&a.gmax[frame * kFrameSlot], vmax);
	}
}

template <int A, int B>
static hipError_t launch_op(const EdgeDeteArgs& a0, int frames, hipStream_t stream)
{
	EdgeDeteArgs a = a0;
	a.blockRows = (a.tilesY + kEdgeWaves - 1) / kEdgeWaves;
	a.groups = a.blockRows * frames;
	hipError_t e = hipMemsetAsync(a.gmax, 0, sizeof(unsigned int) * frames * kFrameSlot, stream);
	if (e != hipSuccess) return e;
	dim3 grid(8 * ((a.groups + 7) / 8) * a.tilesX);
	dim3 block(kEdgeWaves * 64);
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
