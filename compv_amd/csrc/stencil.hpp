// stencil.hpp -- register-resident row-streaming 3x3 gradient for gfx950 (wave64).
//
// Work decomposition (shared by the Canny and Sobel kernels): one wavefront owns a tile of kTileW = 512 columns
// (64 lanes x 8 adjacent pixels, so a wave row-load is one contiguous 512-byte burst) and marches DOWN the rows of
// the tile keeping the separable-filter state in VGPRs.  Every input byte is fetched from HBM once (plus the
// 4/68 row halo between vertically adjacent tiles); nothing intermediate (gx, gy, g -- 6 B/px in the reference,
// core/features/edges/compv_core_feature_canny_dete.cxx:133-147) ever leaves the register file.
//
// Maths restated (reference base/include/compv/base/math/compv_math_convlt.h:98-292 with the
// kernels at base/include/compv/base/compv_features.h:124-133), for vertical-smoothing weights (A,B,A):
//   C[y][x]  = A*I[y-1][x] + B*I[y][x] + A*I[y+1][x]          (vertical smooth)
//   hy[y][x] = A*I[y][x-1] + B*I[y][x] + A*I[y][x+1]          (horizontal smooth)
//   gx = C[y][x+1] - C[y][x-1],  gy = hy[y+1][x] - hy[y-1][x],  g = |gx| + |gy|
// |gx|,|gy| are formed as |a-b| of two non-negative sums (one v_sad_u32 each), the sign of gx^gy from two compares.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace compvhip {

constexpr int kLanePx = 8;                 // pixels per lane per row
constexpr int kTileW = 64 * kLanePx;       // 512 columns per wave
constexpr int kTileH = 64;                 // output rows per wave tile (one per lane in the flood stage)

__device__ __forceinline__ int absdiff(int a, int b)
{
	// v_sad_u32 d, a, b, 0 = |a-b|; written as asm because the compiler otherwise expands the intrinsic to sub/max/min
	int d;
	asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
	return d;
}

// 16 input bytes around the lane's 8 pixels: columns x0-4 .. x0+11
struct RowBytes {
	uint32_t l, m0, m1, r;
};

// Loads never leave [rowptr, rowptr + S): S % 8 == 0 is a precondition of every device entry point.
__device__ __forceinline__ RowBytes load_row(const uint8_t* __restrict__ rowptr, int x0, int S)
{
	RowBytes rb;
	rb.l = rb.m0 = rb.m1 = rb.r = 0u;
	if (x0 < S) {
		const uint2 m = *reinterpret_cast<const uint2*>(rowptr + x0);
		rb.m0 = m.x; rb.m1 = m.y;
		if (x0 >= 4) rb.l = *reinterpret_cast<const uint32_t*>(rowptr + x0 - 4);
		if (x0 + 12 <= S) rb.r = *reinterpret_cast<const uint32_t*>(rowptr + x0 + 8);
	}
	return rb;
}

// v[j] = I[x0 - 2 + j], j = 0..11
__device__ __forceinline__ void unpack12(const RowBytes& rb, int (&v)[12])
{
	v[0] = (rb.l >> 16) & 0xff; v[1] = rb.l >> 24;
	v[2] = rb.m0 & 0xff; v[3] = (rb.m0 >> 8) & 0xff; v[4] = (rb.m0 >> 16) & 0xff; v[5] = rb.m0 >> 24;
	v[6] = rb.m1 & 0xff; v[7] = (rb.m1 >> 8) & 0xff; v[8] = (rb.m1 >> 16) & 0xff; v[9] = rb.m1 >> 24;
	v[10] = rb.r & 0xff; v[11] = (rb.r >> 8) & 0xff;
}

// Gradient row for the 10 columns gi = 0..9  <->  x = x0 - 1 + gi (8 own pixels + one neighbour each side).
struct GradRow {
	int ax[10];   // |gx|
	int ay[10];   // |gy|
	bool ng[10];  // (gx ^ gy) < 0
};

// Rolling vertical state of the separable 3x3 operator with smoothing weights (A,B,A) and derivative (-1,0,1).
template <int A, int B>
struct Grad3State {
	int P[12];    // A*I[y-2] + B*I[y-1]
	int A1[12];   // I[y-1]
	int hyA[10];  // hy[y-1]
	int hyB[10];  // hy[y-2]

	__device__ __forceinline__ void reset()
	{
#pragma unroll
		for (int j = 0; j < 12; ++j) { P[j] = 0; A1[j] = 0; }
#pragma unroll
		for (int j = 0; j < 10; ++j) { hyA[j] = 0; hyB[j] = 0; }
	}

	// Push input row y (v = its 12 unpacked columns); returns the gradient of row y-1 (valid once rows y-2..y were pushed).
	__device__ __forceinline__ void push(const int (&v)[12], GradRow& out)
	{
		int C[12];
#pragma unroll
		for (int j = 0; j < 12; ++j) {
			C[j] = P[j] + A * v[j];
			P[j] = A * A1[j] + B * v[j];
			A1[j] = v[j];
		}
#pragma unroll
		for (int gi = 0; gi < 10; ++gi) {
			const int hyN = A * (v[gi] + v[gi + 2]) + B * v[gi + 1];
			const int right = C[gi + 2], left = C[gi];
			const int top = hyB[gi];
			out.ax[gi] = absdiff(right, left);
			out.ay[gi] = absdiff(hyN, top);
			out.ng[gi] = (right < left) != (hyN < top);
			hyB[gi] = hyA[gi];
			hyA[gi] = hyN;
		}
	}
};

} // namespace compvhip
