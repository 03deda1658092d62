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

// XCD-aware tile mapping.  Workgroup b is observed to run on XCD b % 8, each XCD with a private L2.  Horizontally adjacent
// tiles share their 4-byte column halos (one extra 128-byte line each side per row), so all tilesX tiles of one "row group"
// (same frame, same block-row) are given to the SAME XCD, consecutively: the halo lines then hit in that XCD's L2 instead
// of being fetched from the fabric twice.  A pure performance remap: any placement is correct.
// grid = 8 * ceil(groups/8) * tilesX workgroups (1-D); returns false for padding workgroups.
__device__ __forceinline__ bool xcd_tile_map(int b, int tilesX, int groups, int& tileX, int& group)
{
	const int xcd = b & 7;
	const int k = b >> 3;             // index of this workgroup inside its XCD's queue
	group = (k / tilesX) * 8 + xcd;
	tileX = k - (k / tilesX) * tilesX;
	return group < groups;
}

__device__ __forceinline__ int absdiff(int a, int b)
{
	// v_sad_u32 d, a, b, 0 = |a-b|; written as asm because the compiler otherwise expands the intrinsic to sub/max/min
	int d;
	asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
// Lane masks straight from VOPC compares (the C++ route through bool + ballot costs a v_cndmask + v_cmp_ne per mask).
__device__ __forceinline__ uint64_t mask_ge_i32(int a, int b)
{
	uint64_t m;
	asm("v_cmp_ge_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
	return m;
}
__device__ __forceinline__ uint64_t mask_gt_i32_s(int a, int sb) // a > sb, sb wave-uniform
{
	uint64_t m;
	asm("v_cmp_lt_i32_e64 %0, %2, %1" : "=s"(m) : "v"(a), "s"(sb));
	return m;
}
__device__ __forceinline__ int absdiff_acc(int a, int b, int c)
{
	int d;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); // |a-b| + c
	return d;
}

// 16 input bytes around the lane's 8 pixels: columns x0-4 .. x0+11
struct RowBytes {
	uint32_t l, m0, m1, r;
};

// Loads never leave [rowptr, rowptr + S): S % 8 == 0 is a precondition of every device entry point.  Addresses are
// clamped instead of predicated (no divergent branches): a lane whose 8 pixels or 4-byte halos fall outside the row
// reads other in-row bytes, which only ever feed gradient columns x <= 0 or x >= W-1 -- those are forced to zero by
// the callers' column masks (zero OUTPUT border, compv_math_convlt.h:181-209).
__device__ __forceinline__ RowBytes load_row(const uint8_t* __restrict__ rowptr, int x0, int S)
{
	RowBytes rb;
	const int xm = min(x0, S - 8);
	const int xl = max(xm - 4, 0);
	const int xr = min(xm + 8, S - 4);
	const uint2 m = *reinterpret_cast<const uint2*>(rowptr + xm);
	rb.m0 = m.x; rb.m1 = m.y;
	rb.l = *reinterpret_cast<const uint32_t*>(rowptr + xl);
	rb.r = *reinterpret_cast<const uint32_t*>(rowptr + xr);
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

// Gradient columns gi = 0..9  <->  x = x0 - 1 + gi (8 own pixels + one neighbour each side).

// Rolling vertical state of the separable 3x3 operator with smoothing weights (A,B,A) and derivative (-1,0,1).
// The two-row rings are indexed with the compile-time phase PH = (row index) mod 2, so a row loop unrolled by an
// even factor needs no register-to-register copies to advance the window.
template <int A, int B>
struct Grad3Ring {
	int vr[2][12];   // unpacked input rows y-1 / y-2 (ring)
	int P[12];       // A*I[y-2] + B*I[y-1]
	int hy[2][10];   // hy[y-1] / hy[y-2] (ring)

	__device__ __forceinline__ void reset()
	{
#pragma unroll
		for (int j = 0; j < 12; ++j) { P[j] = 0; vr[0][j] = 0; vr[1][j] = 0; }
#pragma unroll
		for (int j = 0; j < 10; ++j) { hy[0][j] = 0; hy[1][j] = 0; }
	}

	// Push input row y (phase PH = parity of the push count); yields the gradient of row y-1 (valid once rows y-2..y were
	// pushed): g[gi] = |gx|+|gy| for the 10 columns, ax[p] = |gx| and ng[p] = ((gx ^ gy) < 0) for the 8 own pixels.
	template <int PH>
	__device__ __forceinline__ void push(const RowBytes& rb, int (&g)[10], int (&axo)[8], bool (&ng)[8])
	{
		int (&cur)[12] = vr[PH & 1];
		const int (&prev)[12] = vr[(PH + 1) & 1];
		unpack12(rb, cur);
		int C[12];
#pragma unroll
		for (int j = 0; j < 12; ++j) {
			C[j] = P[j] + A * cur[j];
			P[j] = A * prev[j] + B * cur[j];
		}
		int (&hyTop)[10] = hy[PH & 1]; // holds hy[y-2]; overwritten with hy[y] below
#pragma unroll
		for (int gi = 0; gi < 10; ++gi) {
			const int hyN = A * (cur[gi] + cur[gi + 2]) + B * cur[gi + 1];
			const int right = C[gi + 2], left = C[gi];
			const int top = hyTop[gi];
			const int ax = absdiff(right, left);
			g[gi] = absdiff_acc(hyN, top, ax);          // |gy| + |gx|
			if (gi >= 1 && gi <= 8) {
				axo[gi - 1] = ax;
				ng[gi - 1] = (right < left) != (hyN < top);
			}
			hyTop[gi] = hyN;
		}
	}
};

// 5x5 Sobel (Canny kernel size 5): vt {1,4,6,4,1}, hz {1,2,0,-2,-1} (base/include/compv/base/compv_features.h:129-130).
// Separable and linear without saturation (|gx|,|gy| <= 12240), so the vertical pass is done first on the unpacked
// columns and the horizontal pass second -- bit-identical to the reference's hz-then-vt order:
//   C5[x] = I[y-2]+4I[y-1]+6I[y]+4I[y+1]+I[y+2]     gx = C5[x-2] + 2C5[x-1] - 2C5[x+1] - C5[x+2]
//   D5[x] = I[y-2]+2I[y-1]-2I[y+1]-I[y+2]           gy = D5[x-2] + 4D5[x-1] + 6D5[x] + 4D5[x+1] + D5[x+2]
// The rare 5x5 path keeps its four previous input rows in a plain shift register (no phase unrolling).
struct Grad5State {
	int rows[4][14]; // input rows y-4 .. y-1, columns x0-3 .. x0+10

	__device__ __forceinline__ void reset()
	{
#pragma unroll
		for (int r = 0; r < 4; ++r)
#pragma unroll
			for (int j = 0; j < 14; ++j) rows[r][j] = 0;
	}

	// Push input row y; yields the gradient of row y-2 (valid once rows y-4..y were pushed).
	__device__ __forceinline__ void push(const RowBytes& rb, int (&g)[10], int (&axo)[8], bool (&ng)[8])
	{
		int cur[14];
		cur[0] = (rb.l >> 8) & 0xff; cur[1] = (rb.l >> 16) & 0xff; cur[2] = rb.l >> 24;
		cur[3] = rb.m0 & 0xff; cur[4] = (rb.m0 >> 8) & 0xff; cur[5] = (rb.m0 >> 16) & 0xff; cur[6] = rb.m0 >> 24;
		cur[7] = rb.m1 & 0xff; cur[8] = (rb.m1 >> 8) & 0xff; cur[9] = (rb.m1 >> 16) & 0xff; cur[10] = rb.m1 >> 24;
		cur[11] = rb.r & 0xff; cur[12] = (rb.r >> 8) & 0xff; cur[13] = (rb.r >> 16) & 0xff;
		int C5[14], D5[14];
#pragma unroll
		for (int j = 0; j < 14; ++j) {
			C5[j] = rows[0][j] + 4 * (rows[1][j] + rows[3][j]) + 6 * rows[2][j] + cur[j];
			D5[j] = rows[0][j] + 2 * (rows[1][j] - rows[3][j]) - cur[j];
			rows[0][j] = rows[1][j]; rows[1][j] = rows[2][j]; rows[2][j] = rows[3][j]; rows[3][j] = cur[j];
		}
#pragma unroll
		for (int gi = 0; gi < 10; ++gi) {
			const int j = gi + 2;
			const int gx = C5[j - 2] + 2 * (C5[j - 1] - C5[j + 1]) - C5[j + 2];
			const int gy = D5[j - 2] + 4 * (D5[j - 1] + D5[j + 1]) + 6 * D5[j] + D5[j + 2];
			const int ax = abs(gx), ay = abs(gy);
			g[gi] = ax + ay;
			if (gi >= 1 && gi <= 8) {
				axo[gi - 1] = ax;
				ng[gi - 1] = (gx ^ gy) < 0;
			}
		}
	}
};

} // namespace compvhip
