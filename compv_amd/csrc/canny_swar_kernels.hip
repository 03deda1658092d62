// canny_swar_kernels.hip -- fused Sobel 3x3 -> NMS -> tile hysteresis for gfx950, second generation.
//
// Replaces (behind compvhip_canny_u8 / compvhip_plan_canny / compvhip_plan_pipeline, kernel size 3):
//   CompVEdgeDeteCanny::process            core/features/edges/compv_core_feature_canny_dete.cxx:123-331
//   nms_gather / nms_apply / hysteresis    ...canny_dete.cxx:334-528, row leaves :566-680,
//   CompVCannyNMSGatherRow_16mpw_Intrin_AVX2  core/features/edges/intrin/x86/compv_core_feature_canny_dete_intrin_avx2.cxx:150-235
//
// What changed against canny_tile_kernel (canny_kernels.hip, still used for the 5x5 gradient): that kernel runs the direction
// class + four neighbour maxima of the NMS on every pixel and is bound by VALU issue (213 VALU per 512-px row, mostly
// opcodes that issue at half rate on gfx950: tools/microbench/valu_rate_bench2/3).  Here
//   * the dense stage only produces the gradient magnitude, two pixels per 32-bit register (SWAR on u16 halves): sums and
//     differences of 8-bit pixels never carry across the halves once the signed terms are biased (d + 256, gx + 1024,
//     gy + 1024), so plain v_add_u32 / v_sub_u32 / v_or_b32 -- the full-rate opcodes -- do packed arithmetic; |.| is one
//     v_pk_max_u16 of (v, 2*bias - v);
//   * the rows of g (u16, pixel order) go through a 3-row LDS ring; pixels with g > tLow (~10 % on the benchmark frames)
//     are compacted into a per-row candidate list (wave prefix sum by DPP), and the NMS rule of the reference -- direction
//     class from |gx|,|gy| with the Q16 constants 27145 / 158217, compare with the two neighbours along the gradient -- is
//     evaluated one candidate per lane, reading its two neighbours from the ring;
//   * results are OR-ed into 512-bit row masks (weak / strong) in LDS, collected into the registers of the lane that owns the
//     row every 8 rows, and the strong -> weak flood inside the tile is the lane == row carry-chain flood of the first kernel.
// A wave owns 496 output columns x 64 rows: lanes 0 and 63 compute the gradient of the 8 columns either side of the tile (the
// halo the NMS of columns 0 and 495 needs) but own no pixels, so no gradient column is computed twice inside a wave and tiles
// stay 8-byte aligned (496 = 62 x 8).  Every input byte is still fetched once per tile (+ 4/64 row halo, + 16/496 column halo).
#include "stencil.hpp"
#include "kernels.hpp"

#include <cstdlib>
#include <type_traits>

namespace compvhip {

namespace {

constexpr int kSwCols = 496;                 // output columns per wave tile (lanes 1..62)
constexpr int kSwRowBytes = 1024;            // one ring row: 512 u16 in pixel order
constexpr int kSwG = 0;                      // byte offsets inside a wave's LDS block
constexpr int kSwAux = kSwG + 3 * kSwRowBytes;
constexpr int kSwList = kSwAux + 2 * kSwRowBytes;
constexpr int kSwMask = kSwList + 1024;      // [8 rows][16 dwords weak | 16 dwords strong]
constexpr int kSwLdsBytes = kSwMask + 8 * 128;   // 7168 B per wave

constexpr uint32_t kBiasD = 0x01000100u;     // +256 per half: horizontal difference R - L
constexpr uint32_t kBias1k = 0x04000400u;    // +1024 per half: gx, gy
constexpr uint32_t kBias2k = 0x08000800u;    // +2048 per half: g' = g + 2048 (and "2 * bias" of the absolute value)

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
	uint32_t d;
	asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) // (mask & a) | (~mask & b)
{
	uint32_t d;
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
	return d;
}

// x + x as a full-rate v_add_u32 (the compiler canonicalises it to v_lshlrev_b32, which issues at half rate on gfx950)
__device__ __forceinline__ int twice(int x)
{
	int d;
	asm("v_add_u32 %0, %1, %1" : "=v"(d) : "v"(x));
	return d;
}
__device__ __forceinline__ uint32_t twice(uint32_t x) { return (uint32_t)twice((int)x); }

// inclusive prefix sum over the 64 lanes (values < 2^25): four row_shr steps inside the rows of 16, then the row totals
__device__ __forceinline__ int wave_scan_incl(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1 and 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2 and 3
	return x;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
template <bool GAP>
__global__ __launch_bounds__(kCannyWaves * 64, 4) void canny_swar_tile_kernel(CannyArgs a)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds_all[kCannyWaves][kSwLdsBytes];

	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int tileX, group;
	if (!xcd_tile_map(blockIdx.x, a.tilesX, a.groups, tileX, group)) return;
	const int frame = group / a.blockRows;
	const int tileY = (group - frame * a.blockRows) * kCannyWaves + wave;
	if (tileY >= a.tilesY) return; // whole wave

	uint8_t* const lds = &lds_all[wave][0];
	const int W = a.W, H = a.H, S = a.S;
	const int xbase = tileX * kSwCols - 8;          // column of local bit 0 (lane 0 = the left halo lane)
	const int x0 = xbase + lane * kLanePx;
	const int y0 = tileY * kTileH;
	const uint8_t* __restrict__ in = a.in + (size_t)frame * a.inFrameStride;

	int tLow = a.tLow, tHigh = a.tHigh;
	if (a.thrDev) { const int2 t = a.thrDev[frame]; tLow = t.x; tHigh = t.y; }
	tLow = min(__builtin_amdgcn_readfirstlane(tLow), 4000);   // g <= 2040: larger thresholds select nothing, and the packed compare stays in range
	tHigh = min(__builtin_amdgcn_readfirstlane(tHigh), 8000);
	const int tLowQ = tLow + 2048, tHighQ = tHigh + 2048; // thresholds on g' = g + 2048

	// g is forced to 0 (g' = 2048) outside columns [1, W-2]: zero OUTPUT border of the convolution (compv_math_convlt.h:181-209)
	const bool edgeTile = (xbase < 1) || (xbase + 512 > W - 1);
	uint32_t okm[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
	if (edgeTile) {
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int xa = x0 + 2 * k, xb = xa + 1;
			okm[k] = ((xa >= 1 && xa <= W - 2) ? 0x0000ffffu : 0u) | ((xb >= 1 && xb <= W - 2) ? 0xffff0000u : 0u);
		}
	}
	// lanes 0 and 63 own no pixels (they only provide the column halo)
	const bool owner = (lane >= 1 && lane <= 62);
	// tiles whose gradient rows touch the image border rows (g forced to 0 there) or run past the image
	const bool vEdgeTile = (tileY == 0) || (y0 + kTileH + 1 >= H - 1);

	// clamped row loads (never leave [row, row + S); clamped lanes only feed columns whose g is forced to 0)
	const int xm = min(max(x0, 0), S - 8);
	const int xl = min(max(x0 - 4, 0), S - 4);
	const int xr = min(max(x0 + 8, 0), S - 4);
	auto load = [&](int y, uint2& m, uint32_t& l, uint32_t& r) {
		const uint8_t* __restrict__ row = in + (size_t)min(max(y, 0), H - 1) * S;
		// uniform row base + unsigned 32-bit lane offset: global_load with an SGPR base, no per-lane 64-bit address arithmetic
		m = *reinterpret_cast<const uint2*>(row + (uint32_t)xm);
		l = *reinterpret_cast<const uint32_t*>(row + (uint32_t)xl);
		r = *reinterpret_cast<const uint32_t*>(row + (uint32_t)xr);
	};

	// rolling state: two pixels per register, pairs k = (x0 + 2k, x0 + 2k + 1)
	uint32_t P[4] = { 0, 0, 0, 0 };        // d[y-2] + 2 d[y-1]   (bias 768)
	uint32_t dprev[4] = { 0, 0, 0, 0 };    // d[y-1]              (bias 256)
	uint32_t hy[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } }; // horizontal smooth of rows y-1 / y-2 (ring)
	uint32_t gprev[4] = { 0, 0, 0, 0 };    // g' of the previous gradient row (the row whose candidates are processed this step)
	uint64_t Wm[8], Em[8];                 // lane == row: 512-bit weak / strong masks of tile row `lane`, collected every 8 rows
#pragma unroll
	for (int m = 0; m < 8; ++m) { Wm[m] = 0; Em[m] = 0; }

	uint32_t* const maskw = reinterpret_cast<uint32_t*>(lds + kSwMask);
	{ // zero the mask ring (1 KB = 64 lanes x 16 B)
		*reinterpret_cast<uint4*>(lds + kSwMask + lane * 16) = make_uint4(0, 0, 0, 0);
	}

	uint2 nm; uint32_t nl, nr;
	load(y0 - 2, nm, nl, nr);

	// One row step: push input row yin = y0 - 2 + it  ->  gradient row yc = yin - 1  ->  NMS of row yo = yc - 1.
	auto step = [&](auto phase, int it) {
		constexpr int PH = decltype(phase)::value;
		constexpr int sD = PH % 3, sC = (PH + 2) % 3, sU = (PH + 1) % 3; // ring slots of rows yo+1 (written now), yo, yo-1
		constexpr int aNew = PH % 2, aMid = (PH + 1) % 2;
		const int yin = y0 - 2 + it;
		const uint2 m = nm; const uint32_t l = nl, r = nr;
		if (!(a.dbg & 16)) load(yin + 1, nm, nl, nr); // prefetch

		// ---- dense stage: packed pairs straight from the raw dwords (one v_perm each) ----
		uint32_t A[4], L[4], R3;
		A[0] = __builtin_amdgcn_perm(0u, m.x, 0x0c010c00u); A[1] = __builtin_amdgcn_perm(0u, m.x, 0x0c030c02u);
		A[2] = __builtin_amdgcn_perm(0u, m.y, 0x0c010c00u); A[3] = __builtin_amdgcn_perm(0u, m.y, 0x0c030c02u);
		L[0] = __builtin_amdgcn_perm(m.x, l, 0x0c040c03u);      // (p-1, p0)
		L[1] = __builtin_amdgcn_perm(0u, m.x, 0x0c020c01u);     // (p1, p2)
		L[2] = __builtin_amdgcn_perm(m.y, m.x, 0x0c040c03u);    // (p3, p4)
		L[3] = __builtin_amdgcn_perm(0u, m.y, 0x0c020c01u);     // (p5, p6)
		R3 = __builtin_amdgcn_perm(r, m.y, 0x0c040c03u);        // (p7, p8)
		uint32_t gq[4], aux[4];
		uint32_t (&hyTop)[4] = hy[PH & 1]; // hy of row yin-2; overwritten with hy of row yin
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t Lk = L[k], Rk = (k < 3) ? L[k + 1] : R3, Ck = A[k];
			const uint32_t s = Lk + Rk;
			const uint32_t hyN = s + twice(Ck);                        // I[x-1] + 2 I[x] + I[x+1]           (<= 1020)
			const uint32_t d = (Rk | kBiasD) - Lk;                     // I[x+1] - I[x-1] + 256
			const uint32_t gxb = P[k] + d;                             // gx + 1024
			P[k] = dprev[k] + twice(d);
			dprev[k] = d;
			const uint32_t gyb = (hyN | kBias1k) - hyTop[k];           // gy + 1024
			hyTop[k] = hyN;
			const uint32_t mx = pk_max_u16(gxb, kBias2k - gxb);        // |gx| + 1024
			const uint32_t my = pk_max_u16(gyb, kBias2k - gyb);        // |gy| + 1024
			gq[k] = mx + my;                                           // g + 2048
			aux[k] = bfi(kBias1k, gxb ^ gyb, mx);                      // bits 0..9 |gx|, bit 10 = ((gx ^ gy) < 0)
		}
		const int yc = yin - 1;
		if (edgeTile) {
#pragma unroll
			for (int k = 0; k < 4; ++k) gq[k] = bfi(okm[k], gq[k], kBias2k);
		}
		if (vEdgeTile) {
			if (!(yc >= 1 && yc <= H - 2)) { // image border rows (and rows past the image): g = 0
#pragma unroll
				for (int k = 0; k < 4; ++k) gq[k] = kBias2k;
			}
		}
		if (!(a.dbg & 32)) {
		*reinterpret_cast<uint4*>(lds + kSwG + sD * kSwRowBytes + lane * 16) = make_uint4(gq[0], gq[1], gq[2], gq[3]);
		*reinterpret_cast<uint4*>(lds + kSwAux + aNew * kSwRowBytes + lane * 16) = make_uint4(aux[0], aux[1], aux[2], aux[3]);
		}

		// ---- NMS + classification of row yo = yc - 1 on its candidates (pixels with g > tLow) ----
		const int rr = yc - 1 - y0; // tile row 0..63 for it = 4..67
		if (rr >= 0 && !(a.dbg & 4)) {
			__builtin_amdgcn_wave_barrier();
			// Candidate list of the row, one ballot per pixel slot: entry = byte offset of the candidate inside a ring row (2 * local
			// column).  The rank of a lane's pixel p is base_p (scalar popcounts of the earlier slots) + mbcnt of its own slot mask.
			// (Per slot: one compare, the two mbcnt halves and one add.  The list position is kept in half-words relative to the start
			// of the workgroup's LDS, wave base included, so the write address is just rank + rank.)
			const uint16_t* const list = reinterpret_cast<const uint16_t*>(lds + kSwList);
			const int rbase = (wave * kSwLdsBytes + kSwList) >> 1;
			int rtop = rbase;
			if (owner) {
#pragma unroll
				for (int p = 0; p < 8; ++p) {
					const uint32_t gp = (p & 1) ? (gprev[p >> 1] >> 16) : (gprev[p >> 1] & 0xffffu);
					const bool c = gp > (uint32_t)tLowQ;
					const uint64_t mk = __ballot(c);
					if (c) {
						const int r = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, (uint32_t)rtop));
						*reinterpret_cast<uint16_t*>(&lds_all[0][0] + twice(r)) = (uint16_t)(lane * 16 + 2 * p);
					}
					rtop += __popcll(mk);
				}
			}
			const int total = __builtin_amdgcn_readlane(rtop, 1) - rbase; // lane 1 is an owner lane (lane 0 skipped the block above)
			__builtin_amdgcn_wave_barrier();
			const uint8_t* const gU = lds + kSwG + sU * kSwRowBytes;
			const uint8_t* const gC = lds + kSwG + sC * kSwRowBytes;
			const uint8_t* const gD = lds + kSwG + sD * kSwRowBytes;
			const uint8_t* const ax_row = lds + kSwAux + aMid * kSwRowBytes;
			uint32_t* const mrow = maskw + (rr & 7) * 32;
			for (int base = 0; base < ((a.dbg & 2) ? 0 : total); base += 64) {
				const int j = base + lane;
				if (j < total) {
					const int e = list[j];
					// the centre, its aux word and all eight neighbours are fetched at once (one LDS round trip), the direction class then
					// picks one of four neighbour maxima
					const int gc = *reinterpret_cast<const uint16_t*>(gC + e);
					const uint32_t au = *reinterpret_cast<const uint16_t*>(ax_row + e);
					const int nL = *reinterpret_cast<const uint16_t*>(gC + e - 2), nR = *reinterpret_cast<const uint16_t*>(gC + e + 2);
					const int nU = *reinterpret_cast<const uint16_t*>(gU + e), nD = *reinterpret_cast<const uint16_t*>(gD + e);
					const int nUL = *reinterpret_cast<const uint16_t*>(gU + e - 2), nUR = *reinterpret_cast<const uint16_t*>(gU + e + 2);
					const int nDL = *reinterpret_cast<const uint16_t*>(gD + e - 2), nDR = *reinterpret_cast<const uint16_t*>(gD + e + 2);
					const uint32_t ax = au & 0x3ffu;
					const uint32_t ays = (uint32_t)(gc - 2048 - (int)ax) << 16;      // |gy| << 16
					// direction class (constants canny_dete.h:58-61: tan(pi/8), tan(3pi/8) in Q16; 158217 = 27145 + 2^17)
					const uint32_t t1 = __umul24(ax, 27145u);
					const bool k1 = ays < t1;
					const bool k2 = ays < t1 + (ax << 17);
					const bool ngd = (au & 0x400u) != 0;
					// neighbours along the gradient: k1 left/right; k2 diagonal ((gx^gy) < 0: (y+1,x-1),(y-1,x+1), else (y-1,x-1),(y+1,x+1)); else up/down
					const int mh = max(nL, nR), mv = max(nU, nD), md1 = max(nUL, nDR), md2 = max(nDL, nUR);
					const int md = ngd ? md2 : md1;
					const int mx2 = k1 ? mh : (k2 ? md : mv);
					bool weak = gc >= mx2;          // not suppressed: neither neighbour strictly greater (candidates already have g > tLow)
					bool strong = gc > tHighQ;
					const int c = e >> 1;           // local column 0..511
					if (GAP) { // quirk Q3: column coverage of the NMS and of the seed scan, [1,simdEnd) U [cStart,W-1) (canny_dete.cxx:396,514)
						const int x = xbase + c;
						const bool in_cov = (x >= 1 && x < a.simdEnd) || (x >= a.cStart && x < W - 1);
						weak = weak || !in_cov;       // outside the NMS coverage: thresholded only, never a seed
						strong = strong && in_cov;
					}
					const uint32_t bit = 1u << (c & 31);
					if (weak) {
						atomicOr(&mrow[c >> 5], bit);
						if (strong) atomicOr(&mrow[16 + (c >> 5)], bit);
					}
				}
			}
			// every 8 rows the mask ring is collected by the lanes that own those tile rows (lane == row) and cleared
			if ((rr & 7) == 7) {
				__builtin_amdgcn_wave_barrier();
				if ((lane >> 3) == (rr >> 3)) {
					const uint4* src = reinterpret_cast<const uint4*>(maskw + (lane & 7) * 32);
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const uint4 w = src[q], s4 = src[4 + q];
						Wm[2 * q] = (uint64_t)w.x | ((uint64_t)w.y << 32); Wm[2 * q + 1] = (uint64_t)w.z | ((uint64_t)w.w << 32);
						Em[2 * q] = (uint64_t)s4.x | ((uint64_t)s4.y << 32); Em[2 * q + 1] = (uint64_t)s4.z | ((uint64_t)s4.w << 32);
					}
				}
				__builtin_amdgcn_wave_barrier();
				*reinterpret_cast<uint4*>(lds + kSwMask + lane * 16) = make_uint4(0, 0, 0, 0);
			}
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) gprev[k] = gq[k];
	};

	{
		static_assert((kTileH + 4) % 6 == 2, "row loop is unrolled by 6 with a 2-step tail");
		int it = 0;
		for (; it < kTileH + 4 - 2; it += 6) {
			step(std::integral_constant<int, 0>{}, it);
			step(std::integral_constant<int, 1>{}, it + 1);
			step(std::integral_constant<int, 2>{}, it + 2);
			step(std::integral_constant<int, 3>{}, it + 3);
			step(std::integral_constant<int, 4>{}, it + 4);
			step(std::integral_constant<int, 5>{}, it + 5);
		}
		step(std::integral_constant<int, 0>{}, it);
		step(std::integral_constant<int, 1>{}, it + 1);
	}

	if (a.dbg & 8) { if (gprev[0] == 0x12345u) a.out[lane] = (uint8_t)P[0]; return; }
	// ---- lane == row: flood strong into weak inside the tile (512-bit carry chains + row exchange) ----
	auto flood_up = [&]() {
		uint64_t carry = 0;
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const uint64_t w = Wm[m];
			const uint64_t b = Em[m] & w;
			const uint64_t t = w + b;
			const uint64_t c1 = t < w;
			const uint64_t t2 = t + carry;
			const uint64_t c2 = t2 < t;
			carry = c1 | c2;
			Em[m] |= w & ~t2;
		}
	};
	auto flood_down = [&]() {
		uint64_t carry = 0;
#pragma unroll
		for (int m = 7; m >= 0; --m) {
			const uint64_t w = __brevll(Wm[m]);
			const uint64_t b = __brevll(Em[m]) & w;
			const uint64_t t = w + b;
			const uint64_t c1 = t < w;
			const uint64_t t2 = t + carry;
			const uint64_t c2 = t2 < t;
			carry = c1 | c2;
			Em[m] |= __brevll(w & ~t2);
		}
	};
#pragma unroll
	for (int m = 0; m < 8; ++m) Em[m] &= Wm[m]; // seeds are weak pixels by construction; keeps E a subset of W whatever happens
	for (;;) {
		flood_up();
		flood_down();
		bool changed = false;
		uint64_t nb[8];
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			uint64_t up = __shfl_up(Em[m], 1);
			uint64_t dn = __shfl_down(Em[m], 1);
			if (lane == 0) up = 0;
			if (lane == 63) dn = 0;
			nb[m] = up | dn;
		}
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			uint64_t n3 = nb[m] | (nb[m] << 1) | (nb[m] >> 1);
			if (m > 0) n3 |= nb[m - 1] >> 63;
			if (m < 7) n3 |= nb[m + 1] << 63;
			const uint64_t add = Wm[m] & n3 & ~Em[m];
			Em[m] |= add;
			changed |= (add != 0);
		}
		if (!__any(changed) || (a.dbg & 1)) break;
	}

	// ---- outputs: the two 1-bit masks (the edge BYTES are expanded from the final E mask by canny_expand_kernel, after the
	// cross-tile resolve, on a side stream next to the Hough stage: 1 B/px of stores that this kernel no longer waits for) ----
	// The masks leave through LDS so that every store instruction writes whole row segments: a lane == row
	// store of one dword per lane would touch 64 cache lines per instruction (measured: 0.14 ms of a 0.34 ms launch in L2 requests).
	// LDS image of a mask: 64 rows x 19 dwords: [0] = 0, [1..16] = the 16 local dwords, [17] = 0 (pitch 19: conflict-free).
	// Local bits 8..503 are global columns [496 t, 496 t + 496): the tile starts at half-word 31 t of the mask row, i.e. global dword
	// d0 + j = local bits [32 j + 8, 32 j + 40) for even tiles and [32 j - 8, 32 j + 24) for odd ones (d0 = floor(15.5 t)); the first
	// (odd) or last (even) dword of a tile is shared with the neighbouring tile and written as one half-word by each.
	constexpr int kTrPitch = 19;
	uint32_t* const tr = reinterpret_cast<uint32_t*>(lds);
	const int odd = tileX & 1;
	const int d0 = (31 * tileX) >> 1;
	const int rowsHere = min(kTileH, H - y0);
	auto put_mask = [&](const uint64_t (&M)[8]) {
		__builtin_amdgcn_wave_barrier();
		tr[lane * kTrPitch] = 0u; tr[lane * kTrPitch + 17] = 0u;
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			tr[lane * kTrPitch + 1 + 2 * m] = (uint32_t)M[m];
			tr[lane * kTrPitch + 2 + 2 * m] = (uint32_t)(M[m] >> 32);
		}
		__builtin_amdgcn_wave_barrier();
	};
	auto store_mask = [&](uint32_t* __restrict__ gm) {
		const int dl = lane & 15;
		const int gd = d0 + dl;
		const int src = 1 + dl - odd;        // LDS dword holding the low part of global dword gd
#pragma unroll 4
		for (int itr = 0; itr < 16; ++itr) {
			const int row = itr * 4 + (lane >> 4);
			const uint32_t lo = tr[row * kTrPitch + src], hi = tr[row * kTrPitch + src + 1];
			const uint32_t v = odd ? __builtin_amdgcn_alignbit(hi, lo, 24) : __builtin_amdgcn_alignbit(hi, lo, 8);
			if (row < rowsHere && gd < a.wb) {
				uint32_t* dst = gm + (size_t)(y0 + row) * a.wb + gd;
				if (!odd && dl == 15) reinterpret_cast<uint16_t*>(dst)[0] = (uint16_t)v;            // low half-word: the next tile owns the high one
				else if (odd && dl == 0) reinterpret_cast<uint16_t*>(dst)[1] = (uint16_t)(v >> 16);  // high half-word: the previous tile owns the low one
				else *dst = v;
			}
		}
	};
	uint32_t* const ebase = a.ebits + (size_t)frame * a.bitsFrameStride;
	uint32_t* const ubase = a.ubits + (size_t)frame * a.bitsFrameStride;
	put_mask(Em);
	store_mask(ebase);
#pragma unroll
	for (int m = 0; m < 8; ++m) Wm[m] &= ~Em[m]; // U = weak but not (yet) an edge
	put_mask(Wm);
	store_mask(ubase);
}

// ---------------------------------------------------------------------------------------------------------------
hipError_t launch_canny_tiles_swar(const CannyArgs& a0, int frames, bool gap, hipStream_t stream)
{
	CannyArgs a = a0;
	static const int dbg = [] { const char* e = getenv("COMPVHIP_CANNY_DBG"); return e ? atoi(e) : 0; }();
	a.dbg = dbg;
	a.tilesX = (a.W + kSwCols - 1) / kSwCols;
	a.blockRows = (a.tilesY + kCannyWaves - 1) / kCannyWaves;
	a.groups = a.blockRows * frames;
	dim3 grid(8 * ((a.groups + 7) / 8) * a.tilesX);
	dim3 block(kCannyWaves * 64);
	if (gap) hipLaunchKernelGGL((canny_swar_tile_kernel<true>), grid, block, 0, stream, a);
	else hipLaunchKernelGGL((canny_swar_tile_kernel<false>), grid, block, 0, stream, a);
	return hipGetLastError();
}

} // namespace compvhip
