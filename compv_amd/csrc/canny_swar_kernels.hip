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
// A wave owns 240 output columns x 64 rows, 4 pixels per lane: lanes 0, 1 and 62, 63 compute the gradient of the 8 columns either
// side of the tile (the NMS of columns 0 and 239 needs one of them; two lanes per side keep the tile's bit masks half-word aligned:
// 240 = 15 half-words) but own no pixels, so no gradient column is computed twice inside a wave.  Every input byte is still fetched
// once per tile (+ 4/64 row halo, + 16/240 column halo).  Why 4 pixels per lane and not 8: a gfx950 wave issues one instruction
// every ~8.7 cycles whatever its type (valu_rate_bench2, one wave per SIMD), so throughput = resident waves / 8.7 until a pipe
// saturates; 4 pixels per lane halve the registers and the LDS of a wave (<= 64 VGPRs, 3.4 KB) and put 8 waves on every SIMD.
#include "stencil.hpp"
#include "kernels.hpp"

#include <cstdlib>
#include <type_traits>

namespace compvhip {

namespace {

constexpr int kSwPx = 4;                     // pixels per lane
constexpr int kSwCols = 240;                 // output columns per wave tile (lanes 2..61)
constexpr int kSwRowBytes = 512;             // one ring row: 256 u16 in pixel order
constexpr int kSwG = 0;                      // byte offsets inside a wave's LDS block
constexpr int kSwAux = kSwG + 3 * kSwRowBytes;
constexpr int kSwList = kSwAux + 2 * kSwRowBytes;
constexpr int kSwMask = kSwList + 512;       // mask ring: [4 rows][weak | strong][11 dwords]: [0] = 0, [1..8] = 256 bits, [9] = 0, [10] unused
constexpr int kMaskPitch = 11;
constexpr int kSwLdsBytes = kSwMask + 4 * 2 * kMaskPitch * 4;   // 3424 B per wave
constexpr int kImgPitch = 9;                 // lane == row image of one 256-bit mask (tail of the kernel; reuses the ring space)
static_assert(64 * kMaskPitch * 4 <= kSwLdsBytes, "padded mask image must fit the wave's LDS block");

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
__device__ __forceinline__ uint32_t twice(uint32_t x)
{
	uint32_t d;
	asm("v_add_u32 %0, %1, %1" : "=v"(d) : "v"(x));
	return d;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
template <bool GAP>
__global__ __launch_bounds__(kCannyWaves * 64, 8) void canny_swar_tile_kernel(CannyArgs a)
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
	const int xbase = tileX * kSwCols - 8;          // column of local bit 0 (lanes 0, 1 = the left halo lanes)
	const int x0 = xbase + lane * kSwPx;
	const int y0 = tileY * kTileH;
	const uint8_t* __restrict__ in = a.in + (size_t)frame * a.inFrameStride;

	int tLow = a.tLow, tHigh = a.tHigh;
	if (a.thrDev) { const int2 t = a.thrDev[frame]; tLow = t.x; tHigh = t.y; }
	tLow = min(__builtin_amdgcn_readfirstlane(tLow), 4000);   // g <= 2040: larger thresholds select nothing
	tHigh = min(__builtin_amdgcn_readfirstlane(tHigh), 8000);
	const int tLowQ = tLow + 2048, tHighQ = tHigh + 2048;      // thresholds on g' = g + 2048

	// g is forced to 0 (g' = 2048) outside columns [1, W-2]: zero OUTPUT border of the convolution (compv_math_convlt.h:181-209)
	const bool edgeTile = (xbase < 1) || (xbase + 256 > W - 1);
	uint32_t okm[2] = { 0xffffffffu, 0xffffffffu };
	if (edgeTile) {
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const int xa = x0 + 2 * k, xb = xa + 1;
			okm[k] = ((xa >= 1 && xa <= W - 2) ? 0x0000ffffu : 0u) | ((xb >= 1 && xb <= W - 2) ? 0xffff0000u : 0u);
		}
	}
	// lanes 0, 1, 62, 63 own no pixels (column halo): their copy of g' for the candidate test is zeroed
	const bool owner = (lane >= 2 && lane <= 61);
	uint32_t ownv = owner ? 0xffffffffu : 0u;
	asm volatile("" : "+v"(ownv));
	// tiles whose gradient rows touch the image border rows (g forced to 0 there) or run past the image
	const bool vEdgeTile = (tileY == 0) || (y0 + kTileH + 1 >= H - 1);
	const bool borderTile = edgeTile || vEdgeTile;

	// mask geometry: local bits 8..247 are global columns [240 t, 240 t + 240): the tile starts at half-word 15 t of a mask row, i.e.
	// global dword d0 + j = local bits [32 j + 8, 32 j + 40) for even tiles and [32 j - 8, 32 j + 24) for odd ones (d0 = floor(7.5 t));
	// the first (odd) or last (even) dword of a tile is shared with the neighbouring tile and written as one half-word by each.
	const int odd = tileX & 1;
	const int d0 = (15 * tileX) >> 1;
	uint32_t* const ebase = a.ebits + (size_t)frame * a.bitsFrameStride;
	uint32_t* const ubase = a.ubits + (size_t)frame * a.bitsFrameStride;
	// one global dword (or its owned half-word) of a mask row from two adjacent LDS dwords [src], [src + 1] of a padded row image
	auto store_dword = [&](uint32_t* __restrict__ gm, int row, int dl, uint32_t lo, uint32_t hi) {
		const uint32_t v = odd ? __builtin_amdgcn_alignbit(hi, lo, 24) : __builtin_amdgcn_alignbit(hi, lo, 8);
		const int gd = d0 + dl;
		if (row < H && gd < a.wb) {
			uint32_t* dst = gm + (size_t)row * a.wb + gd;
			if (!odd && dl == 7) reinterpret_cast<uint16_t*>(dst)[0] = (uint16_t)v;             // low half-word: the next tile owns the high one
			else if (odd && dl == 0) reinterpret_cast<uint16_t*>(dst)[1] = (uint16_t)(v >> 16);  // high half-word: the previous tile owns the low one
			else *dst = v;
		}
	};

	// Row loads through a buffer descriptor of the frame: the row offset rides in an SGPR (soffset), the lane's column offset is a
	// loop-invariant VGPR -- no per-lane address arithmetic in the row loop.  Columns are clamped into the row (clamped lanes only
	// feed columns whose g is forced to 0), rows into the frame.
	const uint32_t xm = (uint32_t)min(max(x0, 0), S - 4);
	const uint32_t xl = (uint32_t)min(max(x0 - 4, 0), S - 4);
	const uint32_t xr = (uint32_t)min(max(x0 + 4, 0), S - 4);
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(in), 0, (int)((size_t)H * S), 0x00020000);
	auto load = [&](int y, uint32_t& m, uint32_t& l, uint32_t& r) {
		const int so = min(max(y, 0), H - 1) * S;
		m = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xm, so, 0);
		l = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xl, so, 0);
		r = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xr, so, 0);
	};

	// rolling state: two pixels per register, pairs k = (x0 + 2k, x0 + 2k + 1)
	uint32_t P[2] = { 0, 0 };              // d[y-2] + 2 d[y-1]   (bias 768)
	uint32_t dprev[2] = { 0, 0 };          // d[y-1]              (bias 256)
	uint32_t hy[2][2] = { { 0, 0 }, { 0, 0 } }; // horizontal smooth of rows y-1 / y-2 (ring)
	uint32_t gprev[2] = { 0, 0 };          // g' of the previous gradient row (owner lanes only): its candidates are processed this step

	uint32_t* const maskw = reinterpret_cast<uint32_t*>(lds + kSwMask);
	auto zero_ring = [&]() { if (lane < 22) *reinterpret_cast<uint4*>(lds + kSwMask + lane * 16) = make_uint4(0, 0, 0, 0); }; // 352 B
	zero_ring();

	uint32_t nm, nl, nr;
	load(y0 - 2, nm, nl, nr);

	// One row step: push input row yin = y0 - 2 + it  ->  gradient row yc = yin - 1  ->  (NMS) non-maximum suppression of row yo = yc - 1.
	auto step = [&](auto phase, auto with_nms, int it) {
		constexpr int PH = decltype(phase)::value;
		constexpr bool NMS = decltype(with_nms)::value;
		constexpr int sD = PH % 3, sC = (PH + 2) % 3, sU = (PH + 1) % 3; // ring slots of rows yo+1 (written now), yo, yo-1
		constexpr int aNew = PH % 2, aMid = (PH + 1) % 2;
		const int yin = y0 - 2 + it;
		const uint32_t m = nm, l = nl, r = nr;
		load(yin + 1, nm, nl, nr); // prefetch

		// ---- dense stage: packed pairs straight from the raw dwords (one v_perm each) ----
		uint32_t A[2], L[3];
		A[0] = __builtin_amdgcn_perm(0u, m, 0x0c010c00u);     // (p0, p1)
		A[1] = __builtin_amdgcn_perm(0u, m, 0x0c030c02u);     // (p2, p3)
		L[0] = __builtin_amdgcn_perm(m, l, 0x0c040c03u);      // (p-1, p0)
		L[1] = __builtin_amdgcn_perm(0u, m, 0x0c020c01u);     // (p1, p2)
		L[2] = __builtin_amdgcn_perm(r, m, 0x0c040c03u);      // (p3, p4)
		uint32_t gq[2], aux[2];
		uint32_t (&hyTop)[2] = hy[PH & 1]; // hy of row yin-2; overwritten with hy of row yin
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const uint32_t Lk = L[k], Rk = L[k + 1], Ck = A[k];
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
		if (borderTile) { // one wave-uniform test per row; interior tiles skip all of it
			const uint32_t rowm = (yc >= 1 && yc <= H - 2) ? 0xffffffffu : 0u; // image border rows (and rows past the image): g = 0
#pragma unroll
			for (int k = 0; k < 2; ++k) gq[k] = bfi(okm[k] & rowm, gq[k], kBias2k);
		}
		*reinterpret_cast<uint2*>(lds + kSwG + sD * kSwRowBytes + lane * 8) = make_uint2(gq[0], gq[1]);
		*reinterpret_cast<uint2*>(lds + kSwAux + aNew * kSwRowBytes + lane * 8) = make_uint2(aux[0], aux[1]);

		// ---- NMS + classification of row yo = yc - 1 on its candidates (pixels with g > tLow) ----
		if (NMS) {
			const int rr = yc - 1 - y0; // tile row 0..63
			__builtin_amdgcn_wave_barrier();
			// Candidate list of the row, one ballot per pixel slot: entry = byte offset of the candidate inside a ring row (2 * local
			// column), position = scalar popcount of the earlier slots (rides in as the mbcnt base) + mbcnt of the slot's own mask; the
			// store is exec-masked (scalar work: the kernel is bound by the VALU pipe, not by scalar issue).
			uint16_t* const list = reinterpret_cast<uint16_t*>(lds + kSwList);
			const uint32_t listB = (uint32_t)(wave * kSwLdsBytes + kSwList);                 // LDS byte address of the list
			uint32_t base2 = listB >> 1;                                                     // uniform: half-word index of the slot's first entry
#pragma unroll
			for (int p = 0; p < kSwPx; ++p) {
				const uint32_t gp = (p & 1) ? (gprev[p >> 1] >> 16) : (gprev[p >> 1] & 0xffffu);
				const bool c = gp > (uint32_t)tLowQ;
				const uint64_t mk = __ballot(c);
				if (c) {
					const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, base2));
					*reinterpret_cast<uint16_t*>(&lds_all[0][0] + twice(rk)) = (uint16_t)(lane * 8 + 2 * p);
				}
				base2 += (uint32_t)__popcll(mk);
			}
			const int total = (int)(__builtin_amdgcn_readfirstlane(base2) - (listB >> 1));
			__builtin_amdgcn_wave_barrier();
			const uint8_t* const gU = lds + kSwG + sU * kSwRowBytes;
			const uint8_t* const gC = lds + kSwG + sC * kSwRowBytes;
			const uint8_t* const gD = lds + kSwG + sD * kSwRowBytes;
			const uint8_t* const ax_row = lds + kSwAux + aMid * kSwRowBytes;
			uint32_t* const mrowW = maskw + ((rr & 3) * 2) * kMaskPitch + 1;
			uint32_t* const mrowS = mrowW + kMaskPitch;
#pragma nounroll
			for (int base = 0; base < total; base += 64) {
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
					const int c = e >> 1;           // local column 0..255
					if (GAP) { // quirk Q3: column coverage of the NMS and of the seed scan, [1,simdEnd) U [cStart,W-1) (canny_dete.cxx:396,514)
						const int x = xbase + c;
						const bool in_cov = (x >= 1 && x < a.simdEnd) || (x >= a.cStart && x < W - 1);
						weak = weak || !in_cov;       // outside the NMS coverage: thresholded only, never a seed
						strong = strong && in_cov;
					}
					const uint32_t bit = 1u << (c & 31);
					if (weak) {
						atomicOr(&mrowW[c >> 5], bit);
						if (strong) atomicOr(&mrowS[c >> 5], bit);
					}
				}
			}
			// every 4 rows the ring goes to the global masks in their final layout (weak -> U buffer, strong -> E buffer), whole row
			// segments per store instruction (lanes 0..31 the weak rows, 32..63 the strong rows); the tail of the kernel reads them back
			if ((rr & 3) == 3) {
				__builtin_amdgcn_wave_barrier();
				const int mi = lane >> 5, q = (lane >> 3) & 3, dl = lane & 7;
				const uint32_t* rw = maskw + (q * 2 + mi) * kMaskPitch + 1 + dl - odd;
				store_dword(mi ? ebase : ubase, y0 + rr - 3 + q, dl, rw[0], rw[1]);
				__builtin_amdgcn_wave_barrier();
				zero_ring();
			}
		}
#pragma unroll
		for (int k = 0; k < 2; ++k) gprev[k] = gq[k] & ownv;
	};

	{
		using T = std::true_type; using F = std::false_type;
		step(std::integral_constant<int, 0>{}, F{}, 0);
		step(std::integral_constant<int, 1>{}, F{}, 1);
		step(std::integral_constant<int, 2>{}, F{}, 2);
		step(std::integral_constant<int, 3>{}, F{}, 3);
		int it = 4;
		for (; it < kTileH; it += 6) { // it = 4 .. 63: ten trips of six steps
			step(std::integral_constant<int, 4>{}, T{}, it);
			step(std::integral_constant<int, 5>{}, T{}, it + 1);
			step(std::integral_constant<int, 0>{}, T{}, it + 2);
			step(std::integral_constant<int, 1>{}, T{}, it + 3);
			step(std::integral_constant<int, 2>{}, T{}, it + 4);
			step(std::integral_constant<int, 3>{}, T{}, it + 5);
		}
		static_assert(kTileH % 6 == 4, "main loop ends at it = 64");
		step(std::integral_constant<int, 4>{}, T{}, it);
		step(std::integral_constant<int, 5>{}, T{}, it + 1);
		step(std::integral_constant<int, 0>{}, T{}, it + 2);
		step(std::integral_constant<int, 1>{}, T{}, it + 3);
	}

	// ---- lane == row: read the weak / strong masks back (coalesced, through an LDS image), flood strong into weak ----
	uint32_t* const img = reinterpret_cast<uint32_t*>(lds);
	const int rowsHere = min(kTileH, H - y0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's mask stores have reached the L2
	auto fetch_mask = [&](const uint32_t* gm, uint64_t (&M)[4]) {
		__builtin_amdgcn_wave_barrier();
		const int k = lane & 7;
		const int a0 = odd ? k : k - 1;     // local dword k = bits of global dwords d0 + a0, d0 + a0 + 1
#pragma unroll 4
		for (int itr = 0; itr < 8; ++itr) {
			const int row = itr * 8 + (lane >> 3);
			uint32_t g0 = 0, g1 = 0;
			if (row < rowsHere) {
				const uint32_t* src = gm + (size_t)(y0 + row) * a.wb + d0;
				// relaxed agent-scope loads (sc1): served by the L2, never by a stale line of this CU's vector cache
				if (a0 >= 0 && d0 + a0 < a.wb) g0 = __hip_atomic_load(src + a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (a0 + 1 <= 7 && d0 + a0 + 1 < a.wb) g1 = __hip_atomic_load(src + a0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			uint32_t v = odd ? __builtin_amdgcn_alignbit(g1, g0, 8) : __builtin_amdgcn_alignbit(g1, g0, 24);
			if (k == 0) v &= 0xffffff00u;        // local bits 0..7 and 248..255 are the neighbouring tiles' columns
			if (k == 7) v &= 0x00ffffffu;
			img[row * kImgPitch + k] = v;
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int m = 0; m < 4; ++m) M[m] = (uint64_t)img[lane * kImgPitch + 2 * m] | ((uint64_t)img[lane * kImgPitch + 2 * m + 1] << 32);
	};
	uint64_t Wm[4], Em[4];
	fetch_mask(ubase, Wm);
	fetch_mask(ebase, Em);

	auto flood_up = [&]() {
		uint64_t carry = 0;
#pragma unroll
		for (int m = 0; m < 4; ++m) {
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
		for (int m = 3; m >= 0; --m) {
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
	for (int m = 0; m < 4; ++m) Em[m] &= Wm[m]; // seeds are weak pixels by construction; keeps E a subset of W whatever happens
	for (;;) {
		flood_up();
		flood_down();
		bool changed = false;
		uint64_t nb[4];
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			uint64_t up = __shfl_up(Em[m], 1);
			uint64_t dn = __shfl_down(Em[m], 1);
			if (lane == 0) up = 0;
			if (lane == 63) dn = 0;
			nb[m] = up | dn;
		}
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			uint64_t n3 = nb[m] | (nb[m] << 1) | (nb[m] >> 1);
			if (m > 0) n3 |= nb[m - 1] >> 63;
			if (m < 3) n3 |= nb[m + 1] << 63;
			const uint64_t add = Wm[m] & n3 & ~Em[m];
			Em[m] |= add;
			changed |= (add != 0);
		}
		if (!__any(changed)) break;
	}

	// ---- outputs: the two 1-bit masks in their final state: E (edges so far) and U (weak pixels this tile could not resolve).
	// The edge BYTES are expanded from the final E mask by canny_expand_kernel, after the cross-tile resolve, on a side stream next
	// to the Hough stage: 1 B/px of stores that this kernel no longer waits for.  The masks leave through an LDS image (64 rows x
	// 11 dwords: [0] = 0, [1..8] = the local dwords, [9] = 0) so that every store instruction writes whole row segments: a
	// lane == row store of one dword per lane touches 64 cache lines per instruction (measured: 0.14 ms of a 0.34 ms launch).
	uint32_t* const tr = reinterpret_cast<uint32_t*>(lds);
	auto put_store = [&](const uint64_t (&M)[4], uint32_t* __restrict__ gm) {
		__builtin_amdgcn_wave_barrier();
		tr[lane * kMaskPitch] = 0u; tr[lane * kMaskPitch + 9] = 0u;
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			tr[lane * kMaskPitch + 1 + 2 * m] = (uint32_t)M[m];
			tr[lane * kMaskPitch + 2 + 2 * m] = (uint32_t)(M[m] >> 32);
		}
		__builtin_amdgcn_wave_barrier();
		const int dl = lane & 7;
		const int src = 1 + dl - odd;        // LDS dword holding the low part of global dword d0 + dl
#pragma unroll 4
		for (int itr = 0; itr < 8; ++itr) {
			const int row = itr * 8 + (lane >> 3);
			store_dword(gm, y0 + row, dl, tr[row * kMaskPitch + src], tr[row * kMaskPitch + src + 1]);
		}
	};
	put_store(Em, ebase);
#pragma unroll
	for (int m = 0; m < 4; ++m) Wm[m] &= ~Em[m]; // U = weak but not (yet) an edge
	put_store(Wm, ubase);
}

// ---------------------------------------------------------------------------------------------------------------
hipError_t launch_canny_tiles_swar(const CannyArgs& a0, int frames, bool gap, hipStream_t stream)
{
	CannyArgs a = a0;
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
