// canny_swar_kernels.hip -- fused Sobel (3x3 or 5x5) -> NMS -> weak / strong classification for gfx950: the Canny tile kernel.  The text below describes the
// 3x3 instance; what differs for 5x5 is said at the kernel's template.
//
// Replaces (behind compvhip_canny_u8 / compvhip_plan_canny / compvhip_plan_pipeline):
//   CompVEdgeDeteCanny::process            core/features/edges/compv_core_feature_canny_dete.cxx:123-331
//   nms_gather / nms_apply                 ...canny_dete.cxx:334-462, row leaves :566-616,
//   CompVCannyNMSGatherRow_16mpw_Intrin_AVX2  core/features/edges/intrin/x86/compv_core_feature_canny_dete_intrin_avx2.cxx:150-235
// (the hysteresis, :464-528, is canny_resolve_kernel's: this kernel hands it the seeds E = strong and the candidates U = weak & ~strong)
//
// The kernel is bound by VALU issue, not by HBM (tools/microbench/valu_rate_bench2/3: ~2.3 cycles per wave64
// instruction for v_add/sub/and/or/xor/lshr with VGPR or literal operands, ~4.3 for everything else), so its structure follows the
// instruction count:
//   * dense stage: gradient magnitude of every pixel, two pixels per 32-bit register (SWAR on u16 halves).  Sums and differences of
//     8-bit pixels never carry across the halves once the signed terms are biased (d + 256, gx + 1024, gy + 1024), so plain
//     v_add_u32 / v_sub_u32 / v_or_b32 do packed arithmetic; |.| is one v_pk_max_u16 of (v, 2*bias - v).  g' = g + 2048 (u16) and an
//     aux word (|gx|, sign(gx ^ gy)) of every pixel go to small LDS rings in pixel order;
//   * sparse stage: only pixels with g > tLow (~10 % on the benchmark frames) can survive the NMS.  They are compacted into a
//     candidate list (one v_cmp = one ballot per pixel slot, mbcnt ranks) that collects TWO rows before it is processed, so that the
//     64 lanes of the NMS pass are ~80 % occupied (a one-row list filled 24 of 64 lanes: round 2's measurement); one candidate per lane
//     evaluates the reference's direction class (Q16 constants 27145 / 158217) and compares with the two neighbours along the
//     gradient, read from the ring;
//   * results are OR-ed into 256-bit row masks (weak / strong) in LDS; every 4 rows the masks leave in their final global layout
//     (E = strong, U = weak & ~strong) together with the edge BYTES of the strong pixels (16-byte stores).  There is no in-tile flood
//     any more: the 512-bit carry-chain flood of the first two generations cost 20 % of the kernel in lock step on every tile, the
//     band kernel does the same work on packed words, only where there is something to do (canny_kernels.hip, canny_resolve_kernel).
// Round 4 (tools/canny_lab: variants of this kernel compared bit for bit, timed interleaved): with everything above in place the kernel sat on a STORE floor, not on
// the VALU -- the list and the NMS cost nothing while the stores were there.  The 16 edge bytes of a lane leave as ONE 16-byte store (the compiler had split them 12 + 4),
// input rows are fetched two steps ahead, the dense stage uses three-operand forms (v_xad / v_lshl_add / v_add3: 31 instead of 39 instructions per row).  What is left of the
// floor are the 2-byte stores of the mask dword two tiles share (sub-dword stores cost far more than their bytes here): DESIGN.md section 4.1.
// Round 5 built the obvious alternative to the halo lanes -- 256 OWNED columns per wave, the gradient of the two columns either side of the tile from a
// per-tile prologue (lane = row, both sides packed in one register), whole mask dwords and 256-byte aligned edge rows -- bit-exact and 5 % SLOWER
// (profiles/r05/canny_lab_swar256.txt, commit 7eabbb1): the prologue's 144 single-dword gather loads per tile cost 14 % of the kernel, more than the
// halo lanes (6 %) and the 2-byte stores (8 %) together.  The halo lanes stay.
// Round 6 (VERDICT r5 #3 i): kernel size 3 loads every input dword ONCE -- one buffer load per lane and row, three rows ahead -- and takes the byte either side
// of a lane's four pixels from the neighbour lanes' dwords through a two-row exchange buffer in the LDS (ds_write_b32 + 2 ds_read_u8 per row, one step before
// use): a third of the vector-memory requests, bit-exact, 0.989 of the three-loads kernel on 32 x 4K frames, 1.017 on 32 x 1080p -- the launcher takes it for
// launches of at least four rounds of the chip's wave slots (the bound of ANY load-path change -- neighbours faked -- is 0.948: the kernel is bound by VALU issue,
// profiles/r06/canny_lab_r06.txt).
// A wave owns 240 output columns x kSwRows rows, 4 pixels per lane: lanes 0, 1 and 62, 63 compute the gradient of the 8 columns either
// side of the tile (the NMS of columns 0 and 239 needs one of them; two lanes per side keep the tile's bit masks half-word aligned:
// 240 = 15 half-words) but own no pixels.  Every input byte is fetched once per tile (+ 4/64 row halo, + 16/240 column halo).
// 4 pixels per lane: <= 64 VGPRs and 4.9 KB of LDS per wave put 8 waves on every SIMD (one wave issues an instruction only every
// ~8.7 cycles whatever its type, so the issue rate of a SIMD is resident waves / 8.7 until a pipe saturates).
#include "stencil.hpp"
#include "kernels.hpp"

#include <cstdlib>
#include <type_traits>

// tools/canny_lab builds this file several times (one namespace and one set of SWAR_* switches per variant) and runs the variants side by side,
// comparing outputs bit for bit; the product build defines none of them.
#ifndef COMPVHIP_SWAR_NS
#define COMPVHIP_SWAR_NS compvhip
#endif
#ifndef SWAR_ROWS
#define SWAR_ROWS 24
#endif

namespace COMPVHIP_SWAR_NS {
using namespace compvhip;

namespace {

constexpr int kSwPx = 4;                     // pixels per lane
constexpr int kSwCols = 240;                 // output columns per wave tile (lanes 2..61)
constexpr int kRowB = 512;                   // one ring row: 256 u16 in pixel order
// LDS of one wave (byte offsets; the g ring is 2048-aligned so that "row above / below" wraps with one AND):
// [0, 2048): g' ring, 4 rows: row r lives in slot r & 3
constexpr int kAux = 4 * kRowB;              // aux ring, 2 rows: row r lives in slot r & 1
constexpr int kList = kAux + 2 * kRowB;      // candidate list of a row pair: <= 480 u16 entries
constexpr int kNib = kList + 1024;           // result nibbles, 4 rows x 64 bytes: byte l of a row = lane l's four pixels, U flags (weak, not strong) in bits 0..3, E flags (strong) in 4..7
constexpr int kNibRowB = 64;
// kernel size 3 (round 6): the neighbour lanes' pixels travel through the LDS -- a wave loads every input dword ONCE and publishes it in a two-row exchange
// buffer (256 B per row); kernel size 5 keeps its three loads per row and does not touch the buffer
constexpr int kXch = kNib + 4 * kNibRowB + 64;         // (the flush reads up to 8 bytes past the last nibble row)
constexpr int kLdsBytes = kXch + 2 * 256 + 8;          // 4936 B per wave: 32 waves x 5120 (512-byte granules) = the CU's 160 KB

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
// three-operand forms (one instruction each; the constants ride in SGPRs -- VOP3 takes no literal on gfx9)
__device__ __forceinline__ uint32_t lshl1_add(uint32_t a, uint32_t b) { uint32_t d; asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }   // 2a + b
__device__ __forceinline__ uint32_t xad(uint32_t a, uint32_t sk, uint32_t c) { uint32_t d; asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sk), "v"(c)); return d; }   // (a ^ k) + c
__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t sk) { uint32_t d; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(sk)); return d; }
// nibble -> four bytes {0, 0xff}
__device__ __forceinline__ uint32_t nibble_bytes(uint32_t nib)
{
	const uint32_t b = __umul24(nib, 0x00204081u) & 0x01010101u;   // bit j -> byte j
	uint32_t hi;
	asm("v_lshlrev_b32 %0, 8, %1" : "=v"(hi) : "v"(b));           // 255 b = (b << 8) - b (spelled out: the compiler otherwise emits a full 32-bit multiply)
	return hi - b;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// KS = 3: the 3x3 Sobel (compv_features.h:124-127).  KS = 5: the 5x5 Sobel (vt {1,4,6,4,1}, hz {1,2,0,-2,-1}: compv_features.h:129-130) in the same structure --
// round 5; rounds 1-4 ran kernel size 5 on the first-generation kernel (32-bit arithmetic, 8 px per lane, dense NMS, in-tile flood: 0.50 ms per 32 x 4K).
// The packed arithmetic carries over because the 5x5 gradient is linear and small enough for biased u16 halves: per input row the horizontal derivative
// H1 = I[x-2] + 2 I[x-1] - 2 I[x+1] - I[x+2] (+1024) and smooth H2 = I[x-2] + 4 I[x-1] + 6 I[x] + 4 I[x+1] + I[x+2] (<= 4080) of a pixel pair, kept for four
// rows in registers (ring indexed by the row loop's phase); gx = (1,4,6,4,1) . H1 down the column (+16384), gy = H2[y-2] + 2 H2[y-1] - 2 H2[y+1] - H2[y+2]
// (+16383), |.| by v_pk_max_u16 as before, g' = g + 32768 (g <= 24480), aux = |gx| in bits 0..13 and the sign flag in bit 14 -- the same conventions as
// the 3x3 path with wider fields, so the candidate list, the NMS, the flush and the hysteresis hand-over are shared.
// XCH: the LDS exchange path of the input rows (kernel size 3 only; chosen by the launcher for launches that oversubscribe the wave slots).
template <bool GAP, int WAVES, int kSwRows, int KS, bool XCH>
__global__ __launch_bounds__(WAVES * 64, (KS == 3 ? 8 : 7)) void canny_swar_tile_kernel(CannyArgs a)
{
	static_assert(!XCH || KS == 3, "the exchange path is the 3x3 kernel's");
	static_assert(KS == 3 || KS == 5, "Sobel kernel size");
	constexpr int R = KS / 2;                                  // width of the zero OUTPUT border of the gradient
	constexpr uint32_t kG = (KS == 3) ? 2048u : 32768u;         // bias of g' (per half)
	constexpr uint32_t kGpk = kG | (kG << 16);
	constexpr uint32_t kAxMask = (KS == 3) ? 0x3ffu : 0x3fffu;  // |gx| field of the aux word; the sign flag is the bit above it
	constexpr uint32_t kFlag = kAxMask + 1u;
	constexpr uint32_t kFlagPk = kFlag | (kFlag << 16);
	// the g rings of all waves first (2048-byte aligned), then the rest of each wave's block
	__shared__ __attribute__((aligned(2048))) uint8_t lds_all[WAVES * kLdsBytes];

	const int lane = threadIdx.x & 63;
	const int wave = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// the step's counters (consumed from the first hysteresis round on; the previous step's readers are behind us on the stream): 64 per workgroup
	if (a.zero) for (int i = (int)blockIdx.x * (WAVES * 64) + (int)threadIdx.x; i < a.nZero; i += (int)gridDim.x * (WAVES * 64)) a.zero[i] = 0;   // (one trip, in the first workgroups, on any real launch)
	int tileX, group;
	if (!xcd_tile_map(blockIdx.x, a.tilesX, a.groups, tileX, group)) return;
	const int frame = group / a.blockRows;
	const int tileY = (group - frame * a.blockRows) * WAVES + wave;
	if (tileY >= a.tilesY) return; // whole wave

	uint8_t* const ring = lds_all + wave * (4 * kRowB);                               // g' ring
	uint8_t* const rest = lds_all + WAVES * (4 * kRowB) + wave * (kLdsBytes - 4 * kRowB) - kAux; // rest[kAux ..] = aux, list, masks
	const int W = a.W, H = a.H, S = a.S;
	const int xbase = tileX * kSwCols - 8;          // column of local bit 0 (lanes 0, 1 = the left halo lanes)
	const int x0 = xbase + lane * kSwPx;
	const int y0 = tileY * kSwRows;
	const uint8_t* __restrict__ in = a.in + (size_t)frame * a.inFrameStride;

	int tLow = a.tLow, tHigh = a.tHigh;
	if (a.thrDev) { const int2 t = a.thrDev[frame]; tLow = t.x; tHigh = t.y; }
	tLow = min(__builtin_amdgcn_readfirstlane(tLow), KS == 3 ? 4000 : 32767);   // g <= 2040 (24480): larger thresholds select nothing
	tHigh = min(__builtin_amdgcn_readfirstlane(tHigh), KS == 3 ? 8000 : 32767);
	const int tLowQ = tLow + (int)kG, tHighQ = tHigh + (int)kG;                  // thresholds on g' = g + bias

	// g is forced to 0 (g' = bias) outside columns [R, W-1-R]: zero OUTPUT border of the convolution (compv_math_convlt.h:181-209)
	const bool edgeTile = (xbase < R) || (xbase + 256 > W - R);
	uint32_t okm[2] = { 0xffffffffu, 0xffffffffu };
	if (edgeTile) {
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const int xa = x0 + 2 * k, xb = xa + 1;
			okm[k] = ((xa >= R && xa <= W - 1 - R) ? 0x0000ffffu : 0u) | ((xb >= R && xb <= W - 1 - R) ? 0xffff0000u : 0u);
		}
	}
	// lanes 0, 1, 62, 63 own no pixels (column halo): their candidate threshold is out of reach
	const bool owner = (lane >= 2 && lane <= 61);
	uint32_t thrV = owner ? (uint32_t)tLowQ : 0xffffu;
	asm volatile("" : "+v"(thrV));
	// tiles whose gradient rows touch the image border rows (g forced to 0 there) or run past the image
	const bool vEdgeTile = (tileY == 0) || (y0 + kSwRows + 1 >= H - R);
	const bool borderTile = edgeTile || vEdgeTile;

	// mask geometry: local bits 8..247 are global columns [240 t, 240 t + 240): the tile starts at half-word 15 t of a mask row, i.e.
	// global dword d0 + j = local bits [32 j + 8, 32 j + 40) for even tiles and [32 j - 8, 32 j + 24) for odd ones (d0 = floor(7.5 t));
	// the first (odd) or last (even) dword of a tile is shared with the neighbouring tile and written as one half-word by each.
	const int odd = tileX & 1;
	const int d0 = (15 * tileX) >> 1;
	uint32_t* const ebase = a.ebits + (size_t)frame * a.bitsFrameStride;
	uint32_t* const ubase = a.ubits + (size_t)frame * a.bitsFrameStride;
	uint8_t* const obase = a.out + (size_t)frame * a.outFrameStride;

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
	uint32_t P[2] = { 0, 0 };              // 3x3: d[y-2] + 2 d[y-1]   (bias 765)
	uint32_t dprev[2] = { 0, 0 };          //      d[y-1]              (bias 255)
	uint32_t hy[2][2] = { { 0, 0 }, { 0, 0 } }; //  horizontal smooth of rows y-1 / y-2 (ring)
	uint32_t h1[4][2] = {}, h2[4][2] = {}; // 5x5: H1 + 1024 and H2 of the last four input rows (ring: the row pushed at step it sits in slot it & 3)

	// per-lane constants of the sparse stage
	const uint32_t lane8 = (uint32_t)lane * 8u;            // byte offset of the lane's 4 u16 inside a ring row
	const bool fullRow = (tileX * kSwCols + kSwCols <= a.So); // every 16-byte group of the tile's edge bytes lies inside the output row
	auto zero_nibbles = [&]() { *reinterpret_cast<uint32_t*>(rest + kNib + lane * 4) = 0u; };   // 4 rows x 64 B
	zero_nibbles();
	uint32_t listCount = 0;                // entries in the candidate list (wave-uniform)

	uint32_t k255 = 0x00ff00ffu, k4 = 0x00040004u, k3ff = 0x03ff03ffu, k1 = 0x00010001u, k3fff = 0x3fff3fffu;
	asm volatile("" : "+s"(k255), "+s"(k4), "+s"(k3ff), "+s"(k1), "+s"(k3fff));
	// 5x5, per input row and pixel pair: H1 + 1024 = (I[x-2] + 2 I[x-1]) + 1023 - (2 I[x+1] + I[x+2]) + 1 and H2 = (I[x-2] + I[x+2]) + 4 (I[x-1] + I[x] + I[x+1]) + 2 I[x]
	// from the five pairs at distances -2 .. +2 (every sum stays below 2^16: no carry between the halves)
	auto hz5 = [&](uint32_t m2, uint32_t m1, uint32_t c0, uint32_t p1, uint32_t p2, uint32_t& H1, uint32_t& H2) {
		const uint32_t pl = lshl1_add(m1, m2);                 // I[x-2] + 2 I[x-1]      (<= 765)
		const uint32_t pr = lshl1_add(p1, p2);                 // I[x+2] + 2 I[x+1]
		H1 = add3(pl, pr ^ 0x03ff03ffu, k1);                    // pl + (1023 - pr) + 1
		const uint32_t t = (m1 + p1) + c0;
		uint32_t u; asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(u) : "v"(t), "v"(m2 + p2));   // 4 t + I[x-2] + I[x+2]
		H2 = lshl1_add(c0, u);
	};

	// input rows are fetched TWO steps ahead (two register sets that alternate: the row loop is unrolled by four).  With one row in flight per
	// wave the kernel ran 12 % slower once its stores shared the memory pipeline with the loads (tools/canny_lab, round 4)
	constexpr int E = (KS == 5) ? 1 : 0;   // step it takes input row y0 - 2 + E + it and yields gradient row y0 + it - 3
	uint32_t nb[2][3];
	constexpr bool kXchLR = XCH;
	// kXchLR: ONE load per lane and input row, three rows ahead (ring mq); the byte either side of a lane's four pixels comes from the neighbour lanes'
	// dwords through a two-row exchange buffer in the LDS, fetched one step before the row is used (lanes 0 / 63 read a byte beside the buffer: their
	// outermost columns feed nothing)
	uint32_t mq[4] = { 0, 0, 0, 0 }, lrq[2][2] = { { 0, 0 }, { 0, 0 } };
	auto loadm = [&](int y, uint32_t& m) {
		const int so = min(max(y, 0), H - 1) * S;
		m = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)xm, so, 0);
	};
	auto exchange = [&](int slot, uint32_t m, uint32_t& lb, uint32_t& rb) {
		uint8_t* const xb = rest + kXch + slot * 256 + lane * 4;
		*reinterpret_cast<uint32_t*>(xb) = m;
		__builtin_amdgcn_wave_barrier();
		lb = *(xb - 1);
		rb = *(xb + 4);
		__builtin_amdgcn_wave_barrier();
	};
	if constexpr (KS == 5) {
		// two more input rows above the tile than the 3x3 path streams: y0 - 3 and y0 - 2 only feed the row ring (slots 2 and 3: steps "-2" and "-1")
		uint32_t w[2][3];
		load(y0 - 3, w[0][0], w[0][1], w[0][2]);
		load(y0 - 2, w[1][0], w[1][1], w[1][2]);
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const uint32_t m = w[q][0], l = w[q][1], r = w[q][2];
			const uint32_t a0 = __builtin_amdgcn_perm(0u, m, 0x0c010c00u), a1 = __builtin_amdgcn_perm(0u, m, 0x0c030c02u);     // (p0,p1) (p2,p3)
			const uint32_t am = __builtin_amdgcn_perm(0u, l, 0x0c030c02u), ap = __builtin_amdgcn_perm(0u, r, 0x0c010c00u);     // (p-2,p-1) (p4,p5)
			const uint32_t o0 = __builtin_amdgcn_perm(m, l, 0x0c040c03u), o1 = __builtin_amdgcn_perm(0u, m, 0x0c020c01u), o2 = __builtin_amdgcn_perm(r, m, 0x0c040c03u);   // (p-1,p0) (p1,p2) (p3,p4)
			hz5(am, o0, a0, o1, a1, h1[2 + q][0], h2[2 + q][0]);
			hz5(a0, o1, a1, o2, ap, h1[2 + q][1], h2[2 + q][1]);
		}
	}
	if constexpr (kXchLR) {
		loadm(y0 - 2, mq[0]); loadm(y0 - 1, mq[1]); loadm(y0, mq[2]);
		exchange(0, mq[0], lrq[0][0], lrq[0][1]);
	}
	else {
		load(y0 - 2 + E, nb[0][0], nb[0][1], nb[0][2]);
		load(y0 - 1 + E, nb[1][0], nb[1][1], nb[1][2]);
	}

	// One row step: push input row yin = y0 - 2 + it  ->  gradient row yc = yin - 1 = y0 + (it - 3)  ->  ring slot (it - 3) & 3.
	// After the steps with odd it >= 5 the rows 2j, 2j + 1 (j = (it - 5) / 2) of the tile have all three g rows of their neighbourhood
	// in the ring and their candidates in the list: NMS of the pair.  After every second pair the four result rows are flushed.
	auto step = [&](auto phase, auto with_nms, int it) {
		constexpr int PH = decltype(phase)::value;            // it & 3
		constexpr bool NMS = decltype(with_nms)::value;
		constexpr int sNew = (PH + 1) & 3;                    // ring slot of the gradient row produced now
		constexpr int aNew = (PH + 1) & 1;                    // its aux slot = its row parity
		const int yin = y0 - 2 + E + it;
		uint32_t m, l, r;
		if constexpr (kXchLR) {
			m = mq[PH]; l = lrq[PH & 1][0]; r = lrq[PH & 1][1];               // l, r: ONE byte each here (p-1, p4), zero-extended
			loadm(yin + 3, mq[(PH + 3) & 3]);                                  // prefetch, three rows ahead
			exchange((PH + 1) & 1, mq[(PH + 1) & 3], lrq[(PH + 1) & 1][0], lrq[(PH + 1) & 1][1]);   // the next row's neighbours
		}
		else {
			m = nb[PH & 1][0]; l = nb[PH & 1][1]; r = nb[PH & 1][2];
			load(yin + 2, nb[PH & 1][0], nb[PH & 1][1], nb[PH & 1][2]); // prefetch
		}

		// ---- dense stage: packed pairs straight from the raw dwords (one v_perm each) ----
		uint32_t A[2], L[3];
		A[0] = __builtin_amdgcn_perm(0u, m, 0x0c010c00u);     // (p0, p1)
		A[1] = __builtin_amdgcn_perm(0u, m, 0x0c030c02u);     // (p2, p3)
		L[0] = __builtin_amdgcn_perm(m, l, kXchLR ? 0x0c040c00u : 0x0c040c03u);      // (p-1, p0)
		L[1] = __builtin_amdgcn_perm(0u, m, 0x0c020c01u);     // (p1, p2)
		L[2] = __builtin_amdgcn_perm(r, m, 0x0c040c03u);      // (p3, p4)
		uint32_t gq[2], aux[2];
		uint32_t (&hyTop)[2] = hy[PH & 1]; // hy of row yin-2; overwritten with hy of row yin
		if constexpr (KS == 5) {
			const uint32_t am = __builtin_amdgcn_perm(0u, l, 0x0c030c02u), ap = __builtin_amdgcn_perm(0u, r, 0x0c010c00u);   // (p-2, p-1), (p4, p5)
			// the row pushed now goes to ring slot PH (= it & 3), which held input row yin - 4; the rows yin - 3, yin - 2, yin - 1 sit in slots PH + 1, + 2, + 3
			constexpr int s4 = PH & 3, s3 = (PH + 1) & 3, s2 = (PH + 2) & 3, s1 = (PH + 3) & 3;
			uint32_t n1[2], n2[2];
			hz5(am, L[0], A[0], L[1], A[1], n1[0], n2[0]);
			hz5(A[0], L[1], A[1], L[2], ap, n1[1], n2[1]);
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				// gx + 16384 = (1, 4, 6, 4, 1) . (H1 + 1024) down rows yin - 4 .. yin;  gy + 16383 = (H2[yin-4] + 2 H2[yin-3]) + 16383 - (2 H2[yin-1] + H2[yin])
				const uint32_t t = (h1[s3][k] + h1[s1][k]) + h1[s2][k];
				uint32_t u; asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(u) : "v"(t), "v"(h1[s4][k] + n1[k]));
				const uint32_t gxb = lshl1_add(h1[s2][k], u);
				const uint32_t ptop = lshl1_add(h2[s3][k], h2[s4][k]), pbot = lshl1_add(h2[s1][k], n2[k]);
				const uint32_t gyb = xad(pbot, k3fff, ptop);
				h1[s4][k] = n1[k]; h2[s4][k] = n2[k];
				const uint32_t mx = pk_max_u16(gxb, 0x80008000u - gxb);    // |gx| + 16384
				const uint32_t my = pk_max_u16(gyb, 0x7ffe7ffeu - gyb);    // |gy| + 16383
				gq[k] = add3(mx, my, k1);                                  // g + 32768
				aux[k] = bfi(kFlagPk, gxb ^ gyb, mx);                      // bits 0..13 |gx|, bit 14 = ((gx ^ gy) < 0), gy = 0 reading as negative (see below)
			}
			(void)hyTop;
		}
		else {
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const uint32_t Lk = L[k], Rk = L[k + 1], Ck = A[k];
			const uint32_t hyN = lshl1_add(Ck, Lk + Rk);               // I[x-1] + 2 I[x] + I[x+1]           (<= 1020)
			const uint32_t d = xad(Lk, k255, Rk);                      // I[x+1] - I[x-1] + 255    ((L ^ 0xff) = 255 - L per half)
			const uint32_t gxb = add3(P[k], d, k4);                    // gx + 1024                (P carries 3 x 255)
			P[k] = lshl1_add(d, dprev[k]);
			dprev[k] = d;
			const uint32_t gyb = xad(hyTop[k], k3ff, hyN);             // gy + 1023                ((t ^ 0x3ff) = 1023 - t: t <= 1020)
			hyTop[k] = hyN;
			const uint32_t mx = pk_max_u16(gxb, kBias2k - gxb);        // |gx| + 1024
			const uint32_t my = pk_max_u16(gyb, 0x07fe07feu - gyb);    // |gy| + 1023
			gq[k] = add3(mx, my, k1);                                  // g + 2048
			// bit 10 of gxb ^ gyb = sign(gx) != sign(gy), except that gy = 0 reads as negative: a pixel with gy = 0 is in the horizontal class
			// (or has g = 0) and the sign only selects between the two diagonals
			aux[k] = bfi(kBias1k, gxb ^ gyb, mx);                      // bits 0..9 |gx|, bit 10 = ((gx ^ gy) < 0)
		}
		}
		const int yc = yin - R;
		if (borderTile) { // one wave-uniform test per row; interior tiles skip all of it (the empty asm keeps the compiler from turning the branch into selects)
			asm volatile("" : "+v"(gq[0]), "+v"(gq[1]));
			const uint32_t rowm = (yc >= R && yc <= H - 1 - R) ? 0xffffffffu : 0u; // image border rows (and rows past the image): g = 0
#pragma unroll
			for (int k = 0; k < 2; ++k) gq[k] = bfi(okm[k] & rowm, gq[k], kGpk);
		}
		*reinterpret_cast<uint2*>(ring + sNew * kRowB + lane8) = make_uint2(gq[0], gq[1]);

		// ---- sparse stage: NMS + classification of the row pair (2j, 2j + 1) on its candidates ----
		if (NMS) {
			static_assert(!NMS || (PH & 1), "pairs complete on odd steps");
			constexpr int sA = (PH + 3) & 3;                  // ring slot of row 2j: 0 (j even) or 2 (j odd); row 2j + 1 sits in sA + 1
			__builtin_amdgcn_wave_barrier();
			const int total = (int)listCount;
			const uint8_t* const list = rest + kList;
#pragma nounroll
			for (int base = 0; base < total; base += 64) {
				const int jx = base + lane;
				if (jx < total) {
					// entry = (row parity << 9) | byte offset of the candidate inside a ring row.  The LDS pipe, not the VALU, bounded the kernel
					// while every candidate fetched its eight neighbours (round 3 PMC: 60 % of the LDS cycles were bank conflicts of these
					// scattered 2-byte reads): the direction class is evaluated first, from the centre and its aux word, and only the TWO
					// neighbours along the gradient are fetched.
					const uint32_t e = *reinterpret_cast<const uint16_t*>(list + 2 * jx);
					const uint32_t cAbs = e + (uint32_t)(sA * kRowB);                                    // ring offset of the centre
					const uint32_t eU = (e + (uint32_t)(sA * kRowB + 3 * kRowB)) & (4u * kRowB - 1u);   // same column, row above (ring wraps)
					const uint32_t eD = (e + (uint32_t)(sA * kRowB + kRowB)) & (4u * kRowB - 1u);       // row below
					const int gc = *reinterpret_cast<const uint16_t*>(ring + sA * kRowB + e);
					const uint32_t au = *reinterpret_cast<const uint16_t*>(rest + kAux + e);
					const uint32_t ax = au & kAxMask;
					const uint32_t ays = (uint32_t)(gc - (int)kG - (int)ax) << 16;   // |gy| << 16
					// direction class (constants canny_dete.h:58-61: tan(pi/8), tan(3pi/8) in Q16; 158217 = 27145 + 2^17)
					const uint32_t t1 = __umul24(ax, 27145u);
					const bool k1 = ays < t1;
					const bool k2 = ays < t1 + (ax << 17);
					const bool dg = k2 && ((au & kFlag) != 0);
					// neighbours along the gradient: k1 left / right; k2 diagonal ((gx^gy) < 0: (y+1,x-1),(y-1,x+1), else (y-1,x-1),(y+1,x+1)); else up / down,
					// i.e. (X - d, Y + d) with (X, Y, d) = (C, C, 2) | (D, U, 2) | (U, D, 2) | (U, D, 0)
					uint32_t X = dg ? eD : eU, Y = dg ? eU : eD;
					X = k1 ? cAbs : X; Y = k1 ? cAbs : Y;
					const uint32_t dl = k2 ? 2u : 0u;
					const int n1 = *reinterpret_cast<const uint16_t*>(ring + (X - dl)), n2 = *reinterpret_cast<const uint16_t*>(ring + (Y + dl));
					bool weak = gc >= max(n1, n2);  // not suppressed: neither neighbour strictly greater (candidates already have g > tLow)
					bool strong = gc > tHighQ;
					const uint32_t c = (e >> 1) & 255u;   // local column 0..255
					if (GAP) { // quirk Q3: column coverage of the NMS and of the seed scan, [1,simdEnd) U [cStart,W-1) (canny_dete.cxx:396,514)
						const int x = xbase + (int)c;
						const bool in_cov = (x >= 1 && x < a.simdEnd) || (x >= a.cStart && x < W - 1);
						weak = weak || !in_cov;       // outside the NMS coverage: thresholded only, never a seed
						strong = strong && in_cov;
					}
					// result: bit (c & 3) (U: weak, not strong) or 4 + (c & 3) (E: strong) of byte (c >> 2) of nibble row sA + parity = cAbs >> 9.  One dword
					// holds four lanes: candidates of one instruction rarely share it (the mask-word atomics of the previous version put 8 lanes on one address)
					if (weak) {
						uint32_t* const nw = reinterpret_cast<uint32_t*>(rest + kNib) + ((cAbs >> 9) << 4) + (c >> 4);
						atomicOr(nw, (strong ? 0x10u : 0x01u) << ((c & 3u) | ((c & 12u) << 1)));
					}
				}
			}
			listCount = 0;
			if constexpr (PH == 3) {
				// rows 4m .. 4m + 3 of the tile are classified: masks and edge bytes leave in their final global layout
				const int rr0 = it - 7;                          // tile row of nibble row 0
				__builtin_amdgcn_wave_barrier();
				const uint32_t* const nibw = reinterpret_cast<const uint32_t*>(rest + kNib);
				{
					// masks: lanes 0..31 the U rows (weak & ~strong), 32..63 the E rows (strong), whole row segments per store instruction.  Global
					// dword d0 + dl = the pixels of lanes [8 dl + 2, 8 dl + 10) (even tiles) or [8 dl - 2, 8 dl + 6) (odd): 8 result bytes that start
					// in the middle of a dword (the bytes of a neighbouring tile's columns, or of the next row, end in a half-word that is not stored)
					const int mi = lane >> 5, q = (lane >> 3) & 3, dl = lane & 7;
					const uint32_t* rn = nibw + q * 16 + 2 * dl - odd;
					const uint32_t d0w = rn[0], d1w = rn[1], d2w = rn[2];
					const uint32_t nsh = (uint32_t)mi * 4u;                          // U: low nibbles, E: high nibbles
					uint32_t th[2];
#pragma unroll
					for (int hh = 0; hh < 2; ++hh) {
						const uint32_t by = hh ? __builtin_amdgcn_alignbit(d2w, d1w, 16) : __builtin_amdgcn_alignbit(d1w, d0w, 16);  // 4 lanes = 16 pixels
						uint32_t t = (by >> nsh) & 0x0f0f0f0fu;
						t = (t | (t >> 4)) & 0x00ff00ffu;
						th[hh] = (t | (t >> 8)) & 0x0000ffffu;
					}
					const uint32_t v = th[0] | (th[1] << 16);
					const int row = y0 + rr0 + q, gd = d0 + dl;
					if (row < H && gd < a.wb) {
						uint32_t* dst = (mi ? ebase : ubase) + (size_t)row * a.wb + gd;
						if (!odd && dl == 7) reinterpret_cast<uint16_t*>(dst)[0] = (uint16_t)v;             // low half-word: the next tile owns the high one
						else if (odd && dl == 0) reinterpret_cast<uint16_t*>(dst)[1] = (uint16_t)(v >> 16);  // high half-word: the previous tile owns the low one
						else *dst = v;
					}
				}
				{
					// edge bytes of the strong pixels (the resolve rounds add the promoted ones): lane = (row q, 16-pixel group gi = lanes 2 + 4 gi .. 5 + 4 gi):
					// one 16-byte store
					const int q = lane >> 4, gi = lane & 15;
					const uint32_t* rn = nibw + q * 16 + gi;
					const uint32_t by = __builtin_amdgcn_alignbit(rn[1], rn[0], 16);
					const int row = y0 + rr0 + q;
					const int x = tileX * kSwCols + gi * 16;
					if (gi < 15 && row < H && x + 8 <= a.So) {
						uint8_t* dst = obase + (size_t)row * a.So + x;
						const uint32_t b0 = nibble_bytes((by >> 4) & 0xfu), b1 = nibble_bytes((by >> 12) & 0xfu);
						if (fullRow) {
							// (spelled out: with the two tails below in one if / else the compiler merges their common part and emits a 12-byte store plus a
							// 4-byte store at 16-byte stride -- twice the write requests, every one of them partial: 0.193 -> 0.177 ms per launch)
							const uint32_t b2 = nibble_bytes((by >> 20) & 0xfu), b3 = nibble_bytes(by >> 28);
							typedef uint32_t v4u __attribute__((ext_vector_type(4)));
							const v4u bv = { b0, b1, b2, b3 };
							asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(bv) : "memory");
						}
						else if (x + 16 <= a.So) {
							const uint32_t b2 = nibble_bytes((by >> 20) & 0xfu), b3 = nibble_bytes(by >> 28);
							*reinterpret_cast<uint4*>(dst) = make_uint4(b0, b1, b2, b3);
						}
						else *reinterpret_cast<uint2*>(dst) = make_uint2(b0, b1); // So % 8 == 0: an 8-column tail
					}
				}
				__builtin_amdgcn_wave_barrier();
				zero_nibbles();
			}
			__builtin_amdgcn_wave_barrier();
		}

		// aux of the new row (its slot held the aux of row 2j until the NMS above was done with it)
		*reinterpret_cast<uint2*>(rest + kAux + aNew * kRowB + lane8) = make_uint2(aux[0], aux[1]);

		// ---- candidates of the new row join the list: entry = (row parity << 9) | byte offset inside a ring row, position = scalar
		// popcount of the earlier slots (rides in as the mbcnt base) + mbcnt of the slot's own mask; the store is exec-masked ----
		if (it >= 3 && it < kSwRows + 3) {   // gradient rows y0 .. y0 + kSwRows - 1 only (uniform)
			uint8_t* const listw = rest + kList;
			uint32_t cnt = listCount;
#pragma unroll
			for (int p = 0; p < kSwPx; ++p) {
				const uint32_t gk = gq[p >> 1];
				const uint32_t gp = (p & 1) ? (gk >> 16) : (gk & 0xffffu);
				const bool c = gp > thrV;
				const uint64_t mk = __ballot(c);
				if (c) {
					const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, cnt));
					*reinterpret_cast<uint16_t*>(listw + twice(rk)) = (uint16_t)((aNew << 9) | (lane * 8 + 2 * p));
				}
				cnt += (uint32_t)__popcll(mk);
			}
			listCount = __builtin_amdgcn_readfirstlane(cnt);
		}
	};

	{
		using T = std::true_type; using F = std::false_type;
		static_assert(kSwRows % 4 == 0, "the row loop is unrolled by four");
		step(std::integral_constant<int, 0>{}, F{}, 0);
		step(std::integral_constant<int, 1>{}, F{}, 1);
		step(std::integral_constant<int, 2>{}, F{}, 2);
		step(std::integral_constant<int, 3>{}, F{}, 3);
		for (int it = 4; it < kSwRows + 4; it += 4) { // it = 4 .. kSwRows + 3
			step(std::integral_constant<int, 0>{}, F{}, it);
			step(std::integral_constant<int, 1>{}, T{}, it + 1);
			step(std::integral_constant<int, 2>{}, F{}, it + 2);
			step(std::integral_constant<int, 3>{}, T{}, it + 3);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
template <int kWaves, int kRows, int KS>
static hipError_t launch_swar(const CannyArgs& a0, int frames, bool gap, hipStream_t stream)
{
	CannyArgs a = a0;
	a.tilesX = (a.W + kSwCols - 1) / kSwCols;
	a.tilesY = (a.H + kRows - 1) / kRows;
	a.blockRows = (a.tilesY + kWaves - 1) / kWaves;
	a.groups = a.blockRows * frames;
	dim3 grid(8 * ((a.groups + 7) / 8) * a.tilesX);
	dim3 block(kWaves * 64);
	// Input rows through the LDS exchange buffer (one load per lane and row) when the launch is several times the chip's 8192 wave slots -- then the kernel runs at
	// its throughput and a third of the vector-memory requests is worth 1 % (32 x 4K: 46 080 waves, 0.989 of the three-loads kernel) --; smaller launches are a few
	// rounds of latency chains, which the exchange lengthens (32 x 1080p: 11 520 waves, 1.017).  tools/canny_lab forces either path.
	bool xch = (KS == 3) && (long long)a.groups * a.tilesX >= 4 * 8192;
#if defined(SWAR_NO_LDS_LR)
	xch = false;
#elif defined(SWAR_FORCE_LDS_LR)
	xch = (KS == 3);
#endif
	if constexpr (KS == 3) {
		if (xch) {
			if (gap) hipLaunchKernelGGL((canny_swar_tile_kernel<true, kWaves, kRows, KS, true>), grid, block, 0, stream, a);
			else hipLaunchKernelGGL((canny_swar_tile_kernel<false, kWaves, kRows, KS, true>), grid, block, 0, stream, a);
			return hipGetLastError();
		}
	}
	if (gap) hipLaunchKernelGGL((canny_swar_tile_kernel<true, kWaves, kRows, KS, false>), grid, block, 0, stream, a);
	else hipLaunchKernelGGL((canny_swar_tile_kernel<false, kWaves, kRows, KS, false>), grid, block, 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_canny_tiles_swar(const CannyArgs& a0, int frames, bool gap, hipStream_t stream)
{
	// One wave per workgroup (every LDS address of the kernel is lane * k + constant), 24 rows per wave tile.  Per 32 x 4K launch 128 / 64 / 32 rows took
	// 0.232 / 0.202 / 0.192 ms (round 3), 24 .. 40 rows are within 1.5 % of each other there -- the 4-row halo costs more gradient rows on shorter tiles, but
	// more, shorter waves balance the SIMDs better at the end of the launch (a tile's time follows its candidate count).  Smaller frames decide: 32 x 1080p is
	// 8 704 waves of 32 rows on 8 192 wave slots -- 24 rows: 0.953 of the 32-row kernel at 1080p, 0.934 at 720p, 0.991 at 4K (tools/canny_lab, round 5).
	if (a0.ksize == 5) return launch_swar<1, SWAR_ROWS, 5>(a0, frames, gap, stream);
	return launch_swar<1, SWAR_ROWS, 3>(a0, frames, gap, stream);
}

} // namespace COMPVHIP_SWAR_NS
