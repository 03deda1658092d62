// sht_kernels.hip -- standard Hough transform (SHT) line accumulation for gfx950, hand-written HIP.
//
// Replaces, behind compvhip_houghsht_u8 / compvhip_plan_houghsht (include/compv_hip.h):
//   CompVHoughSht::process                  core/features/hough/compv_core_feature_houghsht.cxx:96-262
//   acc_gather + AccGatherRow/RowTimesSinRho ...houghsht.cxx:350-481,607-627 (+ intrin_avx2.cxx:41-98)
//   nms_gather / nms_apply                  ...houghsht.cxx:483-564,629-668 (+ intrin_sse2.cxx:16-101)
//
// Data layout in HBM (per frame):
//   edge bit mask   u32 [H][wb]            produced by the Canny kernels (or by bytes_to_bits for foreign edge maps)
//   edge lists      u32 per image tile     (ly << 16) | lx, compacted with popcount prefix sums (sht_tiles_kernels.hip)
//   accumulator     u16 [T][accPitch]      THETA-major (the reference is int32 rho-major [R][192]); written exactly once, 16-byte stores,
//                                          no global atomics
//   NMS flag planes u8  [T/8][rows]        bit j of byte (group, row) = column 8 group + j survives the 3x3 test and the threshold
//   line keys/cells u32 + u32 [lineCap]    key = frameTag | strength, value = cell (row*T + col), in (row, col) order, the frames' lines one behind the
//                                          other; ordered by strength by sht_sort_kernels.hip (sized on the device) or, for max(W, H) > 4095, by one
//                                          STABLE descending library radix sort over the slots in use
//
// Voting: rho = (x*cosQ[t] + y*sinQ[t]) >> 16 (int32, arithmetic shift), acc[barrier - rho][t]++ for every edge and
// every t -- E*T scattered increments, the whole cost of the reference's SHT: sht_tiles_kernels.hip (lane = theta over image tiles).
// This file: foreign edge maps -> bit masks, sht_nms_kernel / sht_lines_kernel, the library key sort + sht_decode_kernel (the
// fallback of sht_sort_kernels.hip), sht_cartesian_kernel, the accumulator export.
#include "kernels.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

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
// NMS + threshold -> (key, value) pairs in DETERMINISTIC order (accumulator rows, then columns, ascending: nms_apply's own order,
// houghsht.cxx:546-562), so that the sort only has to order by strength.  Two kernels:
//   sht_nms_kernel    3x3 test of 8 theta columns x 8 rho rows per thread straight from registers; the survivors leave as one flag byte per
//                     (row, column group): flag planes [frame][column group][row], 8-byte coalesced stores; the survivors per 64 rows are
//                     added to blockCounts[] (one atomic per row block and column group), the wave's total to the frame's total;
//   sht_lines_kernel  a workgroup (one wave, lane = row) owns 64 COMPLETE accumulator rows (all theta columns), so its survivors occupy one contiguous
//                     range of the frame's key / value arrays: the rows' survivors are counted and prefix-summed in the wave, the range's
//                     start is the sum of the earlier blocks' survivor counts (blockCounts[], accumulated by the NMS), key = frameTag | strength and value = cell (row * T + col) are put in place in
//                     the LDS and stored coalesced; every block zeroes a slice of the frame's unused key slots.
//   (rounds 3-5 had a sht_count_kernel between the two -- survivors per 64 rows re-counted from the flag planes, one wave per row block; since round 6 the
//                     NMS counts as it goes.)
//                     (Measured and dropped: a decoupled look-back chain instead of the block counts -- every block waits for atomic round
//                     trips to the L2: 27 us of that kernel's 42; one atomicAdd per NMS thread -- 184 adds on
//                     every counter: +30 us in the NMS; the NMS inside the lines kernel too -- no flag planes, no second read of the strengths: 0.106 ms
//                     against 0.021 + 0.03: at 150 VGPRs four 3-wave workgroups fit a CU, and load -> test -> barrier -> scan -> look-back ->
//                     barrier -> emit is one latency chain per workgroup.)
// A stable descending radix sort of the (key, value) pairs then gives frame-major, strength-descending, (row, col)-ascending order with
// frameBits + strengthBits key bits (18 at 4K x 32 frames: two 10-bit onesweep passes; the unique 40-bit keys of rounds 1-2, which carried
// the cell because the slots were handed out by atomics in arrival order, took four).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kNmsCols = 8;                  // theta columns per thread
constexpr int kNmsRows = 1024;               // granularity of a flag plane's row count (sht_nms_rows)
constexpr int kNmsBlkRows = 64;              // rows of one row block of sht_lines_kernel: their survivor counts come out of this pass
constexpr int kNmsWaves = 2;                 // waves per workgroup
constexpr int kNmsWgRows = 512;              // rho rows per wave: 8 rows per lane, i.e. 8 row blocks of 64

// A thread owns 8 theta columns x 8 consecutive rho rows of the theta-major accumulator (+ one column / row either side): ten 16-byte coalesced loads
// straight into registers (each accumulator cell is read (kNmsCols+2)/kNmsCols times), the rows above / below its eight with two 2-byte loads per column
// (lines its neighbours fetch anyway): no LDS tile -- the LDS-tiled version of rounds 1-2 ran load -> barrier -> compute per block at 3.5 waves per SIMD:
// 45 us for 87 MB.  A thread's survivors are one byte per row (bit j = column c0 + j).
// Round 6: the survivors of a 64-row block (= 8 lanes) are summed with three lane shuffles and ADDED to blockCounts[] by the octet's first lane (one
// atomic per row block and column group: 23 per counter at T = 180, only where there are survivors), the wave's total to the frame total:
// sht_count_kernel (a second pass over the flag planes, 6.6 us + a launch) is gone for +0.4 us here.  Measured on the way, per 32 x 4K launch, against
// 22.6 (NMS) + 6.6 (count): a workgroup = 64 rows x all column groups, 8 lanes per column, plain stores: 34.5 us; 512 rows x all column groups, a wave
// looping over three column groups: 34.7 us (a third of the loads in flight); 8 column groups x 512 rows per workgroup, counts summed in the LDS, 3 atomics
// per counter: 26.5 us (the barrier keeps eight waves' registers until the slowest is done); this shape: 23.0 us.  Round 3's "+30 us" was one atomic per
// THREAD (184 per counter, 32 counters per 128-byte line).
__global__ __launch_bounds__(kNmsWaves * 64) void sht_nms_kernel(ShtArgs a)
{
	const int frame = blockIdx.z;
	const int cg = blockIdx.y;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int base = ((int)blockIdx.x * kNmsWaves + wave) * kNmsWgRows;
	const int r0 = base + lane * 8;
	const uint16_t* __restrict__ acc = a.acc + (size_t)frame * a.accFrameStride;
	const size_t p = a.accPitch;
	uint32_t survivors = 0;                      // of this lane's 8 rows
	{
	// Two thirds of the accumulator's cells lie outside every tile's rho window of their theta (no pixel of the image maps there): they are
	// never written, stay zero, and their flag bytes (zeroed when the plan was made) are never written either.
	const int2 reach = a.nmsRange[cg];
	if (!(base >= reach.y || base + kNmsWgRows <= reach.x)) {   // wave-uniform
	const int c0 = cg * kNmsCols;
	// All 30 loads of a thread are unconditional (clamped addresses, invalid cells zeroed afterwards), so they are in flight together.
	const int rc = min(r0, a.accPitch - 8);                       // accPitch is a multiple of 64 rows
	const int ru = min(max(r0 - 1, 0), a.accPitch - 1), rd = min(r0 + 8, a.accPitch - 1);
	const bool rok = r0 < a.accPitch, uok = r0 >= 1 && r0 - 1 < a.accPitch, dok = r0 + 8 < a.accPitch;
	uint4 v[kNmsCols + 2];
	uint32_t up[kNmsCols + 2], dn[kNmsCols + 2];   // rows r0 - 1 and r0 + 8 of every column
#pragma unroll
	for (int j = 0; j < kNmsCols + 2; ++j) {
		const int c = c0 - 1 + j;
		const uint16_t* __restrict__ col = acc + (size_t)min(max(c, 0), a.T - 1) * p;
		v[j] = *reinterpret_cast<const uint4*>(col + rc);
		up[j] = col[ru];
		dn[j] = col[rd];
	}
#pragma unroll
	for (int j = 0; j < kNmsCols + 2; ++j) {
		const int c = c0 - 1 + j;
		const bool cok = (c >= 0 && c < a.T);
		if (!(cok && rok)) v[j] = make_uint4(0, 0, 0, 0);
		if (!(cok && uok)) up[j] = 0u;
		if (!(cok && dok)) dn[j] = 0u;
	}
	// x[1..8] = the thread's rows of tile column tj, x[0] / x[9] the rows above / below
	auto readCol = [&](int tj, int (&x)[10]) {
		const uint4 w = v[tj];
		x[0] = (int)up[tj];
		x[1] = w.x & 0xffffu; x[2] = w.x >> 16; x[3] = w.y & 0xffffu; x[4] = w.y >> 16;
		x[5] = w.z & 0xffffu; x[6] = w.z >> 16; x[7] = w.w & 0xffffu; x[8] = w.w >> 16;
		x[9] = (int)dn[tj];
	};
	// Per-row masks of this thread (rows r0 .. r0 + 7): mV = the row exists (r < R), mN = the row takes part in the NMS (1 <= r <= R - 2).
	// The per-cell test below is straight-line code on them -- the version with `if (nmsCol && r >= 1 && ...)` per cell spent as many
	// scalar branch / exec-mask instructions as vector ones (PMC: 754 SALU + 844 VALU per wave).
	int mV[8], mN[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const int r = r0 + k;
		mV[k] = (r < a.R) ? -1 : 0;
		mN[k] = (r >= 1 && r <= a.R - 2) ? -1 : 0;
		asm volatile("" : "+v"(mV[k]), "+v"(mN[k]));   // real register masks (the compiler otherwise keeps them as lane masks and selects per cell)
	}
	int side[3][8]; // max(x[k], x[k+1], x[k+2]) of tile columns j, j+1, j+2 (ring)
	int mid[10], nxt[10];
	{
		int x[10];
		readCol(0, x);
#pragma unroll
		for (int k = 0; k < 8; ++k) side[0][k] = max(max(x[k], x[k + 1]), x[k + 2]);
		readCol(1, mid);
#pragma unroll
		for (int k = 0; k < 8; ++k) side[1][k] = max(max(mid[k], mid[k + 1]), mid[k + 2]);
	}
	uint32_t flags[2] = { 0u, 0u };   // byte k & 3 of flags[k >> 2] = row k of this thread: bit j = column c0 + j survives
#pragma unroll
	for (int j = 0; j < kNmsCols; ++j) {
		readCol(j + 2, nxt);
		int* sl = side[j % 3];
		int* sh = side[(j + 2) % 3];
#pragma unroll
		for (int k = 0; k < 8; ++k) sh[k] = max(max(nxt[k], nxt[k + 1]), nxt[k + 2]);
		const int c = c0 + j;
		int colN = (c >= 1 && c <= a.nmsLastCol) ? -1 : 0;   // quirk Q2: NMS covers theta columns [1, (T-1)&~3]
		int colV = (c < a.T) ? -1 : 0;
		asm volatile("" : "+v"(colN), "+v"(colV));
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const int val = mid[k + 1] & (mV[k] & colV);                                          // 0 where there is no cell: never above the threshold
			const int nb = max(max(sl[k], sh[k]), max(mid[k], mid[k + 2])) & (mN[k] & colN);     // 0 where no NMS applies: never above val
			// pass = (val > threshold) && (nb <= val), from the signs of two differences (all values < 65536): no compare -> scalar mask -> select chain
			const uint32_t p = (uint32_t)(a.threshold - val) & ~(uint32_t)(val - nb);
			flags[k >> 2] |= (p >> (31 - ((k & 3) * 8 + j))) & (1u << ((k & 3) * 8 + j));
		}
#pragma unroll
		for (int k = 0; k < 10; ++k) mid[k] = nxt[k];
	}
	// flag plane of this column group: rows r0 .. r0 + 7, one 8-byte store per thread (a wave writes 512 consecutive bytes)
	uint8_t* __restrict__ plane = a.nmsFlags + ((size_t)frame * a.nmsGroups + cg) * a.nmsRows;
	*reinterpret_cast<uint2*>(plane + r0) = make_uint2(flags[0], flags[1]);
	survivors = (uint32_t)(__popc(flags[0]) + __popc(flags[1]));
	}
	}
	// survivors per 64-row block (8 lanes each): what sht_count_kernel used to re-count from the flag planes
	uint32_t cnt = survivors;
#pragma unroll
	for (int o = 1; o < 8; o <<= 1) cnt += __shfl_xor(cnt, o);
	const int nblk = (a.R + kNmsBlkRows - 1) / kNmsBlkRows;
	const int blk = (base >> 6) + (lane >> 3);
	if ((lane & 7) == 0 && cnt && blk < nblk) atomicAdd(a.blockCounts + frame * nblk + blk, (int)cnt);   // one add per column group: 23 per counter at T = 180; zeroed per step
	uint32_t tot = cnt;
#pragma unroll
	for (int o = 8; o < 64; o <<= 1) tot += __shfl_xor(tot, o);
	if (lane == 0 && tot) atomicAdd(a.frameTotals + (size_t)frame * kFrameSlot, (int)tot);   // the frame's total: where the NEXT frame's lines start (dense key array)
}

constexpr int kLnRows = 64;               // rows per workgroup = lanes of its one wave
constexpr int kLnStage = 512;             // (key, value) slots staged in the LDS (a 64-row block of a 4K frame holds ~300; 4 KB keep all 32 wave slots of a CU usable); denser row blocks store directly

// The flag bytes of one accumulator row (lane = row) for 32 column groups starting at cg0, packed into 8 registers: 32 independent
// (clamped, hence unconditional) loads in flight together -- a `count += popc(load)` loop is compiled into load -> wait -> add per column
// group, 23 exposed memory latencies.
__device__ __forceinline__ void sht_load_flags(const uint8_t* __restrict__ planes, int nmsRows, int groups, int cg0, uint32_t (&fw)[8])
{
	uint32_t b[32];
#pragma unroll
	for (int i = 0; i < 32; ++i) b[i] = (uint32_t)planes[(size_t)min(cg0 + i, groups - 1) * nmsRows];
#pragma unroll
	for (int i = 0; i < 32; ++i) b[i] = (cg0 + i < groups) ? b[i] : 0u;
#pragma unroll
	for (int i = 0; i < 8; ++i) fw[i] = b[4 * i] | (b[4 * i + 1] << 8) | (b[4 * i + 2] << 16) | (b[4 * i + 3] << 24);
}

// One WAVE per 64 complete accumulator rows, lane = row: no barriers, every workgroup of the launch resident at once (the version with
// 184-thread workgroups and LDS tables ran load -> barrier -> scan -> look-back -> barrier -> emit as one latency chain per workgroup:
// 0.07 ms against the 0.046 ms of the rank / emit / pad kernels it replaced).
__global__ __launch_bounds__(kLnRows) void sht_lines_kernel(ShtArgs a)
{
	__shared__ uint32_t s_keys[kLnStage], s_vals[kLnStage];
	// XCD-aware order (workgroup i runs on XCD i % 8): the row blocks of one frame run on ONE XCD: neighbouring blocks share
	// the cache lines of the flag planes and of the key / value arrays through that XCD's L2
	const int nblk = (a.R + kLnRows - 1) / kLnRows;
	const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
	const int frame = (kk / nblk) * 8 + xcd, blk = kk % nblk;
	if (frame >= a.frames) return; // uniform
	const int lane = threadIdx.x;
	const int r = blk * kLnRows + lane;                       // < nmsRows (whole NMS blocks); rows >= R have no survivors
	const int groups = a.nmsGroups;
	const uint8_t* __restrict__ planes = a.nmsFlags + (size_t)frame * groups * a.nmsRows + r;
	const uint16_t* __restrict__ acc = a.acc + (size_t)frame * a.accFrameStride + r;

	// 1. survivors of the lane's row, exclusive scan over the block's rows
	auto loadFlags = [&](int cg0, uint32_t (&fw)[8]) { sht_load_flags(planes, a.nmsRows, groups, cg0, fw); };
	uint32_t fw0[8];      // column groups 0 .. 31: all of them up to T = 256
	loadFlags(0, fw0);
	uint32_t cnt = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) cnt += (uint32_t)__popc(fw0[i]);
	for (int cg0 = 32; cg0 < groups; cg0 += 32) {
		uint32_t fw[8];
		loadFlags(cg0, fw);
#pragma unroll
		for (int i = 0; i < 8; ++i) cnt += (uint32_t)__popc(fw[i]);
	}
	uint32_t incl = cnt;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t n = __shfl_up(incl, o);
		if (lane >= o) incl += n;
	}
	const uint32_t total = __shfl(incl, 63);
	// 2. the block's first slot = the survivors of the frame's earlier row blocks (blockCounts[] of sht_nms_kernel): a few independent loads and a
	// wave reduction.  (A decoupled look-back chain over the blocks -- no counters, status words written by this kernel -- cost 27 us of this
	// kernel's 42: every block waits for atomic round trips to the L2.)
	const int* __restrict__ bc = a.blockCounts + (size_t)frame * nblk;
	uint32_t first = 0, count = 0;   // survivors of the earlier row blocks / of the whole frame
	for (int i0 = 0; i0 < nblk; i0 += 4 * kLnRows) {
		uint32_t v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) v[u] = (uint32_t)bc[min(i0 + u * kLnRows + lane, nblk - 1)];   // clamped: unconditional loads
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int i = i0 + u * kLnRows + lane;
			first += (i < blk) ? v[u] : 0u;
			count += (i < nblk) ? v[u] : 0u;
		}
	}
	// the frame's first slot in the DENSE key array = the lines (clamped to lineCap) of the earlier frames; `grand` = those of all frames.  The sort then
	// covers the slots that exist instead of frames * lineCap padded ones (the reference sorts lines.size() elements, houghsht.cxx:241-249)
	const uint32_t capU = (uint32_t)min(a.lineCap, (size_t)0xffffffffu);
	uint32_t fbase = 0, grand = 0;
	for (int g0 = 0; g0 < a.frames; g0 += kLnRows) {
		const int g = g0 + lane;
		const uint32_t c = min((uint32_t)max(a.frameTotals[(size_t)min(g, a.frames - 1) * kFrameSlot], 0), capU);   // clamped index: unconditional load
		fbase += (g < frame) ? c : 0u;
		grand += (g < a.frames) ? c : 0u;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { first += __shfl_xor(first, o); count += __shfl_xor(count, o); fbase += __shfl_xor(fbase, o); grand += __shfl_xor(grand, o); }
	uint32_t* __restrict__ keys = a.lineKeys + (size_t)fbase;
	uint32_t* __restrict__ vals = a.lineVals + (size_t)fbase;
	// 3. the survivors in (row, column) order: key = frameTag | strength, value = cell (row * T + col), put in place in the LDS and stored
	// coalesced (a lane storing its own few items would touch 64 different cache lines per instruction)
	if (total) {
		const bool staged = total <= (uint32_t)kLnStage;
		const uint32_t frameTag = (uint32_t)(a.frames - 1 - frame) << a.strengthBits;
		uint32_t pos = incl - cnt;   // block-local
		// A lane only puts its survivors' cells in place (LDS); the strengths are gathered afterwards, one staged slot per lane: independent
		// loads, 64 in flight (a lane gathering its own row's strengths one after the other exposed a memory latency per survivor)
		auto emitFlags = [&](int cg0, const uint32_t (&fw)[8]) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				uint32_t f = fw[i];                        // 4 column groups = 32 theta columns, ascending
				const int c0 = (cg0 + 4 * i) * kNmsCols;
				while (f) {
					const int j = __ffs(f) - 1;
					f &= f - 1;
					const uint32_t c = (uint32_t)(c0 + j), val = (uint32_t)r * (uint32_t)a.T + c;
					if (staged) { s_keys[pos] = (c << 6) | (uint32_t)lane; s_vals[pos] = val; }
					else if ((size_t)first + pos < a.lineCap) { keys[(size_t)first + pos] = frameTag | (uint32_t)acc[(size_t)c * a.accPitch]; vals[(size_t)first + pos] = val; }
					++pos;
				}
			}
		};
		if (cnt) emitFlags(0, fw0);
		for (int cg0 = 32; cg0 < groups; cg0 += 32) {
			uint32_t fw[8];
			loadFlags(cg0, fw);
			emitFlags(cg0, fw);
		}
		if (staged) {
			__syncthreads();   // one wave: the LDS stores of the other lanes
			const uint16_t* __restrict__ acc0 = a.acc + (size_t)frame * a.accFrameStride + (size_t)blk * kLnRows;
			// four slots per lane and step: the four strength loads are in flight together
			for (uint32_t i0 = lane; i0 < total; i0 += 4 * kLnRows) {
				uint32_t pk[4], vv[4], st4[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t i = min(i0 + u * kLnRows, total - 1);   // clamped: unconditional loads
					pk[u] = s_keys[i]; vv[u] = s_vals[i];
				}
#pragma unroll
				for (int u = 0; u < 4; ++u) st4[u] = (uint32_t)acc0[(size_t)(pk[u] >> 6) * a.accPitch + (pk[u] & 63u)];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t i = i0 + u * kLnRows;
					const size_t p = (size_t)first + i;
					if (i < total && p < a.lineCap) { keys[p] = frameTag | st4[u]; vals[p] = vv[u]; }
				}
			}
		}
	}
	// 4. the line count, the slots in use, and this block's slice of the slots between them and the end of the sorted range (a zero key sorts
	// last: every real key carries a strength > 0).  (Left to one block alone, the zeroing is 24 us on the critical path of a frame with few lines.)
	if (blk == 0 && lane == 0) {
		a.lineCounts[frame] = (int)min(count, 0x7fffffffu);
		if (a.outCounts) a.outCounts[frame] = (int32_t)min(count, 0x7fffffffu);
		if (frame == 0) *a.lineTotal = grand;
	}
	if (a.hostStep && blk == 0 && frame == 0 && lane < 4) {   // what compvhip_plan_wait looks at, straight into its pinned slot (visible to the host when the step's event has fired)
		if (lane == 0) a.hostStep[0] = (int)grand;
		a.hostStep[kFrameSlot + lane] = a.stepFlags[lane];
	}
	if ((size_t)grand < a.sortN) {
		const size_t nb = (size_t)a.frames * (size_t)nblk, bi = (size_t)frame * (size_t)nblk + (size_t)blk;
		const size_t pad = a.sortN - grand, per = (pad + nb - 1) / nb;
		for (size_t i = bi * per + lane; i < min((bi + 1) * per, pad); i += kLnRows) a.lineKeys[(size_t)grand + i] = 0u;
	}
}

struct LineOut { float rho; float theta; int32_t strength; int32_t row; int32_t col; };

// After the global descending sort the lines of frame f start at sum_{g<f} min(count_g, lineCap).
__global__ __launch_bounds__(256) void sht_decode_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, const int* __restrict__ counts, size_t lineCap,
                                                         int T, int barrier, float thetaStep, int maxLines, int strengthBits, LineOut* __restrict__ lines, size_t outCap)
{
	const int frame = blockIdx.y;
	// the frame's offset: thread g adds the count of frame g (one thread walking the earlier frames' counts exposed a memory latency per
	// frame: 10 us for the last frames of a batch of 32)
	__shared__ unsigned long long s_part[4];
	unsigned long long before = 0;
	for (int g = threadIdx.x; g < frame; g += 256) {
		const size_t cg = (size_t)max(counts[g], 0);
		before += cg < lineCap ? cg : lineCap;
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = before;
	__syncthreads();
	const size_t off = (size_t)(s_part[0] + s_part[1] + s_part[2] + s_part[3]);
	size_t n = (size_t)max(counts[frame], 0);
	if (n > lineCap) n = lineCap;
	if (maxLines > 0 && n > (size_t)maxLines) n = (size_t)maxLines;
	if (n > outCap) n = outCap;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t k = keys[off + i];
	const uint32_t cell = vals[off + i];
	const int row = (int)(cell / (uint32_t)T), col = (int)(cell - (uint32_t)row * (uint32_t)T);
	LineOut o;
	o.rho = (float)(barrier - row);              // static_cast<float>(barrier - row), houghsht.cxx:661
	o.theta = __fmul_rn((float)col, thetaStep);  // col * theta (f32), houghsht.cxx:662
	o.strength = (int32_t)(k & ((1u << strengthBits) - 1u));
	o.row = row; o.col = col;
	lines[(size_t)frame * outCap + i] = o;
}

// CompVHoughSht::toCartesian (houghsht.cxx:566-589) on the device line arrays.  cos(theta) and 1/sin(theta) come from HOST
// tables indexed by the line's theta column (libm cosf/sinf, as the reference; device trig would not be bit-identical);
// the kernel only multiplies and subtracts, with single correctly-rounded operations.
__global__ __launch_bounds__(256) void sht_cartesian_kernel(const LineOut* __restrict__ lines, const int* __restrict__ counts, size_t lineCap, int maxLines, int T,
                                                            const float* __restrict__ cosT, const float* __restrict__ invSinT, float widthF, float r, float4* __restrict__ out)
{
	// rho - W*a must stay a rounded product followed by a rounded subtraction, as on the CPU: the library is built with
	// -ffp-contract=off (the __f*_rn intrinsics are plain operators that a later FMA contraction would still fuse)
	const int frame = blockIdx.y;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	size_t n = (size_t)max(counts[frame], 0);
	if (n > lineCap) n = lineCap;
	if (maxLines > 0 && n > (size_t)maxLines) n = (size_t)maxLines;   // counts[] holds the uncut line count: the decode stage wrote min(count, lineCap, maxLines) slots
	if (i >= n) return;
	const LineOut l = lines[(size_t)frame * lineCap + i];
	float4 o;
	if (l.theta == 0.f) o = make_float4(l.rho, r, l.rho, -r); // perfect vertical line
	else if (l.col < 0 || l.col >= T) o = make_float4(0.f, 0.f, 0.f, 0.f);   // not a line of this plan (foreign / uninitialised slot): no table entry to read
	else {
		const float a = cosT[l.col], b = invSinT[l.col];
		o = make_float4(0.f, __fmul_rn(l.rho, b), widthF, __fmul_rn(__fsub_rn(l.rho, __fmul_rn(widthF, a)), b));
	}
	out[(size_t)frame * lineCap + i] = o;
}

__global__ __launch_bounds__(256) void sht_acc_transpose_kernel(const uint16_t* __restrict__ accT, int R, int T, int accPitch, int32_t* __restrict__ out, size_t outStride)
{
	__shared__ int32_t tile[32][33];
	const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
	for (int j = ty; j < 32; j += 8) {
		const int c = c0 + j, r = r0 + tx;
		tile[j][tx] = (c < T && r < R) ? (int32_t)accT[(size_t)c * accPitch + r] : 0;
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

int sht_nms_groups(int T) { return (T + kNmsCols - 1) / kNmsCols; }
size_t sht_nms_rows(int R) { return (size_t)((R + kNmsRows - 1) / kNmsRows) * kNmsRows; }   // rows of a flag plane (whole NMS blocks)
int sht_lines_blocks(int R) { return (R + kLnRows - 1) / kLnRows; }

hipError_t launch_sht_lines(const ShtArgs& a, int frames, hipStream_t stream)
{
	const int nblk = sht_lines_blocks(a.R);
	static_assert(kNmsBlkRows == kLnRows, "the NMS counts the survivors of the row blocks sht_lines_kernel owns");
	static_assert(kNmsRows % kNmsWgRows == 0 && kNmsWgRows % kNmsBlkRows == 0, "flag planes hold whole NMS waves; a wave holds whole row blocks");
	dim3 ngrid((a.R + kNmsWaves * kNmsWgRows - 1) / (kNmsWaves * kNmsWgRows), (a.T + kNmsCols - 1) / kNmsCols, frames);
	hipLaunchKernelGGL(sht_nms_kernel, ngrid, dim3(kNmsWaves * 64), 0, stream, a);
	hipLaunchKernelGGL(sht_lines_kernel, dim3((unsigned)(8 * ((frames + 7) / 8) * nblk)), dim3(kLnRows), 0, stream, a);
	return hipGetLastError();
}

// one stable descending radix sort over the (key, value) slots of all frames (rocPRIM device primitive)
hipError_t sht_sort_pairs(void* temp, size_t& tempBytes, const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valsIn, uint32_t* valsOut, size_t n,
                          int keyBits, hipStream_t stream)
{
	// 10-bit digits: the 18-bit key of the 4K benchmark (5 frame bits + 13 strength bits) takes two onesweep passes
	using Onesweep = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 10>, 10, rocprim::block_radix_rank_algorithm::match>;
	using Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, Onesweep>;
	return rocprim::radix_sort_pairs_desc<Config>(temp, tempBytes, keysIn, keysOut, valsIn, valsOut, n, 0u, (unsigned int)keyBits, stream);
}

hipError_t launch_sht_decode(const uint32_t* keys, const uint32_t* vals, const int* counts, size_t lineCap, int frames, int T, int barrier, float thetaStep,
                             int maxLines, int strengthBits, void* lines, size_t outCap, hipStream_t stream)
{
	size_t n = lineCap < outCap ? lineCap : outCap;
	if (maxLines > 0 && (size_t)maxLines < n) n = (size_t)maxLines;
	if (n == 0) return hipSuccess;
	dim3 grid((unsigned)((n + 255) / 256), frames);
	hipLaunchKernelGGL(sht_decode_kernel, grid, dim3(256), 0, stream, keys, vals, counts, lineCap, T, barrier, thetaStep, maxLines, strengthBits,
	                   reinterpret_cast<LineOut*>(lines), outCap);
	return hipGetLastError();
}

hipError_t launch_sht_cartesian(const void* lines, const int* counts, size_t lineCap, int maxLines, int frames, int T, const float* cosT, const float* invSinT,
                                float widthF, float r, float* out, hipStream_t stream)
{
	if (!lineCap) return hipSuccess;
	const size_t span = (maxLines > 0 && (size_t)maxLines < lineCap) ? (size_t)maxLines : lineCap;
	dim3 grid((unsigned)((span + 255) / 256), frames);
	hipLaunchKernelGGL(sht_cartesian_kernel, grid, dim3(256), 0, stream, reinterpret_cast<const LineOut*>(lines), counts, lineCap, maxLines, T, cosT, invSinT, widthF, r,
	                   reinterpret_cast<float4*>(out));
	return hipGetLastError();
}

hipError_t launch_sht_acc_transpose(const uint16_t* accT, int R, int T, int accPitch, int32_t* out, size_t outStride, hipStream_t stream)
{
	dim3 grid((R + 31) / 32, (T + 31) / 32);
	hipLaunchKernelGGL(sht_acc_transpose_kernel, grid, dim3(256), 0, stream, accT, R, T, accPitch, out, outStride);
	return hipGetLastError();
}

} // namespace compvhip
