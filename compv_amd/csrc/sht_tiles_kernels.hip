// sht_tiles_kernels.hip -- Hough (SHT) vote accumulation, second generation: lane = theta, image tiles, bank = lane.
//
// Replaces, behind compvhip_houghsht_u8 / compvhip_plan_houghsht / compvhip_plan_pipeline:
//   CompVHoughSht::acc_gather + CompVHoughShtAccGatherRow_* / CompVHoughShtRowTimesSinRho_*
//   core/features/hough/compv_core_feature_houghsht.cxx:350-481,607-627 (+ intrin/x86/compv_core_feature_houghsht_intrin_avx2.cxx:41-98)
//
// Why: the first-generation kernel (sht_kernels.hip: lane = edge, one workgroup per theta pair, the whole rho column in LDS) is
// bound by the LDS atomic pipe at the cost of RANDOM addresses -- 7.3 cycles per ds_add wave-instruction, 8.4 on real frames where
// collinear edges also pile onto one address -- against 4.1 cycles when the 32 lanes of each half-wave hit 32 different banks
// (tools/microbench/lds_atomic_bench2).  Here the 64 lanes of a wave are 64 theta bins voting for ONE edge: the histogram of a
// workgroup is [window row][32 dwords], dword l of a row = the u16 counters of theta bins l (low half) and 32 + l (high half), so
// lane l always hits bank l & 31: conflict-free by construction, no two lanes of an instruction ever share an address, and the
// edge coordinates are wave-uniform (scalar loads, scalar operands).  64 full rho columns do not fit the LDS, so the edges are
// binned into image tiles: a tile of diagonal d only reaches a window of <= d + 1 rho rows per theta (<= 1264 rows = 158 KB).
// A workgroup = (frame, tile, 64 theta bins); it writes its window theta-major to a partial accumulator -- a plane of count bytes,
// plus the high bytes of the few columns that hold a count >= 256 -- and a small reduce kernel adds the tiles' windows into the
// accumulator (each partial cell is written once and read once: 2 x 2.8 MB per 4K frame).
// This also removes the W + H <= 20 479 limit of the first generation: the LDS only ever holds one tile's window.
//
// Exactness: rho = (x cosQ + y sinQ) >> 16 with x = x0 + lx, y = y0 + ly.  With C = x0 cosQ + y0 sinQ = Chi * 65536 + Clo
// (Clo in [0, 65536)): rho = Chi + q, q = (lx cosQ + ly sinQ + Clo) >> 16, and the accumulator row barrier - rho =
// rowBase + w with w = qmax - q = (K - lx cosQ - ly sinQ) >> 16, K = qmax * 65536 + 65535 - Clo (host tables, int64 there).
#include "kernels.hpp"

namespace compvhip {

// ---------------------------------------------------------------------------------------------------------------
// bit masks -> per-tile edge lists, entry = (ly << 16) | lx (tile-local).  One workgroup = kTcRows rows of one tile; the words of
// the chunk are spread over the threads, popcounts are prefix-summed in the workgroup, ONE atomic reserves the chunk's slice of the
// tile's list (the order of the entries inside a tile does not matter: votes commute).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kTcThreads = 256;
constexpr int kTcRows = 64;
constexpr int kTcStage = 6144; // entries of a chunk staged in the LDS (denser chunks store directly)

__global__ __launch_bounds__(kTcThreads) void sht_compact_tiles_kernel(ShtArgs a, ShtTileArgs v)
{
	__shared__ int s_wave[kTcThreads / 64];
	__shared__ int s_base;
	__shared__ uint32_t s_stage[kTcStage];
	const int frame = blockIdx.y;
	const int chunksPerTile = (v.TH + kTcRows - 1) / kTcRows;
	const int tile = blockIdx.x / chunksPerTile, chunk = blockIdx.x - tile * chunksPerTile;
	const int ty = tile / v.nx, tx = tile - ty * v.nx;
	const int x0 = tx * v.TW, y0 = ty * v.TH;
	const int wpr = v.TW >> 5;                                  // mask words per tile row
	const uint32_t wprInv = (1u << 20) / (uint32_t)wpr + 1u;    // i / wpr == (i * wprInv) >> 20 for every i < 64 * 40 and wpr <= 40 (verified exhaustively)
	const int ly0 = chunk * kTcRows;
	const int rows = min(min(kTcRows, v.TH - ly0), a.H - (y0 + ly0));
	const int nwords = max(rows, 0) * wpr;
	const uint32_t* __restrict__ bits = a.ebits + (size_t)frame * a.bitsFrameStride;
	const int w0 = x0 >> 5;
	constexpr int kPer = (kTcRows * 40 + kTcThreads - 1) / kTcThreads; // words per thread; tiles are at most 1264 = kShtMaxWindow columns wide (40 words)
	uint32_t wv[kPer]; int wl[kPer];
	int cnt = 0;
	if (nwords == 0) return; // uniform
	// The loads are unconditional (clamped addresses, masked afterwards): all kPer of a thread are in flight together.  Written as
	// `if (i < nwords) wv[k] = bits[...]` the compiler emits branch, load, s_waitcnt per word -- ten exposed memory latencies per thread.
#pragma unroll
	for (int k = 0; k < kPer; ++k) {
		const int i = min((int)threadIdx.x + k * kTcThreads, nwords - 1);
		const int r = (int)(((uint32_t)i * wprInv) >> 20), c = i - r * wpr;   // (an integer division per word cost a third of the kernel)
		wv[k] = bits[(size_t)(y0 + ly0 + r) * a.wb + min(w0 + c, a.wb - 1)];
		wl[k] = ((ly0 + r) << 16) | (c << 5);
	}
#pragma unroll
	for (int k = 0; k < kPer; ++k) {
		const int i = (int)threadIdx.x + k * kTcThreads;
		const int c = (wl[k] & 0xffff) >> 5;
		if (!(i < nwords && w0 + c < a.wb)) wv[k] = 0u;
		cnt += __popc(wv[k]);
	}
	int incl = cnt;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int n = __shfl_up(incl, o);
		if (lane >= o) incl += n;
	}
	if (lane == 63) s_wave[wave] = incl;
	__syncthreads();
	int wbase = 0, total = 0;
#pragma unroll
	for (int k = 0; k < kTcThreads / 64; ++k) {
		const int t = s_wave[k];
		if (k < wave) wbase += t;
		total += t;
	}
	if (total == 0) return; // uniform
	// (the frame's edge count is summed from the tile counts by the voting kernel: one atomicAdd per chunk on edgeCounts[frame] put all
	// 4608 workgroups of a 32 x 4K launch on ONE cache line -- the counters of 32 frames -- and the L2 serialises the atomics of a line:
	// 37 of this kernel's 57 us)
	if (threadIdx.x == 0) s_base = atomicAdd(&v.tileCounts[frame * v.tiles + tile], total);
	__syncthreads();
	uint32_t* __restrict__ dst = a.edges + ((size_t)frame * v.tiles + tile) * v.tileCap;
	const int first = wbase + (incl - cnt);
	if (total <= kTcStage) {
		// the usual case: the chunk's entries are put in order in the LDS and leave as whole 256-byte rows (a thread storing its own
		// entries one by one touches 64 different cache lines per store instruction: the L2 request rate, not the bytes, was the cost)
		int lp = first;
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			uint32_t b = wv[k];
			while (b) {
				const int bit = __ffs(b) - 1;
				b &= b - 1;
				s_stage[lp++] = (uint32_t)(wl[k] + bit);
			}
		}
		__syncthreads();
		const size_t room = (size_t)s_base < v.tileCap ? v.tileCap - (size_t)s_base : 0;
		const int nout = (int)min((size_t)total, room);
		for (int i = threadIdx.x; i < nout; i += kTcThreads) dst[(size_t)s_base + i] = s_stage[i];
		return;
	}
	size_t pos = (size_t)s_base + first;
#pragma unroll
	for (int k = 0; k < kPer; ++k) {
		uint32_t b = wv[k];
		while (b) {
			const int bit = __ffs(b) - 1;
			b &= b - 1;
			if (pos < v.tileCap) dst[pos] = (uint32_t)(wl[k] + bit);
			++pos;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// voting: workgroup = (frame, tile, group of 64 theta bins), 1024 threads; lane = theta bin.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kVtThreads = 1024;
constexpr int kVtUnroll = 32;   // edges per block of scalar loads

__global__ __launch_bounds__(kVtThreads) void sht_vote_tiles_kernel(ShtArgs a, ShtTileArgs v)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t hist[]; // [Rw][32]: dword l of row w = counters of theta l (low half) and 32 + l (high half)
	// (the voting code addresses the histogram from LDS offset 0: no static LDS objects in this kernel)
	// XCD-aware order (workgroup b runs on XCD b % 8): the groups of one (frame, tile) unit run on the same XCD, so the unit's edge list
	// is fetched from HBM once and re-read from that XCD's L2 by the other groups.
	const int units = a.frames * v.tiles;
	const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
	const int g = k % v.groups;
	const int unit = (k / v.groups) * 8 + xcd;
	if (unit >= units) return;
	const int frame = unit / v.tiles, tile = unit - frame * v.tiles;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

	const int nAll = v.tileCounts[unit];
	if (g == 0 && threadIdx.x == 0 && nAll) atomicAdd(&a.edgeCounts[frame], nAll);   // edges of the frame (entries past a full list included)
	const int n = min(nAll, (int)v.tileCap);
	const int words = v.Rw * 32;
	uint32_t* const s_any = hist + words;   // two dwords behind the histogram: lane mask of the columns that hold a count >= 256
	if (tid < 2) s_any[tid] = 0u;
	for (int i = tid * 4; i < words; i += kVtThreads * 4) *reinterpret_cast<uint4*>(&hist[i]) = make_uint4(0, 0, 0, 0);

	const int t = g * 64 + lane;
	const int tt = min(t, a.T - 1);
	const int nc = -a.cosQ[tt], ns = -a.sinQ[tt];
	const int K = v.kt[(size_t)tile * a.T + tt];
	uint32_t inc = (t < a.T) ? ((lane & 32) ? 0x10000u : 1u) : 0u; // a theta past T adds nothing
	asm volatile("" : "+v"(inc));
	const uint32_t lane4 = (uint32_t)(lane & 31) * 4u;
	__syncthreads();

	const uint32_t* __restrict__ list = a.edges + (size_t)unit * v.tileCap;
	// The waves take interleaved blocks of kVtUnroll edges.  The block address is wave-uniform: one SCALAR load per block (s_load_dwordx8,
	// issued a block ahead), the coordinates are unpacked by the scalar unit and enter the two v_mad_i32_i24 as scalar operands: per
	// vote 2 multiply-adds + 2 address operations + the LDS atomic.
	typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
	auto vote = [&](uint32_t e) {
		const int lx = (int)(e & 0xffffu), ly = (int)(e >> 16);
		int val, ad;
		asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(val) : "s"(lx), "v"(nc), "v"(K));      // K - lx cos
		asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(val) : "s"(ly), "v"(ns), "v"(val));    // ... - ly sin: window row in the high half
		const uint32_t row = (uint32_t)val >> 16;
		asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(ad) : "v"(row), "v"(lane4));           // row * 128 B + bank * 4 B
		asm volatile("ds_add_u32 %0, %1" : : "v"(ad), "v"(inc) : "memory");
	};
	// (Measured and dropped in round 3: four votes at a time, stage by stage -- four independent dependency chains per wave instead of one
	// register reused vote after vote: 0.389 ms against 0.370 ms; with four waves per SIMD the chains of different waves already overlap,
	// and four atomics back to back from one wave only queue up in front of the LDS.)
	// Blocks of kVtUnroll = 32 edges (two s_load_dwordx16), one block in flight while the previous one is voted: a scalar load that misses
	// the constant cache takes ~800 cycles here, 32 votes of one wave take ~2000.  Two register sets alternate (no copies).
	// (A load and the s_waitcnt that covers it are tied by a "+s" operand: the compiler must not read the registers before the wait.)
	const int nblk = n / kVtUnroll;
	constexpr int kStep = kVtThreads / 64;
	u32x16 a0, a1, b0, b1;
#define VT_LOAD(r0, r1, blk) asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r0), "=&s"(r1) : "s"(list + (size_t)(blk) * kVtUnroll) : "memory")
#define VT_WAIT(r0, r1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1) : : "memory")
#define VT_VOTE(r0, r1) _Pragma("unroll") for (int u = 0; u < 16; ++u) vote(r0[u]); _Pragma("unroll") for (int u = 0; u < 16; ++u) vote(r1[u])
	int b = wave;
	if (b < nblk) {
		VT_LOAD(a0, a1, b);
		VT_WAIT(a0, a1);
		for (;;) {
			const int bn = b + kStep;
			if (bn < nblk) VT_LOAD(b0, b1, bn);
			VT_VOTE(a0, a1);
			if (bn >= nblk) break;
			VT_WAIT(b0, b1);
			b = bn + kStep;
			if (b < nblk) VT_LOAD(a0, a1, b);
			VT_VOTE(b0, b1);
			if (b >= nblk) break;
			VT_WAIT(a0, a1);
		}
	}
#undef VT_LOAD
#undef VT_WAIT
#undef VT_VOTE
	for (int i = nblk * kVtUnroll + wave; i < n; i += kStep) vote(__builtin_amdgcn_readfirstlane(list[i]));
	__builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the asm ds_add are invisible to the compiler's counters
	__syncthreads();

	// flush, theta-major, into the tile's partial window partLo[frame][tile][theta][w]: lane = theta reads 16 consecutive window rows of its own
	// column (bank = lane: conflict-free) and stores the LOW BYTES of the 16 counts as one 16-byte store.  Counts of 256 and more are rare (a
	// tile's share of a strong line: 0.03 % of the window cells of the benchmark frames, in 2 - 5 of a tile's 180 columns): the columns that
	// hold one are flagged, and a second pass stores the high bytes of those columns only.  The reduce kernel adds the byte planes back together.
	// Round 3 stored u16 counts (167 MB written + 172 MB read back per 32 x 4K launch; 96 % of the window cells are shared by 3.7 tiles on
	// average, so they do have to make the round trip -- as bytes now).
	// (a wave writes 128 consecutive rows = one whole 128-byte line per theta back to back, so that the L2 merges the eight 16-byte pieces
	// before the line is evicted)
	const size_t column = ((size_t)unit * v.Tpad) + (size_t)g * 64 + lane;
	uint8_t* __restrict__ partLo = v.partLo + column * v.rwPitch;
	uint8_t* __restrict__ partHi = v.partHi + column * v.rwPitch;
	// two counter dwords -> (lo0, lo1, hi0, hi1); lanes 32..63 own the high halves
	const uint32_t sel1 = (lane & 32) ? 0x07030602u : 0x05010400u;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	auto bytes16 = [&](int w0, u32x4& lo, u32x4& hi) {   // Rw is a multiple of 16
		uint32_t h[16];
#pragma unroll
		for (int j = 0; j < 16; ++j) h[j] = hist[(w0 + j) * 32 + (lane & 31)];
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t p01 = __builtin_amdgcn_perm(h[4 * q + 1], h[4 * q], sel1), p23 = __builtin_amdgcn_perm(h[4 * q + 3], h[4 * q + 2], sel1);
			lo[q] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
			hi[q] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
		}
	};
	uint32_t any = 0u;
	for (int wb = wave * 128; wb < v.Rw; wb += (kVtThreads / 64) * 128)
	for (int w0 = wb; w0 < min(wb + 128, v.Rw); w0 += 16) {
		u32x4 lo, hi;
		bytes16(w0, lo, hi);
		any |= hi.x | hi.y | hi.z | hi.w;
		// (one 16-byte store, spelled out: the loop vectoriser otherwise splits it into four dword stores, 4x the store instructions)
		// (tiles at the image border are not clipped: part of their windows lies outside the accumulator's rows and never receives a vote)
		uint8_t* const dst = partLo + w0;
		asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(lo) : "memory");
	}
	const uint64_t mine = __ballot(any != 0u);
	if (lane == 0 && mine) { atomicOr(&s_any[0], (uint32_t)mine); atomicOr(&s_any[1], (uint32_t)(mine >> 32)); }
	__syncthreads();
	const uint32_t m0 = s_any[0], m1 = s_any[1];
	const bool flagged = (((lane & 32) ? m1 : m0) >> (lane & 31)) & 1u;
	if (wave == 0) v.colFlag[column] = flagged ? 1 : 0;
	if ((m0 | m1) == 0u) return; // uniform
	if (flagged) {
		for (int wb = wave * 128; wb < v.Rw; wb += (kVtThreads / 64) * 128)
		for (int w0 = wb; w0 < min(wb + 128, v.Rw); w0 += 16) {
			u32x4 lo, hi;
			bytes16(w0, lo, hi);
			uint8_t* const dst = partHi + w0;
			asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(dst), "v"(hi) : "memory");
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// reduce: acc[frame][theta][r] = sum over the tiles whose window of that theta covers row r.  For a fixed theta the window rows of
// a tile are contiguous in r: every read and the write are coalesced along r.  One thread = 16 consecutive rows (16 count bytes per
// window, two 16-byte stores), one workgroup = 1024 rows of one theta (workgroups whose rows no window of the theta reaches return at once); a tile whose window misses the thread's rows is skipped (at 4K
// 67 % of the accumulator's cells are covered by no tile -- they are never written and stay zero --, 5 % by one, the rest by 3.7 of the
// 12 on average).  The byte planes are added two cells per 32-bit lane (even bytes / odd bytes of a dword), the high-byte plane only for
// the flagged columns.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRdThreads = 64;
constexpr int kRdRows = 16;

__global__ __launch_bounds__(kRdThreads) void sht_reduce_tiles_kernel(ShtArgs a, ShtTileArgs v)
{
	const int frame = blockIdx.z, theta = blockIdx.y;
	const int2 reach = v.reach[theta];
	if ((int)(blockIdx.x * kRdThreads * kRdRows) >= reach.y || (int)((blockIdx.x + 1) * kRdThreads * kRdRows) <= reach.x) return;   // uniform: no window of this theta comes near
	const int r0 = (blockIdx.x * kRdThreads + threadIdx.x) * kRdRows;   // a multiple of 16, like every window start and Rw (planVoteTiles)
	if (r0 >= a.accPitch) return;                                          // accPitch is a multiple of 64
	uint32_t E[4] = { 0, 0, 0, 0 }, O[4] = { 0, 0, 0, 0 };   // per dword d of the 16 bytes: cells 4d, 4d + 2 (halves of E[d]) and 4d + 1, 4d + 3 (O[d])
	int covers = 0;
	for (int tile = 0; tile < v.tiles; ++tile) {
		const int base = v.rowBase[(size_t)tile * a.T + theta];
		const int w0 = r0 - base;
		if (w0 < 0 || w0 >= v.Rw) continue;
		++covers;
		const size_t column = ((size_t)frame * v.tiles + tile) * v.Tpad + theta;
		const uint4 c = *reinterpret_cast<const uint4*>(v.partLo + column * v.rwPitch + w0);
		E[0] += c.x & 0x00ff00ffu; O[0] += (c.x >> 8) & 0x00ff00ffu;
		E[1] += c.y & 0x00ff00ffu; O[1] += (c.y >> 8) & 0x00ff00ffu;
		E[2] += c.z & 0x00ff00ffu; O[2] += (c.z >> 8) & 0x00ff00ffu;
		E[3] += c.w & 0x00ff00ffu; O[3] += (c.w >> 8) & 0x00ff00ffu;
		if (v.colFlag[column]) {   // uniform in the workgroup (one theta)
			const uint4 h = *reinterpret_cast<const uint4*>(v.partHi + column * v.rwPitch + w0);
			// a cell never exceeds 2 max(W, H) < 65536: the packed halves cannot carry
			E[0] += (h.x & 0x00ff00ffu) << 8; O[0] += h.x & 0xff00ff00u;
			E[1] += (h.y & 0x00ff00ffu) << 8; O[1] += h.y & 0xff00ff00u;
			E[2] += (h.z & 0x00ff00ffu) << 8; O[2] += h.z & 0xff00ff00u;
			E[3] += (h.w & 0x00ff00ffu) << 8; O[3] += h.w & 0xff00ff00u;
		}
	}
	if (!covers) return;
	uint32_t o[8];
#pragma unroll
	for (int d = 0; d < 4; ++d) {
		o[2 * d] = (E[d] & 0xffffu) | (O[d] << 16);
		o[2 * d + 1] = (E[d] >> 16) | (O[d] & 0xffff0000u);
	}
	uint4* dst = reinterpret_cast<uint4*>(a.acc + (size_t)frame * a.accFrameStride + (size_t)theta * a.accPitch + r0);
	dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
	dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// ---------------------------------------------------------------------------------------------------------------
hipError_t launch_sht_compact_tiles(const ShtArgs& a, const ShtTileArgs& v, int frames, hipStream_t stream)
{
	const int chunksPerTile = (v.TH + kTcRows - 1) / kTcRows;
	dim3 grid((unsigned)(v.tiles * chunksPerTile), frames);
	hipLaunchKernelGGL(sht_compact_tiles_kernel, grid, dim3(kTcThreads), 0, stream, a, v);
	return hipGetLastError();
}

size_t sht_vote_tiles_lds_bytes(int Rw) { return (size_t)Rw * 128 + 16; }   // histogram + the column flags of the flush

hipError_t launch_sht_vote_tiles(const ShtArgs& a, const ShtTileArgs& v, int frames, hipStream_t stream)
{
	const size_t lds = sht_vote_tiles_lds_bytes(v.Rw);
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	static size_t attr_lds[64] = {};
	int dev = 0;
	(void)hipGetDevice(&dev);
	dev = (dev >= 0 && dev < 64) ? dev : 0;
	if (lds > attr_lds[dev]) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sht_vote_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
		attr_lds[dev] = lds;
	}
	const int units = frames * v.tiles;
	dim3 grid((unsigned)(8 * ((units + 7) / 8) * v.groups));
	hipLaunchKernelGGL(sht_vote_tiles_kernel, grid, dim3(kVtThreads), lds, stream, a, v);
	return hipGetLastError();
}

hipError_t launch_sht_reduce_tiles(const ShtArgs& a, const ShtTileArgs& v, int frames, hipStream_t stream)
{
	dim3 grid((unsigned)((a.R + kRdThreads * kRdRows - 1) / (kRdThreads * kRdRows)), a.T, frames);
	hipLaunchKernelGGL(sht_reduce_tiles_kernel, grid, dim3(kRdThreads), 0, stream, a, v);
	return hipGetLastError();
}

} // namespace compvhip
