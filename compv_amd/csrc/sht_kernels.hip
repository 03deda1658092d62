// sht_kernels.hip -- standard Hough transform (SHT) line accumulation for gfx950, hand-written HIP.
//
// Replaces, behind compvhip_houghsht_u8 / compvhip_plan_houghsht (include/compv_hip.h):
//   CompVHoughSht::process                  core/features/hough/compv_core_feature_houghsht.cxx:96-262
//   acc_gather + AccGatherRow/RowTimesSinRho ...houghsht.cxx:350-481,607-627 (+ intrin_avx2.cxx:41-98)
//   nms_gather / nms_apply                  ...houghsht.cxx:483-564,629-668 (+ intrin_sse2.cxx:16-101)
//
// Data layout in HBM (per frame):
//   edge bit mask   u32 [H][wb]            produced by the Canny kernels (or by bytes_to_bits for foreign edge maps)
//   edge list       u32 [E]   (y<<16)|x    compacted with wave ballot/popcount prefix sums (first generation; per-tile lists: sht_tiles_kernels.hip)
//   accumulator     u16 [T][accPitch]      THETA-major (the reference is int32 rho-major [R][192]); each vote workgroup
//                                          owns kShtThetaPerGroup whole theta columns, so the accumulator is
//                                          written exactly once, coalesced, with no global atomics
//   line keys       u64 [frames][lineCap]  frameTag | strength << cellBits | (cellMask - cell), cell = row*T + col: unique keys, one
//                                          descending radix sort over all frames
//
// Voting: rho = (x*cosQ[t] + y*sinQ[t]) >> 16 (int32, arithmetic shift), acc[barrier - rho][t]++ for every edge and
// every t -- E*T scattered increments, the whole cost of the reference's SHT.  The DEFAULT voting path is the second generation in
// sht_tiles_kernels.hip (lane = theta over image tiles).  The first generation kept in this file (COMPVHIP_SHT_VOTE=legacy; the
// A/B reference, limited to W + H <= 20 479): each workgroup privatises the rho histogram of 2 theta bins in LDS (two u16 counters
// per dword: a cell can never exceed the number of pixels in a 1-px-wide band < 65536 for W,H <= 32767) and votes with ds_add_u32;
// its edge list is stored transposed per compaction block (sht_compact_kernel) so that a wave's 64 simultaneous votes land on
// different image rows: raster neighbours share rho around theta = 90 deg and would otherwise serialise on one LDS address.
// Shared by both generations: sht_nms_kernel, the key sort, sht_decode_kernel, sht_cartesian_kernel, the accumulator export.
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
// bit mask -> edge list.  Each thread owns kCompactWords consecutive 32-px words; popcounts are prefix-summed inside
// the wave (shuffles) and across the workgroup's waves (LDS), and ONE global atomic per workgroup reserves the output
// range (a single device-scope counter sustains only ~90 atomics/us, MI355X_MICROARCH "dequeue" row).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCompactThreads = 256;
constexpr int kCompactWords = 8;
constexpr int kCompactStage = 8192; // edges a block can expand through LDS (it owns 256*8*32 = 65536 pixels). Measured alternatives:
// 4 words/4096: compact 0.112 ms, vote 0.53 ms (less decorrelation); 16 words/16384: compact 0.119 ms, vote 0.47 ms; 8/8192: 0.072 / 0.48 ms

__global__ __launch_bounds__(kCompactThreads) void sht_compact_kernel(ShtArgs a)
{
	__shared__ int s_wave[kCompactThreads / 64];
	__shared__ int s_base;
	__shared__ uint32_t s_stage[kCompactStage];
	const int frame = blockIdx.y;
	const size_t nwords = (size_t)a.H * a.wb; // wb is a multiple of 16, so nwords % kCompactWords == 0
	const size_t w0 = ((size_t)blockIdx.x * kCompactThreads + threadIdx.x) * kCompactWords;
	uint32_t bits[kCompactWords];
#pragma unroll
	for (int k = 0; k < kCompactWords; ++k) bits[k] = 0u;
	if (w0 < nwords) {
		const uint4* src = reinterpret_cast<const uint4*>(a.ebits + (size_t)frame * a.bitsFrameStride + w0);
#pragma unroll
		for (int v = 0; v < kCompactWords / 4; ++v) {
			const uint4 q = src[v];
			bits[4 * v] = q.x; bits[4 * v + 1] = q.y; bits[4 * v + 2] = q.z; bits[4 * v + 3] = q.w;
		}
	}
	int cnt = 0;
#pragma unroll
	for (int k = 0; k < kCompactWords; ++k) cnt += __popc(bits[k]);
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
	for (int k = 0; k < kCompactThreads / 64; ++k) {
		const int t = s_wave[k];
		if (k < wave) wbase += t;
		total += t;
	}
	if (total == 0) return; // uniform
	if (threadIdx.x == 0) s_base = atomicAdd(&a.edgeCounts[frame], total);
	__syncthreads();
	uint32_t* __restrict__ dst = a.edges + (size_t)frame * a.edgeCap;
	// all kCompactWords words of a thread lie in one image row (wb % 8 == 0): one division per thread
	const uint32_t row = (uint32_t)(w0 / a.wb);
	const uint32_t yx00 = (row << 16) | (((uint32_t)w0 - row * (uint32_t)a.wb) * 32u);
	if (total <= kCompactStage) {
		// usual case: expand into LDS, then copy the block's slice of the list out with coalesced stores.  The slice is stored
		// TRANSPOSED: raster entry i = 64 r + c of the block (a [nr][64] matrix whose last row holds m entries) goes to list
		// position c (nr-1) + min(c, m) + r, i.e. the matrix is emitted column by column, so 64 consecutive list entries are
		// 64 raster positions apart -- the voting kernel can then hand consecutive entries to the 64 lanes of one ds_add
		// without piling onto a single rho bin near theta = 90 deg.  (Computed per raster entry: no division.)
		const int nr = (total + 63) >> 6;
		const int m = total - (nr - 1) * 64;
		int lp = wbase + incl - cnt;
#pragma unroll
		for (int k = 0; k < kCompactWords; ++k) {
			uint32_t b = bits[k];
			const uint32_t yx0 = yx00 + 32u * k;
			while (b) {
				const int bit = __ffs(b) - 1;
				b &= b - 1;
				const int c = lp & 63, r = lp >> 6;
				s_stage[c * (nr - 1) + min(c, m) + r] = yx0 + (uint32_t)bit;
				++lp;
			}
		}
		__syncthreads();
		for (int q = threadIdx.x; q < total; q += kCompactThreads) {
			const size_t pos = (size_t)s_base + q;
			if (pos < a.edgeCap) dst[pos] = s_stage[q];
		}
		return;
	}
	size_t pos = (size_t)s_base + wbase + (incl - cnt);
	if (cnt) {
#pragma unroll
		for (int k = 0; k < kCompactWords; ++k) {
			uint32_t b = bits[k];
			const uint32_t yx0 = yx00 + 32u * k;
			while (b) {
				const int bit = __ffs(b) - 1;
				b &= b - 1;
				if (pos < a.edgeCap) dst[pos] = yx0 + (uint32_t)bit;
				++pos;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// voting
// ---------------------------------------------------------------------------------------------------------------
// LDS: TG/2 arrays [Rp] of packed u16 counter pairs, nothing else. The edge list arrives already decorrelated (see
// sht_compact_kernel: 64 consecutive entries are 64 raster positions apart), so every wave streams its own 64-edge chunks
// with plain coalesced loads and the main loop has NO workgroup barrier: the LDS atomic pipe never drains.
constexpr int kVoteUnroll = 4;                           // 64-edge chunks in flight per wave
constexpr int kVoteFramesInL2 = 2;                       // frames per XCD whose theta groups are in flight together

size_t sht_vote_lds_bytes(int R, int tg)
{
	return (size_t)((R + 31) & ~31) * (tg / 2) * sizeof(uint32_t);
}

// SC ("scaled"): TG == 2 and W+H < 8192.  The Q16 products are formed with 4*cosQ / 4*sinQ so that the HIGH HALF of the
// 32-bit sum is the histogram BYTE offset (4 * rho index, < 65536): one v_and_b32 with an SDWA WORD_1 source select replaces
// the shift + scaled address add -- 4 VALU per vote (2 SDWA v_mul_i32_i24, v_add3, v_and) instead of 5.
template <int TG, bool SC>
__global__ __launch_bounds__(kShtVoteThreads) void sht_vote_kernel(ShtArgs a)
{
	constexpr int kShtThetaPerGroup = TG;
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	const int R = a.R;
	// [TG/2][Rp]: pair 0 = thetas (t0,t0+1), pair 1 = (t0+2,t0+3) as u16 halves.  Pair-major so that one ds_add_u32
	// instruction (fixed pair, 64 different rho) can spread over all banks ([R][2] would only ever touch half).
	const int Rp = (R + 31) & ~31;
	uint32_t* hist = smem;
	// XCD-aware placement (workgroup b runs on XCD b % 8, each XCD has its own L2): frame f is always voted on XCD f % 8, so an
	// XCD's L2 only ever holds the edge lists of ceil(frames/8) frames, which its resident workgroups (different theta groups of those
	// frames) re-read from L2 instead of HBM.  Within an XCD the theta groups are walked in groupOrder (expensive first), frames fastest.
	// The XCD's frames are taken kVoteFramesInL2 at a time (2 x 1.5 MB of edges at 4K against a 4 MB L2).
	const int framesPerXcd = (a.frames + 7) >> 3;
	const int groups = (a.T + kShtThetaPerGroup - 1) / kShtThetaPerGroup;
	const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
	const int perSub = kVoteFramesInL2 * groups;
	const int sub = k / perSub, rem = k - sub * perSub;
	const int grank = rem / kVoteFramesInL2;
	const int fi = sub * kVoteFramesInL2 + (rem - grank * kVoteFramesInL2);
	const int frame = fi * 8 + xcd;
	if (fi >= framesPerXcd || frame >= a.frames) return; // padding workgroup
	const int shard = blockIdx.y;
	const int t0 = a.groupOrder[grank] * kShtThetaPerGroup;
	const int tid = threadIdx.x;

	const int nthreads = blockDim.x; // 256..1024 (launch_sht_vote)
	for (int i = tid; i < (TG / 2) * Rp; i += nthreads) hist[i] = 0u;

	// hist index of theta k = barrier - ((x*cosQ + y*sinQ) >> 16) = (K - x*cosQ - y*sinQ) >> 16 with K = (barrier << 16) + 65535
	// (exact: barrier - floor(v/65536) == floor((barrier*65536 + 65535 - v)/65536)): two v_mad_i32_i24, a shift and the address add
	int ncq[kShtThetaPerGroup], nsq[kShtThetaPerGroup];
	uint32_t inc[kShtThetaPerGroup];
#pragma unroll
	for (int k = 0; k < kShtThetaPerGroup; ++k) {
		const int t = min(t0 + k, a.T - 1);
		ncq[k] = -a.cosQ[t] * (SC ? 4 : 1);
		nsq[k] = -a.sinQ[t] * (SC ? 4 : 1);
		inc[k] = (t0 + k < a.T) ? ((k & 1) ? 0x10000u : 1u) : 0u; // a theta past T adds nothing (branch-free inner loop)
		if constexpr (SC) asm volatile("" : "+v"(inc[k])); // keep the addend in a VGPR (ds_add data operand), not re-materialised per vote
	}
	// SC: the LDS byte address of hist[0] rides in the high half too, so the masked high half IS the ds_add address
	const uint32_t K = (((uint32_t)a.barrier << 16) + 65535u) * (SC ? 4u : 1u) + (SC ? (((uint32_t)reinterpret_cast<uintptr_t>(hist) & 0xffffu) << 16) : 0u);
	const uint32_t offMask = 0xfffcu;
	const int nvalid = min(kShtThetaPerGroup, a.T - t0);

	const int n = min(a.edgeCounts[frame], (int)a.edgeCap);
	const int sBeg = (int)(((long long)n * shard) / a.shards);
	const int sEnd = (int)(((long long)n * (shard + 1)) / a.shards);
	const int cnt = sEnd - sBeg;
	const uint32_t* __restrict__ edges = a.edges + (size_t)frame * a.edgeCap + sBeg;
	__syncthreads(); // histogram zeroed

	auto voteEdge = [&](uint32_t xy) {
		const int x = (int)(xy & 0xffffu), y = (int)(xy >> 16);
#pragma unroll
		for (int k = 0; k < kShtThetaPerGroup; ++k) {
			const uint32_t v = (uint32_t)__mul24(x, ncq[k]) + ((uint32_t)__mul24(y, nsq[k]) + K); // modulo 2^32 by construction
			if constexpr (SC) {
				uint32_t off;
				asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(off) : "v"(offMask), "v"(v));
				asm volatile("ds_add_u32 %0, %1" : : "v"(off), "v"(inc[k]) : "memory");
			}
			else atomicAdd(&hist[(k >> 1) * Rp + (v >> 16)], inc[k]);
		}
	};
	// Main loop: whole steps of kVoteUnroll * nthreads edges, no bounds or sentinel tests (the per-edge bookkeeping would cost as
	// many VALU cycles as the votes: the kernel sits at ~80 % of BOTH the VALU and the LDS pipe).  The step base is wave-uniform
	// (scalar pointer arithmetic), the lane offset a constant VGPR; loads of step i+1 are in flight while step i votes; the
	// two register sets alternate (loop unrolled by two) so no copies are needed.
	const uint32_t step = (uint32_t)kVoteUnroll * (uint32_t)nthreads;
	const uint32_t full = (uint32_t)cnt / step;
	auto fetch = [&](uint32_t it, uint32_t (&e)[kVoteUnroll]) {
		const uint32_t* __restrict__ base = edges + (size_t)it * step; // uniform
#pragma unroll
		for (int u = 0; u < kVoteUnroll; ++u) e[u] = (base + u * nthreads)[(uint32_t)tid];
	};
	auto vote = [&](const uint32_t (&e)[kVoteUnroll]) {
#pragma unroll
		for (int u = 0; u < kVoteUnroll; ++u) voteEdge(e[u]);
	};
	uint32_t ra[kVoteUnroll], rb[kVoteUnroll];
	uint32_t it = 0;
	if (full > 0) fetch(0, ra);
	for (; it + 1 < full; it += 2) {
		fetch(it + 1, rb);
		vote(ra);
		if (it + 2 < full) fetch(it + 2, ra);
		vote(rb);
	}
	if (it < full) vote(ra);
	// tail: fewer than one step left
	for (int j = (int)(full * step) + tid; j < cnt; j += nthreads) voteEdge(edges[j]);
	if constexpr (SC) __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the asm ds_add are invisible to the compiler's counters
	__syncthreads();

	// flush: the accumulator is u16 (a cell never exceeds 65535, see above), theta-major; each thread writes two
	// adjacent rho rows of one theta as one dword (accPitch is even)
	uint16_t* __restrict__ acc = a.acc + (size_t)frame * a.accFrameStride + (size_t)t0 * a.accPitch;
	for (int r2 = tid; r2 < Rp / 2; r2 += nthreads) {
		const int r = 2 * r2;
		const uint32_t a0 = hist[r], a1 = hist[r + 1];
		const uint32_t b0 = (TG > 2) ? hist[Rp + r] : 0u, b1 = (TG > 2) ? hist[Rp + r + 1] : 0u;
		// rows r (low half) and r+1 (high half) for each of the 4 thetas
		const uint32_t w[4] = { (a0 & 0xffffu) | (a1 << 16), (a0 >> 16) | (a1 & 0xffff0000u), (b0 & 0xffffu) | (b1 << 16), (b0 >> 16) | (b1 & 0xffff0000u) };
#pragma unroll
		for (int k = 0; k < kShtThetaPerGroup; ++k) {
			if (k < nvalid) {
				uint32_t* dst = reinterpret_cast<uint32_t*>(acc + (size_t)k * a.accPitch + r);
				if (a.shards == 1) *dst = w[k];
				else if (w[k]) atomicAdd(dst, w[k]); // halves cannot carry into each other: every cell total <= 65535
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// NMS + threshold -> line keys (sht_nms_kernel below: LDS-tiled 3x3 test, one global atomic per workgroup for the key slots).
// key = frameTag << (strengthBits+cellBits) | strength << cellBits | (cellMask - cell), cell = row*T + col: unique, and a single
// descending radix sort over all frames yields frame-major, strength-descending, (row,col)-ascending order.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kNmsThreads = 128;
constexpr int kNmsCols = 8;                  // theta columns per block
constexpr int kNmsRows = kNmsThreads * 8;    // rho rows per block: 8 per thread
constexpr int kNmsTileRows = kNmsRows + 16;  // + 8 rows of halo either side (keeps every 16-byte load aligned)

// One block owns kNmsCols theta columns x kNmsRows rho rows of the theta-major accumulator. The tile plus a one-cell halo is
// staged in LDS with 16-byte coalesced loads (each accumulator cell is read (kNmsCols+2)/kNmsCols times), so the 3x3
// neighbourhood test never issues a scattered global load. Survivors are flagged in a 64-bit mask per thread, slots are
// reserved with ONE atomic per block, and the keys are rebuilt from the LDS tile.
__global__ __launch_bounds__(kNmsThreads) void sht_nms_kernel(ShtArgs a)
{
	__shared__ __attribute__((aligned(16))) uint16_t s_tile[kNmsCols + 2][kNmsTileRows];
	__shared__ int s_wave[kNmsThreads / 64];
	__shared__ int s_base;
	const int frame = blockIdx.z;
	const int c0 = blockIdx.y * kNmsCols;
	const int base = blockIdx.x * kNmsRows;
	const int t = threadIdx.x;
	const uint16_t* __restrict__ acc = a.acc + (size_t)frame * a.accFrameStride;
	const size_t p = a.accPitch;
	{
		uint4 v[kNmsCols + 2];
		uint4 x = make_uint4(0, 0, 0, 0);
		const int r0 = base - 8 + t * 8;
		const int xr0 = base - 8 + (kNmsThreads + (t & 1)) * 8; // the two extra 8-row groups: threads 2j, 2j+1 fetch them for tile column j
		const int xc = c0 - 1 + (t >> 1);
#pragma unroll
		for (int j = 0; j < kNmsCols + 2; ++j) {
			const int c = c0 - 1 + j;
			v[j] = make_uint4(0, 0, 0, 0);
			if (c >= 0 && c < a.T && r0 >= 0 && r0 < a.accPitch) v[j] = *reinterpret_cast<const uint4*>(acc + (size_t)c * p + r0);
		}
		if (t < 2 * (kNmsCols + 2) && xc >= 0 && xc < a.T && xr0 < a.accPitch) x = *reinterpret_cast<const uint4*>(acc + (size_t)xc * p + xr0);
#pragma unroll
		for (int j = 0; j < kNmsCols + 2; ++j) *reinterpret_cast<uint4*>(&s_tile[j][t * 8]) = v[j];
		if (t < 2 * (kNmsCols + 2)) *reinterpret_cast<uint4*>(&s_tile[t >> 1][(kNmsThreads + (t & 1)) * 8]) = x;
	}
	__syncthreads();

	// column j of the block = tile column j+1; this thread's rows are tile rows [8+8t, 16+8t), neighbours at 7+8t and 16+8t
	auto readCol = [&](int tj, int (&x)[10]) {
		const uint4 w = *reinterpret_cast<const uint4*>(&s_tile[tj][8 + t * 8]);
		x[0] = s_tile[tj][7 + t * 8];
		x[1] = w.x & 0xffffu; x[2] = w.x >> 16; x[3] = w.y & 0xffffu; x[4] = w.y >> 16;
		x[5] = w.z & 0xffffu; x[6] = w.z >> 16; x[7] = w.w & 0xffffu; x[8] = w.w >> 16;
		x[9] = s_tile[tj][16 + t * 8];
	};
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
	uint32_t flags[2] = { 0u, 0u };
#pragma unroll
	for (int j = 0; j < kNmsCols; ++j) {
		readCol(j + 2, nxt);
		int* sl = side[j % 3];
		int* sh = side[(j + 2) % 3];
#pragma unroll
		for (int k = 0; k < 8; ++k) sh[k] = max(max(nxt[k], nxt[k + 1]), nxt[k + 2]);
		const int c = c0 + j;
		const bool nmsCol = (c >= 1 && c <= a.nmsLastCol);
		if (c < a.T) {
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const int r = base + t * 8 + k;
				const int val = mid[k + 1];
				bool pass = (r < a.R) && (val > a.threshold);
				if (nmsCol && r >= 1 && r <= a.R - 2) {
					const int nb = max(max(sl[k], sh[k]), max(mid[k], mid[k + 2]));
					pass = pass && (nb <= val);
				}
				flags[j >> 2] |= (pass ? 1u : 0u) << ((j & 3) * 8 + k);
			}
		}
#pragma unroll
		for (int k = 0; k < 10; ++k) mid[k] = nxt[k];
	}
	const int cnt = __popc(flags[0]) + __popc(flags[1]);
	int incl = cnt;
	const int lane = t & 63, wave = t >> 6;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int n = __shfl_up(incl, o);
		if (lane >= o) incl += n;
	}
	if (lane == 63) s_wave[wave] = incl;
	__syncthreads();
	int wbase = 0, total = 0;
#pragma unroll
	for (int k = 0; k < kNmsThreads / 64; ++k) {
		const int n = s_wave[k];
		if (k < wave) wbase += n;
		total += n;
	}
	if (total == 0) return; // uniform
	if (t == 0) s_base = atomicAdd(&a.lineCounts[frame], total);
	__syncthreads();
	size_t pos = (size_t)s_base + wbase + (incl - cnt);
	uint64_t* __restrict__ dst = a.lineKeys + (size_t)frame * a.lineCap;
	const uint64_t frameTag = (uint64_t)(a.frames - 1 - frame) << (a.strengthBits + a.cellBits);
	const uint32_t cellMask = (1u << a.cellBits) - 1u;
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		uint32_t f = flags[h];
		while (f) {
			const int b = __ffs(f) - 1;
			f &= f - 1;
			const int j = h * 4 + (b >> 3), k = b & 7;
			const uint32_t val = s_tile[j + 1][8 + t * 8 + k];
			const uint32_t cell = (uint32_t)(base + t * 8 + k) * (uint32_t)a.T + (uint32_t)(c0 + j);
			if (pos < a.lineCap) dst[pos] = frameTag | ((uint64_t)val << a.cellBits) | (uint64_t)(cellMask - cell);
			++pos;
		}
	}
}

// Key slots [min(count, lineCap), lineCap) of every frame are zeroed (a zero key sorts last: every real key carries a strength > 0).
constexpr int kPadThreads = 256;
constexpr int kPadSlots = 8;
__global__ __launch_bounds__(kPadThreads) void sht_pad_keys_kernel(uint64_t* __restrict__ keys, const int* __restrict__ counts, size_t lineCap)
{
	const int frame = blockIdx.y;
	const size_t used = (size_t)max(counts[frame], 0);
	const size_t i0 = ((size_t)blockIdx.x * kPadThreads + threadIdx.x) * kPadSlots;
	if (i0 + kPadSlots <= used) return;
	uint64_t* __restrict__ k = keys + (size_t)frame * lineCap;
#pragma unroll
	for (int j = 0; j < kPadSlots; ++j) {
		const size_t i = i0 + j;
		if (i >= used && i < lineCap) k[i] = 0ull;
	}
}

struct LineOut { float rho; float theta; int32_t strength; int32_t row; int32_t col; };

// After the global descending sort the lines of frame f start at sum_{g<f} min(count_g, lineCap).
__global__ __launch_bounds__(256) void sht_decode_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ counts, size_t lineCap, int T, int barrier,
                                                         float thetaStep, int maxLines, int cellBits, int strengthBits, LineOut* __restrict__ lines, size_t outCap)
{
	const int frame = blockIdx.y;
	size_t off = 0;
	for (int g = 0; g < frame; ++g) {
		const size_t cg = (size_t)max(counts[g], 0);
		off += cg < lineCap ? cg : lineCap;
	}
	size_t n = (size_t)max(counts[frame], 0);
	if (n > lineCap) n = lineCap;
	if (maxLines > 0 && n > (size_t)maxLines) n = (size_t)maxLines;
	if (n > outCap) n = outCap;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t k = keys[off + i];
	const uint32_t cellMask = (1u << cellBits) - 1u;
	const uint32_t cell = cellMask - (uint32_t)(k & cellMask);
	const int row = (int)(cell / (uint32_t)T), col = (int)(cell - (uint32_t)row * (uint32_t)T);
	LineOut o;
	o.rho = (float)(barrier - row);              // static_cast<float>(barrier - row), houghsht.cxx:661
	o.theta = __fmul_rn((float)col, thetaStep);  // col * theta (f32), houghsht.cxx:662
	o.strength = (int32_t)((k >> cellBits) & ((1u << strengthBits) - 1u));
	o.row = row; o.col = col;
	lines[(size_t)frame * outCap + i] = o;
}

// CompVHoughSht::toCartesian (houghsht.cxx:566-589) on the device line arrays.  cos(theta) and 1/sin(theta) come from HOST
// tables indexed by the line's theta column (libm cosf/sinf, as the reference; device trig would not be bit-identical);
// the kernel only multiplies and subtracts, with single correctly-rounded operations.
__global__ __launch_bounds__(256) void sht_cartesian_kernel(const LineOut* __restrict__ lines, const int* __restrict__ counts, size_t lineCap, const float* __restrict__ cosT,
                                                            const float* __restrict__ invSinT, float widthF, float r, float4* __restrict__ out)
{
	// rho - W*a must stay a rounded product followed by a rounded subtraction, as on the CPU: the library is built with
	// -ffp-contract=off (the __f*_rn intrinsics are plain operators that a later FMA contraction would still fuse)
	const int frame = blockIdx.y;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	size_t n = (size_t)max(counts[frame], 0);
	if (n > lineCap) n = lineCap;
	if (i >= n) return;
	const LineOut l = lines[(size_t)frame * lineCap + i];
	float4 o;
	if (l.theta == 0.f) o = make_float4(l.rho, r, l.rho, -r); // perfect vertical line
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

hipError_t launch_sht_compact(const ShtArgs& a, int frames, hipStream_t stream)
{
	// a.edgeCounts / a.lineCounts were zeroed by the caller (one fill per step for every counter of the plan, api.cpp)
	const size_t nwords = (size_t)a.H * a.wb;
	const size_t perBlock = (size_t)kCompactThreads * kCompactWords;
	dim3 grid((unsigned)((nwords + perBlock - 1) / perBlock), frames);
	hipLaunchKernelGGL(sht_compact_kernel, grid, dim3(kCompactThreads), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_sht_vote(const ShtArgs& a, int frames, hipStream_t stream)
{
	const int tg = a.thetaPerGroup == 4 ? 4 : 2;
	const bool sc = (tg == 2) && (a.barrier < 8192);
	const size_t lds = sht_vote_lds_bytes(a.R, tg);
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	const void* fn = tg == 4 ? reinterpret_cast<const void*>(sht_vote_kernel<4, false>)
	               : sc ? reinterpret_cast<const void*>(sht_vote_kernel<2, true>) : reinterpret_cast<const void*>(sht_vote_kernel<2, false>);
	// the opt-in to > 64 KB of dynamic LDS is a per-device function attribute: remember it per device (one process may own several)
	static size_t attr_lds[64][3] = {};
	const int slot = tg == 4 ? 0 : (sc ? 1 : 2);
	int dev = 0;
	(void)hipGetDevice(&dev);
	dev = (dev >= 0 && dev < 64) ? dev : 0;
	if (lds > attr_lds[dev][slot]) {
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
		attr_lds[dev][slot] = lds;
	}
	if (a.shards > 1) {
		hipError_t e = hipMemsetAsync(a.acc, 0, sizeof(uint16_t) * a.accFrameStride * frames, stream);
		if (e != hipSuccess) return e;
	}
	const int groups = (a.T + tg - 1) / tg;
	const int fx = (frames + 7) / 8, subs = (fx + kVoteFramesInL2 - 1) / kVoteFramesInL2;
	dim3 grid(8 * subs * kVoteFramesInL2 * groups, a.shards); // see the XCD-aware (frame, theta group) decoding in the kernel
	static const int threads = [] { const char* e = getenv("COMPVHIP_SHT_VOTE_THREADS"); const int v = e ? atoi(e) : 0; return (v >= 64 && v <= 1024 && (v % 64) == 0) ? v : kShtVoteThreads; }(); // tuning knob
	if (tg == 4) hipLaunchKernelGGL((sht_vote_kernel<4, false>), grid, dim3(threads), lds, stream, a);
	else if (sc) hipLaunchKernelGGL((sht_vote_kernel<2, true>), grid, dim3(threads), lds, stream, a);
	else hipLaunchKernelGGL((sht_vote_kernel<2, false>), grid, dim3(threads), lds, stream, a);
	return hipGetLastError();
}

hipError_t launch_sht_nms(const ShtArgs& a, int frames, hipStream_t stream)
{
	dim3 grid((a.R + kNmsRows - 1) / kNmsRows, (a.T + kNmsCols - 1) / kNmsCols, frames);
	hipLaunchKernelGGL(sht_nms_kernel, grid, dim3(kNmsThreads), 0, stream, a);
	// unused key slots must sort last: zero only the slots past each frame's count (the whole array used to be zero-filled before the
	// NMS -- 16.8 MB per 32-frame step at 4K for 12 % unused slots)
	dim3 pgrid((unsigned)((a.lineCap + kPadThreads * kPadSlots - 1) / (kPadThreads * kPadSlots)), frames);
	hipLaunchKernelGGL(sht_pad_keys_kernel, pgrid, dim3(kPadThreads), 0, stream, a.lineKeys, a.lineCounts, a.lineCap);
	return hipGetLastError();
}

// one descending radix sort over the key slots of all frames (rocPRIM device primitive)
hipError_t sht_sort_keys(void* temp, size_t& tempBytes, const uint64_t* keysIn, uint64_t* keysOut, size_t lineCap, int frames, int keyBits,
                         hipStream_t stream)
{
	// 10-bit digits: the 40-bit key of the 4K benchmark takes 4 onesweep passes instead of the 5 of the library's 8-bit default
	using Onesweep = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 10>, 10, rocprim::block_radix_rank_algorithm::match>;
	using Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, Onesweep>;
	return rocprim::radix_sort_keys_desc<Config>(temp, tempBytes, keysIn, keysOut, lineCap * (size_t)frames, 0u, (unsigned int)keyBits, stream);
}

hipError_t launch_sht_decode(const uint64_t* keys, const int* counts, size_t lineCap, int frames, int T, int barrier, float thetaStep,
                             int maxLines, int cellBits, int strengthBits, void* lines, size_t outCap, hipStream_t stream)
{
	size_t n = lineCap < outCap ? lineCap : outCap;
	if (maxLines > 0 && (size_t)maxLines < n) n = (size_t)maxLines;
	if (n == 0) return hipSuccess;
	dim3 grid((unsigned)((n + 255) / 256), frames);
	hipLaunchKernelGGL(sht_decode_kernel, grid, dim3(256), 0, stream, keys, counts, lineCap, T, barrier, thetaStep, maxLines, cellBits, strengthBits,
	                   reinterpret_cast<LineOut*>(lines), outCap);
	return hipGetLastError();
}

hipError_t launch_sht_cartesian(const void* lines, const int* counts, size_t lineCap, int frames, const float* cosT, const float* invSinT, float widthF,
                                float r, float* out, hipStream_t stream)
{
	if (!lineCap) return hipSuccess;
	dim3 grid((unsigned)((lineCap + 255) / 256), frames);
	hipLaunchKernelGGL(sht_cartesian_kernel, grid, dim3(256), 0, stream, reinterpret_cast<const LineOut*>(lines), counts, lineCap, cosT, invSinT, widthF, r,
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
