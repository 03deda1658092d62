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
//   line keys       u64 [frames][lineCap]  frameTag | strength << cellBits | (cellMask - cell), cell = row*T + col: unique keys, one
//                                          descending radix sort over all frames
//
// Voting: rho = (x*cosQ[t] + y*sinQ[t]) >> 16 (int32, arithmetic shift), acc[barrier - rho][t]++ for every edge and
// every t -- E*T scattered increments, the whole cost of the reference's SHT: sht_tiles_kernels.hip (lane = theta over image tiles).
// This file: foreign edge maps -> bit masks, sht_nms_kernel, the key sort, sht_decode_kernel, sht_cartesian_kernel, the accumulator export.
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
// NMS + threshold -> lines, in three small steps that make the emission order DETERMINISTIC (accumulator rows, then columns, ascending:
// nms_apply's own order, houghsht.cxx:546-562), so that the sort only has to order by strength:
//   sht_nms_kernel   LDS-tiled 3x3 test; the survivors of 8 theta columns x 8 rho rows per thread leave as one flag byte per (row, column
//                    group): flag planes [frame][column group][row], 8-byte coalesced stores, no atomics;
//   sht_rank_kernel  survivors per row (popcounts of the row's flag bytes), exclusive scan inside chunks of 1024 rows -> rowBase[frame][row]
//                    and the chunk totals (the emit kernel adds the few totals in front of a row's chunk);
//   sht_emit_kernel  one workgroup per 64 rows: their survivors in (row, column) order -> key = frameTag | strength, value = cell (row * T + col)
//                    at slot chunk base + rowBase + in-row offset of the frame's key / value arrays (staged in the LDS, coalesced stores).
// A stable descending radix sort of the (key, value) pairs then gives frame-major, strength-descending, (row, col)-ascending order with
// frameBits + strengthBits key bits (18 at 4K x 32 frames: two 10-bit onesweep passes; the unique 40-bit keys of rounds 1-2, which carried
// the cell because the slots were handed out by atomics in arrival order, took four).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kNmsThreads = 128;
constexpr int kNmsCols = 8;                  // theta columns per block
constexpr int kNmsRows = kNmsThreads * 8;    // rho rows per block: 8 per thread

// One block owns kNmsCols theta columns x kNmsRows rho rows of the theta-major accumulator; a thread 8 consecutive rows of all columns
// (+ one column either side): ten 16-byte coalesced loads straight into registers (each accumulator cell is read (kNmsCols+2)/kNmsCols
// times), the rows above / below its eight with two 2-byte loads per column (lines its neighbours fetch anyway): no LDS tile, no
// barrier -- the waves of the launch are independent and hide each other's load latency (the LDS-tiled version of rounds 1-2 ran
// load -> barrier -> compute per block at 3.5 waves per SIMD: 45 us for 87 MB).
// A thread's survivors are one byte per row (bit j = column c0 + j).
__global__ __launch_bounds__(kNmsThreads) void sht_nms_kernel(ShtArgs a)
{
	const int frame = blockIdx.z;
	const int c0 = blockIdx.y * kNmsCols;
	const int base = blockIdx.x * kNmsRows;
	const int t = threadIdx.x;
	// Two thirds of the accumulator's cells lie outside every tile's rho window of their theta (no pixel of the image maps there): they are
	// never written, stay zero, and their flag bytes (zeroed when the plan was made) are never written either.
	const int2 reach = a.nmsRange[blockIdx.y];
	if (base >= reach.y || base + kNmsRows <= reach.x) return; // uniform
	const uint16_t* __restrict__ acc = a.acc + (size_t)frame * a.accFrameStride;
	const size_t p = a.accPitch;
	const int r0 = base + t * 8;
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
	// flag plane of this column group: rows base + 8 t .. + 7, one 8-byte store per thread (a wave writes 512 consecutive bytes)
	uint8_t* __restrict__ plane = a.nmsFlags + ((size_t)frame * a.nmsGroups + blockIdx.y) * a.nmsRows;
	*reinterpret_cast<uint2*>(plane + base + t * 8) = make_uint2(flags[0], flags[1]);
}

// rowBase[frame][row] = survivors in the rows above INSIDE the row's chunk of kRankThreads rows; chunkTotals[frame][chunk] = survivors of
// the chunk.  One workgroup per (chunk, frame): a single coalesced pass, one block scan.
constexpr int kRankThreads = 1024;
__global__ __launch_bounds__(kRankThreads) void sht_rank_kernel(ShtArgs a)
{
	__shared__ int s_wave[kRankThreads / 64];
	const int frame = blockIdx.y;
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int r = blockIdx.x * kRankThreads + t;
	const uint8_t* __restrict__ planes = a.nmsFlags + (size_t)frame * a.nmsGroups * a.nmsRows;
	// survivors of the row's earlier column groups, one u16 per (group, row): the emit kernel starts every flag byte at rowBase + this offset
	uint16_t* __restrict__ offs = a.nmsOffs + (size_t)frame * a.nmsGroups * a.nmsRows;
	int cnt = 0;
	if (r < a.R) for (int g = 0; g < a.nmsGroups; ++g) {
		offs[(size_t)g * a.nmsRows + r] = (uint16_t)cnt;
		cnt += __popc((uint32_t)planes[(size_t)g * a.nmsRows + r]);
	}
	int incl = cnt;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int n = __shfl_up(incl, o);
		if (lane >= o) incl += n;
	}
	if (lane == 63) s_wave[wave] = incl;
	__syncthreads();
	int wbase = 0, total = 0;
#pragma unroll
	for (int k = 0; k < kRankThreads / 64; ++k) {
		const int n = s_wave[k];
		if (k < wave) wbase += n;
		total += n;
	}
	if (r < a.R) a.rowBase[(size_t)frame * a.nmsRows + r] = (uint32_t)(wbase + incl - cnt);
	if (t == 0) a.chunkTotals[frame * gridDim.x + blockIdx.x] = total;
}

// One workgroup = kEmitRows consecutive accumulator rows, all column groups (thread = (row, group phase)): the survivors of those rows
// occupy ONE contiguous slot range of the frame's key / value arrays (rows ascending, columns ascending), so they are put in place in
// the LDS first and leave as coalesced stores (a thread storing its own few items would touch 64 different cache lines per instruction).
// Every accumulator read is independent of every other.  Block 0 of every frame also publishes the frame's line count.
constexpr int kEmitThreads = 256;
constexpr int kEmitRows = 64;
constexpr int kEmitStage = 2048;   // slots staged in the LDS; denser row blocks store directly
__global__ __launch_bounds__(kEmitThreads) void sht_emit_kernel(ShtArgs a, int chunks)
{
	__shared__ uint32_t s_keys[kEmitStage], s_vals[kEmitStage];
	const int frame = blockIdx.y;
	const int r0 = blockIdx.x * kEmitRows;
	const int r = r0 + (threadIdx.x & (kEmitRows - 1)), phase = threadIdx.x / kEmitRows;
	const int* __restrict__ ct = a.chunkTotals + frame * chunks;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		int n = 0;
		for (int k = 0; k < chunks; ++k) n += ct[k];
		a.lineCounts[frame] = n;
	}
	// slot range of this block: [first, last) (kRankThreads is a multiple of kEmitRows: the block's rows lie in one chunk)
	size_t chunkBase = 0;
	for (int k = 0; k < r0 / kRankThreads; ++k) chunkBase += (size_t)ct[k];
	const uint32_t* __restrict__ rb = a.rowBase + (size_t)frame * a.nmsRows;
	const int rEnd = min(r0 + kEmitRows, a.R);
	const size_t first = chunkBase + rb[r0];
	size_t last;
	if (rEnd < a.R && (rEnd % kRankThreads) != 0) last = chunkBase + rb[rEnd];
	else last = chunkBase + (size_t)ct[r0 / kRankThreads];          // the block ends its chunk (or the accumulator)
	const size_t span = last - first;
	const bool staged = span <= (size_t)kEmitStage;
	uint32_t* __restrict__ keys = a.lineKeys + (size_t)frame * a.lineCap;
	uint32_t* __restrict__ vals = a.lineVals + (size_t)frame * a.lineCap;
	if (r < a.R) {
		const size_t rowPos = chunkBase + rb[r];
		const uint32_t frameTag = (uint32_t)(a.frames - 1 - frame) << a.strengthBits;
		for (int g = phase; g < a.nmsGroups; g += kEmitThreads / kEmitRows) {
			const size_t pi = ((size_t)frame * a.nmsGroups + g) * a.nmsRows + r;
			const uint32_t f = a.nmsFlags[pi];
			if (!f) continue;
			const uint16_t* __restrict__ acc = a.acc + (size_t)frame * a.accFrameStride + (size_t)(g * kNmsCols) * a.accPitch + r;
			uint32_t v[kNmsCols];
#pragma unroll
			for (int j = 0; j < kNmsCols; ++j) v[j] = ((f >> j) & 1u) ? (uint32_t)acc[(size_t)j * a.accPitch] : 0u;
			size_t pos = rowPos + (size_t)a.nmsOffs[pi];
			const uint32_t cell0 = (uint32_t)r * (uint32_t)a.T + (uint32_t)(g * kNmsCols);
#pragma unroll
			for (int j = 0; j < kNmsCols; ++j) {
				if ((f >> j) & 1u) {
					if (staged) { s_keys[pos - first] = frameTag | v[j]; s_vals[pos - first] = cell0 + (uint32_t)j; }
					else if (pos < a.lineCap) { keys[pos] = frameTag | v[j]; vals[pos] = cell0 + (uint32_t)j; }
					++pos;
				}
			}
		}
	}
	if (!staged) return; // uniform
	__syncthreads();
	for (size_t i = threadIdx.x; i < span; i += kEmitThreads) {
		const size_t pos = first + i;
		if (pos < a.lineCap) { keys[pos] = s_keys[i]; vals[pos] = s_vals[i]; }
	}
}

// Key slots [min(count, lineCap), lineCap) of every frame are zeroed (a zero key sorts last: every real key carries a strength > 0).
constexpr int kPadThreads = 256;
constexpr int kPadSlots = 8;
__global__ __launch_bounds__(kPadThreads) void sht_pad_keys_kernel(uint32_t* __restrict__ keys, const int* __restrict__ counts, size_t lineCap)
{
	const int frame = blockIdx.y;
	const size_t used = (size_t)max(counts[frame], 0);
	const size_t i0 = ((size_t)blockIdx.x * kPadThreads + threadIdx.x) * kPadSlots;
	if (i0 + kPadSlots <= used) return;
	uint32_t* __restrict__ k = keys + (size_t)frame * lineCap;
#pragma unroll
	for (int j = 0; j < kPadSlots; ++j) {
		const size_t i = i0 + j;
		if (i >= used && i < lineCap) k[i] = 0u;
	}
}

struct LineOut { float rho; float theta; int32_t strength; int32_t row; int32_t col; };

// After the global descending sort the lines of frame f start at sum_{g<f} min(count_g, lineCap).
__global__ __launch_bounds__(256) void sht_decode_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, const int* __restrict__ counts, size_t lineCap,
                                                         int T, int barrier, float thetaStep, int maxLines, int strengthBits, LineOut* __restrict__ lines, size_t outCap)
{
	const int frame = blockIdx.y;
	__shared__ size_t s_off;
	if (threadIdx.x == 0) {   // once per block, not once per thread
		size_t o = 0;
		for (int g = 0; g < frame; ++g) {
			const size_t cg = (size_t)max(counts[g], 0);
			o += cg < lineCap ? cg : lineCap;
		}
		s_off = o;
	}
	__syncthreads();
	const size_t off = s_off;
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

int sht_rank_chunks(int R) { return (R + kRankThreads - 1) / kRankThreads; }
size_t sht_nms_rows(int R) { return (size_t)((R + kNmsRows - 1) / kNmsRows) * kNmsRows; }   // rows of a flag plane (whole NMS blocks)
int sht_nms_groups(int T) { return (T + kNmsCols - 1) / kNmsCols; }

hipError_t launch_sht_nms(const ShtArgs& a, int frames, hipStream_t stream)
{
	dim3 grid((a.R + kNmsRows - 1) / kNmsRows, (a.T + kNmsCols - 1) / kNmsCols, frames);
	hipLaunchKernelGGL(sht_nms_kernel, grid, dim3(kNmsThreads), 0, stream, a);
	const int chunks = (a.R + kRankThreads - 1) / kRankThreads;
	hipLaunchKernelGGL(sht_rank_kernel, dim3(chunks, frames), dim3(kRankThreads), 0, stream, a);
	hipLaunchKernelGGL(sht_emit_kernel, dim3((a.R + kEmitRows - 1) / kEmitRows, frames), dim3(kEmitThreads), 0, stream, a, chunks);
	// unused key slots must sort last: zero the slots past each frame's count (a zero key sorts last: every real key carries a strength > 0)
	dim3 pgrid((unsigned)((a.lineCap + kPadThreads * kPadSlots - 1) / (kPadThreads * kPadSlots)), frames);
	hipLaunchKernelGGL(sht_pad_keys_kernel, pgrid, dim3(kPadThreads), 0, stream, a.lineKeys, a.lineCounts, a.lineCap);
	return hipGetLastError();
}

// one stable descending radix sort over the (key, value) slots of all frames (rocPRIM device primitive)
hipError_t sht_sort_pairs(void* temp, size_t& tempBytes, const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valsIn, uint32_t* valsOut, size_t lineCap,
                          int frames, int keyBits, hipStream_t stream)
{
	// 10-bit digits: the 18-bit key of the 4K benchmark (5 frame bits + 13 strength bits) takes two onesweep passes
	using Onesweep = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 10>, 10, rocprim::block_radix_rank_algorithm::match>;
	using Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, Onesweep>;
	return rocprim::radix_sort_pairs_desc<Config>(temp, tempBytes, keysIn, keysOut, valsIn, valsOut, lineCap * (size_t)frames, 0u, (unsigned int)keyBits, stream);
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
