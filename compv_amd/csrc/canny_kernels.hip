// canny_kernels.hip -- Canny (Sobel gradient -> NMS -> hysteresis) for gfx950, hand-written HIP.
//
// Replaces, behind compvhip_canny_u8 / compvhip_plan_canny (include/compv_hip.h):
//   CompVEdgeDeteCanny::process            core/features/edges/compv_core_feature_canny_dete.cxx:123-331
//   nms_gather / nms_apply / hysteresis    ...canny_dete.cxx:334-528 and the row leaves :566-680,
//   CompVCannyNMSGatherRow_16mpw_Intrin_AVX2 / CompVCannyHysteresisRow_16mpw_Intrin_SSE2 (intrin/x86/*)
//
// HBM layout per frame:  in  u8 [H][S]                 (read once, 1 B/px)
//                        out u8 [H][So]  {0,0xff}      (written once, 1 B/px)
//                        E,U bitmasks u32 [H][wb]      (1 bit/px each: E = edge so far, U = weak but unresolved)
//
// Kernel 1: the tile kernel -- gradient, NMS, weak / strong classification; writes E (strong seeds), U (weak, unresolved) and the bytes of E:
//   canny_swar_tile_kernel (canny_swar_kernels.hip), kernel sizes 3 and 5.  (The first-generation kernel -- one wave per 512x64 tile, 32-bit
//   arithmetic, dense NMS, 512-bit carry-chain flood inside the tile -- served kernel size 5 until round 5: 0.50 ms per 32 x 4K against 0.41.)
// Kernel 2 (canny_resolve_kernel): the hysteresis, on the 1-bit masks only (0.25 B/px): one workgroup per 64-row band floods E into U
//   (register-resident column sweeps); repeated (device flags, no data-dependent host work) until no band changes.  The
//   fixed point "all pixels with g_nms > tLow 8-connected to a pixel with g_nms > tHigh" is unique, hence
//   bit-exact whatever the propagation order (the reference's own multithreaded bands race the same way).
#include "stencil.hpp"
#include "kernels.hpp"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace compvhip {

// ---------------------------------------------------------------------------------------------------------------
// Kernel 2: cross-tile hysteresis on the bit masks.  One workgroup = one band of kBandH rows x up to kBandWords
// 32-px words.  E rows carry a one-row / one-word halo (read-only context owned by neighbouring bands).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kResolveThreads) void canny_resolve_kernel(ResolveArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	// round r only runs if round r-1 changed something (rounds are enqueued speculatively, see host code)
	if (a.round > 0 && a.flags[a.round - 1] == 0) return;

	const int frame = blockIdx.z;
	const int band = blockIdx.y;
	const int chunk = blockIdx.x;
	const int wb = a.wb;
	const int w0 = chunk * kBandWords;               // first word of this chunk
	const int cw = min(kBandWords, wb - w0);          // words in this chunk
	const int y0 = band * kBandH;
	const int rows = min(kBandH, a.H - y0);
	const int ew = cw + 2;                            // E row pitch in LDS (halo word each side)
	uint32_t* sE = smem;                              // (rows+2) x ew

	uint32_t* __restrict__ gE = a.ebits + (size_t)frame * a.bitsFrameStride;
	uint32_t* __restrict__ gU = a.ubits + (size_t)frame * a.bitsFrameStride;
	const int tid = threadIdx.x;

	// Per-workgroup change flags, four generations (round & 3): a workgroup whose 3x3 neighbourhood of (band, chunk) cells changed
	// nothing in the previous round is at its fixed point already -- its E/U words and its halo are what it last saw -- and only
	// records "no change".  Every workgroup that gets here writes its own cell, so the buffer needs no clearing.
	const int nb = gridDim.y, nc = gridDim.x;
	const size_t cells = (size_t)gridDim.z * nb * nc;
	uint8_t* const mine = a.dirty + (size_t)(a.round & 3) * cells + ((size_t)frame * nb + band) * nc + chunk;
	if (a.round > 0) {
		const uint8_t* prev = a.dirty + (size_t)((a.round - 1) & 3) * cells + (size_t)frame * nb * nc;
		// (nine unconditional loads -- clamped cells repeat a neighbour that is read anyway -- in flight together: with a bounds test in
		// front of each, the compiler waits for every byte before it fetches the next)
		int any = 0;
#pragma unroll
		for (int db = -1; db <= 1; ++db)
#pragma unroll
			for (int dc = -1; dc <= 1; ++dc) any |= prev[min(max(band + db, 0), nb - 1) * nc + min(max(chunk + dc, 0), nc - 1)];
		if (!any) { if (tid == 0) *mine = 0; return; } // uniform
		// This band reached its fixed point the last time it ran: only NEW edge pixels in its halo can promote anything, and only through a
		// candidate (U) pixel of its border that touches one of them.  Look at the border first -- 6 rows and 6 word columns instead of the whole
		// band: in round 1 nearly every neighbour has changed something, but few of those changes touch a candidate across the border.
		int hit = 0;
		for (int i = tid; i < 2 * cw; i += kResolveThreads) {          // top and bottom rows
			const int bottom = i >= cw, c = bottom ? i - cw : i;
			const int r = bottom ? rows - 1 : 0, yh = bottom ? y0 + rows : y0 - 1;   // border row of the band, halo row outside it
			if (yh < 0 || yh >= a.H) continue;
			const size_t gi = (size_t)(y0 + r) * wb + w0 + c;
			const uint32_t cand = gU[gi] & ~gE[gi];
			if (!cand) continue;
			const uint32_t* hrow = gE + (size_t)yh * wb + w0 + c;
			const uint32_t hc = hrow[0];
			uint32_t nbits = hc | (hc << 1) | (hc >> 1);
			if (w0 + c > 0) nbits |= hrow[-1] >> 31;
			if (w0 + c + 1 < wb) nbits |= hrow[1] << 31;
			hit |= (cand & nbits) != 0;
		}
		for (int i = tid; i < 2 * rows; i += kResolveThreads) {        // leftmost and rightmost word columns (chunked rows only)
			const int right = i >= rows, r = right ? i - rows : i;
			const int wc = right ? w0 + cw - 1 : w0, wh = right ? w0 + cw : w0 - 1;    // border word column, halo word column
			if (wh < 0 || wh >= wb) continue;
			const size_t gi = (size_t)(y0 + r) * wb + wc;
			const uint32_t cand = (gU[gi] & ~gE[gi]) & (right ? 0x80000000u : 1u);
			if (!cand) continue;
			uint32_t h = gE[(size_t)(y0 + r) * wb + wh];
			if (y0 + r > 0) h |= gE[(size_t)(y0 + r - 1) * wb + wh];
			if (y0 + r + 1 < a.H) h |= gE[(size_t)(y0 + r + 1) * wb + wh];
			hit |= right ? (h & 1u) != 0 : (h >> 31) != 0;
		}
		if (!__syncthreads_or(hit)) { if (tid == 0) *mine = 0; return; }
	}

	// A thread owns up to kResolveRows CONSECUTIVE rows of one word column: k = tid % cw, row group tid / cw.  Its U words and its own
	// E words live in registers; the LDS copy of E serves the neighbours (the rows above / below the group, bits 31 / 0 of the word
	// columns left / right).  A sweep down then up the group's rows carries a vertical or diagonal chain through all of them in ONE
	// iteration, words without candidates cost one test, and a word whose candidates do not touch the word border needs no LDS access.
	const int k = tid % cw, grp = tid / cw;
	const int nGroups = kResolveThreads / cw;                         // >= 8: cw <= kBandWords = 64
	const int rpg = (rows + nGroups - 1) / nGroups;                   // <= kResolveRows
	const int r0 = grp * rpg;
	const int n = (grp < nGroups) ? max(min(rpg, rows - r0), 0) : 0;  // rows of this thread
	uint32_t u[kResolveRows], e[kResolveRows + 2];
	const uint32_t ewInv = (1u << 19) / (uint32_t)ew + 1u;   // i / ew == (i * ewInv) >> 19 for every i < 66 * 66, 3 <= ew <= 66 (verified exhaustively)
	// The band's E words with their halo go to the LDS: up to nine words per thread, loaded unconditionally from clamped addresses and masked afterwards
	// (`if (inside) v = gE[...]` made every word wait for the one before), and ISSUED TOGETHER WITH the thread's U words: the test "anything unresolved in
	// this band?" needs the U words only, but nearly every band of a real frame passes it, and two memory latencies one after the other per workgroup
	// cost 1.6 us per launch (round 6: 59 -> 54 us for the three rounds of a 32 x 4K step).
	const int nE = (rows + 2) * ew;
	constexpr int kEPer = ((kBandH + 2) * (kBandWords + 2) + kResolveThreads - 1) / kResolveThreads;   // 9: (kBandH + 2) x (kBandWords + 2) words over the workgroup
	uint32_t ev[kEPer];
#pragma unroll
	for (int q = 0; q < kEPer; ++q) {
		const int i = min(tid + q * kResolveThreads, nE - 1);
		const int r = (int)(((uint32_t)i * ewInv) >> 19), c = i - r * ew;
		const int y = y0 - 1 + r, w = w0 - 1 + c;
		const bool ok = (y >= 0 && y < a.H && w >= 0 && w < wb);
		ev[q] = gE[(size_t)min(max(y, 0), a.H - 1) * wb + min(max(w, 0), wb - 1)];
		if (!ok) ev[q] = 0u;
	}
	int haveU = 0;
#pragma unroll
	for (int j = 0; j < kResolveRows; ++j) {
		u[j] = (j < n) ? gU[(size_t)(y0 + r0 + j) * wb + w0 + k] : 0u;
		haveU |= (u[j] != 0);
	}
	if (!__syncthreads_or(haveU)) { if (tid == 0) *mine = 0; return; }  // nothing unresolved in this band
#pragma unroll
	for (int q = 0; q < kEPer; ++q) {
		const int i = tid + q * kResolveThreads;
		if (i < nE) sE[i] = ev[q];
	}
	__syncthreads();
	uint32_t* const col = sE + (size_t)(n > 0 ? r0 : 0) * ew + k + 1;   // LDS word of (row r0 - 1, column k); row j of the group is col[(j + 1) * ew]
#pragma unroll
	for (int j = 0; j < kResolveRows + 2; ++j) e[j] = (j <= n + 1) ? col[j * ew] : 0u;

	// one word: candidates next to an edge pixel are promoted, then the promotion runs along the horizontal candidate runs of the word
	auto visit = [&](uint32_t uw, uint32_t up, uint32_t& ce, uint32_t dn, uint32_t* lds, int& changed) {
		const uint32_t cand = uw & ~ce;
		if (!cand) return;
		const uint32_t c = up | ce | dn;
		uint32_t nbits = c | (c << 1) | (c >> 1);
		if (cand & 0x80000001u) {   // bit 0 / 31 of the neighbouring word columns, rows above .. below
			const uint32_t L = lds[-ew - 1] | lds[-1] | lds[ew - 1], R = lds[-ew + 1] | lds[1] | lds[ew + 1];
			nbits |= (L >> 31) | (R << 31);
		}
		uint32_t add = cand & nbits;
		if (!add) return;
		const uint32_t upr = cand & ~(cand + add);
		const uint32_t rc = __brev(cand), ra = __brev(add);
		const uint32_t dnr = __brev(rc & ~(rc + ra));
		add |= upr | dnr;
		ce |= add;
		*lds = ce;      // single owner per word; neighbours may read old or new (monotone)
		changed = 1;
	};
	for (;;) {
		int changed = 0;
		e[0] = col[0];                       // the rows of the groups above / below may have moved
		if (n > 0) {
#pragma unroll
			for (int j = 0; j < kResolveRows; ++j)
				if (j == n - 1) e[j + 2] = col[(j + 2) * ew];
		}
#pragma unroll
		for (int j = 0; j < kResolveRows; ++j)
			if (j < n) visit(u[j], e[j], e[j + 1], e[j + 2], col + (j + 1) * ew, changed);
#pragma unroll
		for (int j = kResolveRows - 2; j >= 0; --j)
			if (j < n) visit(u[j], e[j], e[j + 1], e[j + 2], col + (j + 1) * ew, changed);
		if (!__syncthreads_or(changed)) break;
	}

	// write back: promoted = E_new & U_old
	int wrote = 0;
	uint8_t* __restrict__ out = a.out ? a.out + (size_t)frame * a.outFrameStride : nullptr;
#pragma unroll
	for (int j = 0; j < kResolveRows; ++j) {
		if (j >= n || !u[j]) continue;
		uint32_t p = u[j] & e[j + 1];
		if (!p) continue;
		const size_t gi = (size_t)(y0 + r0 + j) * wb + w0 + k;
		gE[gi] = e[j + 1];
		gU[gi] = u[j] & ~p;
		wrote = 1;
		if (a.out) { // byte map patched in place
			uint8_t* orow = out + (size_t)(y0 + r0 + j) * a.So + (size_t)(w0 + k) * 32;
			while (p) {
				const int b = __ffs(p) - 1;
				p &= p - 1;
				orow[b] = 0xff;
			}
		}
	}
	const int w = __syncthreads_or(wrote);
	if (tid == 0) {
		*mine = w ? 1 : 0;
		if (w) a.flags[a.round] = 1; // benign race: every writer stores 1
	}
}

// ---------------------------------------------------------------------------------------------------------------
// PERCENT_OF_MEAN thresholds: sum of pixels per frame (CompVMathUtils::sum<u8,u32>, canny_dete.cxx:243), then
// mean/thresholds exactly as :252-266.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_sum_kernel(const uint8_t* __restrict__ in, int W, int H, int S, size_t frameStride,
                                                        unsigned int* __restrict__ sums)
{
	const int frame = blockIdx.y;
	const uint8_t* p = in + (size_t)frame * frameStride;
	unsigned int acc = 0;
	const int wq = W >> 2;
	for (int y = blockIdx.x; y < H; y += gridDim.x) {
		const uint8_t* row = p + (size_t)y * S;
		for (int q = threadIdx.x; q < wq; q += blockDim.x) {
			const uint32_t v = reinterpret_cast<const uint32_t*>(row)[q];
			acc += (v & 0xff) + ((v >> 8) & 0xff) + ((v >> 16) & 0xff) + (v >> 24);
		}
		for (int x = (wq << 2) + threadIdx.x; x < W; x += blockDim.x) acc += row[x];
	}
	for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
	if ((threadIdx.x & 63) == 0) atomicAdd(&sums[frame * kFrameSlot], acc);
}

__global__ void mean_thresholds_kernel(const unsigned int* __restrict__ sums, int W, int H, float fLow, float fHigh, int2* __restrict__ thr, int frames)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= frames) return;
	unsigned int mean = (sums[f * kFrameSlot] / (unsigned int)(W * H)) & 0xffu; // static_cast<uint8_t>
	mean = mean < 1u ? 1u : mean;
	unsigned int lo = (unsigned int)(int)__fmul_rn((float)mean, fLow) & 0xffffu;   // static_cast<uint16_t>(mean * f)
	unsigned int hi = (unsigned int)(int)__fmul_rn((float)mean, fHigh) & 0xffffu;
	lo = lo < 1u ? 1u : lo;
	const unsigned int t = lo + 2u;
	hi = (t > hi ? t : hi) & 0xffffu;
	thr[f] = make_int2((int)lo, (int)hi);
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
// The SWAR + candidate-list kernel of canny_swar_kernels.hip writes E (strong seeds), U (weak, unresolved) and the edge bytes of E for both kernel
// sizes; canny_resolve_kernel does the hysteresis.
hipError_t launch_canny_tiles(const CannyArgs& a0, int frames, bool gap, hipStream_t stream)
{
	if (a0.ksize != 3 && a0.ksize != 5) return hipErrorInvalidValue;
	return launch_canny_tiles_swar(a0, frames, gap, stream);
}

size_t canny_resolve_dirty_bytes(int H, int wb, int frames)
{
	const size_t bands = (H + kBandH - 1) / kBandH, chunks = (wb + kBandWords - 1) / kBandWords;
	return 4 * bands * chunks * (size_t)frames;
}

size_t resolve_lds_bytes()
{
	return (size_t)(kBandH + 2) * (kBandWords + 2) * sizeof(uint32_t);
}

hipError_t launch_canny_resolve(const ResolveArgs& a, int frames, hipStream_t stream)
{
	// the opt-in to > 64 KB of dynamic LDS is a per-device function attribute: remember it per device (one process may own several)
	static bool attr_set[64] = {};
	const size_t lds = resolve_lds_bytes();
	int dev = 0;
	(void)hipGetDevice(&dev);
	dev = (dev >= 0 && dev < 64) ? dev : 0;
	if (!attr_set[dev]) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(canny_resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
		attr_set[dev] = true;
	}
	const int bands = (a.H + kBandH - 1) / kBandH;
	const int chunks = (a.wb + kBandWords - 1) / kBandWords;
	dim3 grid(chunks, bands, frames);
	hipLaunchKernelGGL(canny_resolve_kernel, grid, dim3(kResolveThreads), lds, stream, a);
	return hipGetLastError();
}

hipError_t launch_mean_thresholds(const uint8_t* in, int W, int H, int S, size_t frameStride, int frames, float fLow, float fHigh,
                                  unsigned int* sums, int2* thr, hipStream_t stream)
{
	hipError_t e = hipMemsetAsync(sums, 0, sizeof(unsigned int) * frames * kFrameSlot, stream);
	if (e != hipSuccess) return e;
	dim3 grid(min(H, 64), frames);
	hipLaunchKernelGGL(frame_sum_kernel, grid, dim3(256), 0, stream, in, W, H, S, frameStride, sums);
	hipLaunchKernelGGL(mean_thresholds_kernel, dim3((frames + 63) / 64), dim3(64), 0, stream, sums, W, H, fLow, fHigh, thr, frames);
	return hipGetLastError();
}

} // namespace compvhip
