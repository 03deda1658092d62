// kernels.hpp -- argument blocks and launchers shared between the .hip kernel files and the C-ABI host code (api.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/compv_hip.h" // compvhip_pixfmt

namespace compvhip {

// ---- Canny ------------------------------------------------------------------------------------------------
#ifndef COMPVHIP_BAND_H
#define COMPVHIP_BAND_H 64
#endif
constexpr int kBandH = COMPVHIP_BAND_H;            // rows per resolve band
#ifndef COMPVHIP_BAND_WORDS
#define COMPVHIP_BAND_WORDS 64
#endif
constexpr int kBandWords = COMPVHIP_BAND_WORDS;        // 32-px words per resolve chunk (2048 columns): 0.056 ms per step at 4K, 0.075 ms with 128
constexpr int kResolveThreads = 512;
// Per-frame counters that many workgroups hit with atomics (Sobel gmax, the pixel sum of the mean thresholds) sit one per 128-byte line: the
// L2 serialises the atomics of a line, and the counters of 32 frames side by side are ONE line.
constexpr int kFrameSlot = 32;
constexpr int kResolveRows = kBandH * kBandWords / kResolveThreads; // rows of one word column a resolve thread owns (8)

struct CannyArgs {
	const uint8_t* in;
	uint8_t* out;
	uint32_t* ebits;
	uint32_t* ubits;
	const int2* thrDev;       // per-frame {tLow,tHigh} (PERCENT_OF_MEAN) or nullptr
	size_t inFrameStride, outFrameStride, bitsFrameStride; // elements
	int W, H, S, So;
	int wb;                   // words per bitmask row (= tilesX*16)
	int tilesX, tilesY;
	int tLow, tHigh;
	int simdEnd, cStart;      // quirk Q3 coverage: [1,simdEnd) U [cStart,W-1)
	int blockRows, groups;    // filled by the launcher: workgroup rows per frame, row groups in the launch (XCD-aware map)
	int ksize;                // Sobel kernel size of the gradient: 3 or 5
	int* zero; int nZero;     // counters the STEP needs cleared before its next kernel (edge / line / tile / block counts, round flags): the first nZero / 64 workgroups
	                          // of the tile kernel clear 64 each -- a hipMemsetAsync per step was one more launch on the lane's chain (nullptr: nothing to clear)
};

struct ResolveArgs {
	uint32_t* ebits;
	uint32_t* ubits;
	uint8_t* out;             // byte map to patch with the promoted pixels (nullptr: masks only)
	int* flags;               // flags[round] = 1 when round changed something
	uint8_t* dirty;           // [4][frames][bands][chunks]: which workgroups changed their band in round (r & 3)
	size_t outFrameStride, bitsFrameStride;
	int H, So, wb;
	int round;
};

hipError_t launch_canny_tiles(const CannyArgs& a, int frames, bool gap, hipStream_t stream);
hipError_t launch_canny_tiles_swar(const CannyArgs& a, int frames, bool gap, hipStream_t stream); // kernel size 3 (canny_swar_kernels.hip)
hipError_t launch_canny_resolve(const ResolveArgs& a, int frames, hipStream_t stream);
size_t canny_resolve_dirty_bytes(int H, int wb, int frames);
hipError_t launch_mean_thresholds(const uint8_t* in, int W, int H, int S, size_t frameStride, int frames, float fLow, float fHigh,
                                  unsigned int* sums, int2* thr, hipStream_t stream);

// ---- Sobel / Scharr / Prewitt detector -------------------------------------------------------------------
struct EdgeDeteArgs {
	const uint8_t* in;
	uint8_t* out;
	unsigned int* gmax;       // per frame, one counter per 128-byte line: gmax[frame * kFrameSlot]
	size_t inFrameStride, outFrameStride;
	int W, H, S, So;
	int tilesX, tilesY;
	int blockRows, groups;    // filled by the launcher (XCD-aware map)
};
hipError_t launch_edge_dete(const EdgeDeteArgs& a, int op, int frames, hipStream_t stream);

// ---- pre-processing (grayscale, Otsu) -----------------------------------------------------------------------
struct GrayArgs {
	const uint8_t* in;        // [frames][H][S samples of bpp bytes]
	uint8_t* out;             // [frames][H][So]
	int W, H, S, So;
};
hipError_t launch_gray(const GrayArgs& a, int fmt, int frames, hipStream_t stream);
constexpr int kOtsuMaxChunks = 128; // row chunks (= workgroups) per frame of the histogram kernel
// hist: [frames][kOtsuMaxChunks][256] u32 scratch (partial histograms); otsu: int32 per frame (or null); thr: int2 {tLow,tHigh} per frame (or null)
hipError_t launch_otsu(const uint8_t* in, int W, int H, int S, size_t frameStride, int frames, float fLowFactor, float fHighFactor, uint32_t* hist,
                       int32_t* otsu, void* thr, hipStream_t stream);

// ---- fixed-point separable convolution (Gaussian pre-blur) ---------------------------------------------------
constexpr int kFxpMaxTaps = 15;
struct FxpArgs {
	const uint8_t* in;
	uint8_t* out;
	size_t inFrameStride, outFrameStride;
	int W, H, S, So;
	uint32_t kern[kFxpMaxTaps]; // Q16 weights of this pass
};
hipError_t launch_convlt_fxp(const uint8_t* in, uint8_t* tmp, uint8_t* out, int W, int H, int S, size_t frameStride, int frames, const uint16_t* vtKern,
                             const uint16_t* hzKern, int K, hipStream_t stream);

// integer separable correlation, int16 out (CompVMathConvlt::convlt1<u8|s16, s16, s16>)
struct I16Args {
	const void* in;           // u8 or s16 rows of S elements
	int16_t* out;             // s16 rows of So elements
	int W, H, S, So, K;
	int kern[kFxpMaxTaps];    // taps of this pass
};
hipError_t launch_convlt_i16(const void* in, bool inIsU8, int16_t* tmp, int16_t* out, int W, int H, int S, int So, const int16_t* vtKern, const int16_t* hzKern,
                             int K, hipStream_t stream);

// ---- Hough SHT ---------------------------------------------------------------------------------------------
struct ShtArgs {
	const uint32_t* ebits;    // edge bit masks [frames][H][wb]
	uint32_t* edges;          // per-tile edge lists of every frame: (ly << 16) | lx, edgeCap entries per frame
	int* edgeCounts;          // per frame
	uint16_t* acc;            // [frames][T][accPitch], u16: a cell never exceeds 65535
	const int32_t* sinQ;      // [T]
	const int32_t* cosQ;      // [T]
	uint32_t* lineKeys;       // [frames * lineCap] sort keys: frameTag << strengthBits | strength -- DENSE: frame f's lines follow frame f-1's (min(count, lineCap) each)
	uint32_t* lineVals;       // [frames * lineCap] their accumulator cells: row * T + col
	uint8_t* nmsFlags;        // [frames][nmsGroups][nmsRows] NMS survivors: bit j of byte (group, row) = column 8 group + j
	int nmsRows;              // rows of a flag plane
	int* blockCounts;         // [frames][lineBlocks] NMS survivors per 64 accumulator rows (one row block of sht_lines_kernel); accumulated by sht_nms_kernel (zeroed per step)
	int lineBlocks;
	const int2* nmsRange;     // [nmsGroups] accumulator rows [x, y) the windows of the group's columns (+ one either side) can reach, widened by one row
	int nmsGroups;            // groups of 8 theta columns
	int* lineCounts;          // per frame
	const int* stepFlags;     // asynchronous steps: the hysteresis round flags of this step (device) ...
	int* hostStep;            // ... and where the host wants them: a device-mapped PINNED HOST slot, [0] = the line total, [kFrameSlot .. + 3] = the first four round flags;
	                          // written by one wave of sht_lines_kernel (nullptr: not an asynchronous step) -- a device-to-host copy per step was one more operation on the lane's chain
	int32_t* outCounts;       // the caller's per-frame line counts (device, may be nullptr): written with lineCounts by sht_lines_kernel -- a device copy per step was one more launch
	int* frameTotals;         // [frames * kFrameSlot] NMS survivors per frame, one counter per 128-byte line (sht_nms_kernel adds its row blocks, zeroed per step)
	unsigned int* lineTotal;  // sum over the frames of min(survivors, lineCap) = the key slots in use (written by sht_lines_kernel)
	size_t sortN;             // key slots the sort will cover: sht_lines_kernel zeroes [lineTotal, sortN) (0: nothing to pad -- the sort is sized after the fact)
	size_t bitsFrameStride, edgeCap, accFrameStride, lineCap;
	int W, H, wb;
	int R, T, accPitch, barrier;
	int threshold, nmsLastCol;
	int frames;
	int strengthBits;         // bits of the strength field of a line key (2^strengthBits > 2*max(W,H) >= any cell count)
};
// voting (sht_tiles_kernels.hip): image tiles, lane = theta
struct ShtTileArgs {
	const int32_t* kt;        // [tiles][T]  window constant K of (tile, theta): window row = (K - lx cosQ - ly sinQ) >> 16
	const int32_t* rowBase;   // [tiles][T]  accumulator row of window row 0
	uint8_t* partLo;          // [frames][tiles][Tpad][rwPitch]  theta-major partial accumulators (one window per tile and theta): low bytes of the counts
	uint8_t* partHi;          // same shape: high bytes, valid only for the (tile, theta) columns whose colFlag is set (some count of the column >= 256)
	uint8_t* colFlag;         // [frames][tiles][Tpad]  written by every vote workgroup for its 64 columns
	const int2* reach;        // [T]  accumulator rows [x, y) the tiles' windows of a theta cover (their union's hull)
	int* tileCounts;          // [frames][tiles] edges per tile
	int nx, ny, TW, TH, tiles;  // tile grid; TW % 32 == 0
	int Rw, rwPitch;          // window rows (<= 1264), pitch of a partial row (multiples of 16)
	int Tpad, groups;         // theta bins padded to groups * 64
	size_t tileCap;           // edge-list entries per tile (TW * TH)
};
constexpr int kShtMaxWindow = 1264;   // window rows of a vote workgroup: 1264 * 128 B = 158 KB of LDS
hipError_t launch_sht_compact_tiles(const ShtArgs& a, const ShtTileArgs& v, int frames, hipStream_t stream);
hipError_t launch_sht_vote_tiles(const ShtArgs& a, const ShtTileArgs& v, int frames, hipStream_t stream);
hipError_t launch_sht_reduce_tiles(const ShtArgs& a, const ShtTileArgs& v, int frames, hipStream_t stream);
hipError_t launch_bytes_to_bits(const uint8_t* edges, int W, int H, int S, size_t frameStride, uint32_t* ebits, int wb, size_t bitsFrameStride,
                                int frames, hipStream_t stream);
hipError_t launch_sht_lines(const ShtArgs& a, int frames, hipStream_t stream);
hipError_t launch_sht_decode(const uint32_t* keys, const uint32_t* vals, const int* counts, size_t lineCap, int frames, int T, int barrier, float thetaStep,
                             int maxLines, int strengthBits, void* lines /*compvhip_line*/, size_t outCap, hipStream_t stream);
int sht_nms_groups(int T);
size_t sht_nms_rows(int R);
int sht_lines_blocks(int R);
// acc [T][pitch] -> reference layout [R][stride]
// maxLines > 0: only the first min(count, lineCap, maxLines) lines of a frame are converted (what sht_decode_kernel wrote)
hipError_t launch_sht_cartesian(const void* lines /*compvhip_line*/, const int* counts, size_t lineCap, int maxLines, int frames, int T, const float* cosT,
                                const float* invSinT, float widthF, float r, float* out /*[frames][lineCap][4]*/, hipStream_t stream);
hipError_t launch_sht_acc_transpose(const uint16_t* accT, int R, int T, int accPitch, int32_t* out, size_t outStride, hipStream_t stream);
// the line sort sized on the device (sht_sort_kernels.hip): counting sort on the strength, stable ranks from chunks sorted in the LDS
constexpr int kShtSortChunk = 4096;             // lines per chunk (one workgroup)
constexpr int kShtSortMaxStrengthBits = 13;     // 8192 strength bins: max(W, H) <= 4095
constexpr int kShtSortMaxChunks = 32;           // chunks per frame it is used for (line capacities up to 131 072 per frame)
struct ShtSortArgs {
	uint32_t* sortedKeys;     // [frames * lineCap] per line of a sorted chunk: (inverted strength << 12) | rank inside its run of equal strengths
	uint32_t* sortedVals;     // [frames * lineCap] its accumulator cell
	uint16_t* chunkHist;      // [frames][chunks][8192] lines per inverted strength of a chunk (written whole by the chunks that have lines)
	uint32_t* strengthStart;  // [frames][8192] first slot of an inverted strength in the frame's sorted line list
	int chunks;               // chunks per frame = ceil(lineCap / 4096)
};
hipError_t launch_sht_sort_lines(const ShtArgs& a, const ShtSortArgs& q, int frames, float thetaStep, int maxLines, void* lines /*compvhip_line*/, size_t outCap,
                                 hipStream_t stream);
// one stable descending radix sort over the first n (key, value) slots; temp == nullptr queries tempBytes (for n = the capacity)
hipError_t sht_sort_pairs(void* temp, size_t& tempBytes, const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valsIn, uint32_t* valsOut, size_t n,
                          int keyBits, hipStream_t stream);

} // namespace compvhip
