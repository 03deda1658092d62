// api.cpp -- host side of the C ABI declared in include/compv_hip.h (compiled with hipcc, no torch types).
//
// Host-pointer entry points mirror the reference's process() contract (synchronous, results in host memory):
//   compvhip_canny_u8     <- CompVEdgeDeteCanny::process   core/features/edges/compv_core_feature_canny_dete.cxx:123-331
//   compvhip_edge_dete_u8 <- CompVCornerDeteEdgeBase::process core/features/edges/compv_core_feature_edge_dete.cxx:55-206
//   compvhip_houghsht_u8  <- CompVHoughSht::process        core/features/hough/compv_core_feature_houghsht.cxx:96-262
// Device-pointer ("plan") entry points run the same kernels on batches of frames resident in HBM.
#include "../../include/compv_hip.h"
#include "kernels.hpp"
#include "kht.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include <hip/hip_runtime.h>
#if defined(__linux__)
#include <sched.h>
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <memory>
#include <vector>

using namespace compvhip;

namespace {
constexpr int kMaxRounds = 4096;       // hysteresis round flag slots (a multiple of 4); a frame that needs more rounds reuses them (enqueueResolve)
constexpr int kSpecRounds = 3;         // rounds enqueued speculatively between two convergence checks
constexpr size_t kMinLineCap = 1u << 16;  // per-frame line-key slots: max(caller's lineCap, 65536), clamped to R*T (include/compv_hip.h, compvhip_plan_houghsht)
constexpr int kAsyncDepth = 4;            // outstanding compvhip_plan_pipeline_async steps per plan
constexpr size_t kMaxTimeline = 4096;     // timing entries kept while nobody reads them (asynchronous steps)
} // namespace

// Device + host scratch of ONE KHT frame in flight: the context owns one for its host entry point, a plan one per worker thread of
// compvhip_plan_houghkht (every worker has its own HIP stream; nothing in here is shared between threads).
struct KhtScratch {
	hipStream_t stream = nullptr; bool ownStream = false;
	int32_t* counts = nullptr; size_t countsElems = 0;
	KhtVoteParams* params = nullptr; size_t paramsCap = 0;
	KhtCell* cells = nullptr; size_t cellsCap = 0;
	int* cellCount = nullptr;
	KhtPoint* pts = nullptr; size_t ptsCap = 0;
	KhtSpan* spans = nullptr; size_t spansCap = 0;
	KhtKernel* kernelsDev = nullptr;
	KhtStringDesc* strings = nullptr; size_t stringsCap = 0;
	uint32_t* counts32 = nullptr;
	KhtSpan* scratch = nullptr;
	KhtSubdivFrame* stack = nullptr;
	KhtBitPlane plane;                                         // the linker's working copy (zero border, destroyed by the walk)
	KhtPoint* linked = nullptr; size_t linkedCap = 0;         // points of the strings, string after string: PINNED host memory, written by the linker, uploaded without staging
	KhtPeaksWork peaks;                                        // sort records, visited map, axes of the peak stage
	std::vector<KhtCell> cellsHost;                            // the vote cells of the frame, downloaded
	double stageMs[6] = {};   // link, subdivide (GPU), statistics (GPU), prune + Gmin, vote + peaks (GPU), sort + sweep of the last call
	std::string err;
};

struct KhtBatchFrame {      // host state of one frame of the batch; persists from call to call (vectors keep their capacity)
	KhtBitPlane plane; size_t most = 0, ptsOff = 0, nPts = 0;
	std::vector<KhtRange> strings; size_t slotBase = 0, slots = 0;
	uint32_t nClusters = 0;
	std::vector<KhtKernel> kernels; double hmax = 0.0, GS = 1.0; bool haveGS = false;
	std::vector<KhtVoteParams> params; size_t paramsBase = 0;
	std::vector<KhtCell> cells; size_t cellOff = 0; int cellCount = 0;
	KhtPeaksWork peaks; std::vector<KhtLine> out;
	double ms[6] = {};
	int code = COMPVHIP_OK; std::string err;
};
struct KhtBatchState {
	hipStream_t stream = nullptr;
	uint32_t* dBits = nullptr; uint32_t* hostBits = nullptr; size_t bitsWords = 0;      // [frames of a group][wpr * H]: device / pinned
	std::vector<hipEvent_t> ready;                                                     // frame f's bit plane has arrived
	KhtPoint* linked = nullptr; size_t linkedCap = 0; KhtPoint* pts = nullptr; size_t ptsCap = 0;   // points of every frame's strings: pinned arena (the linkers write it) / device
	KhtStringDesc* strings = nullptr; uint32_t* counts32 = nullptr; size_t stringsCap = 0; KhtStringDesc* stringsHost = nullptr; size_t stringsHostCap = 0;
	uint32_t* totals = nullptr;                                                        // device [kKhtBatch + 1]: clusters per frame, truncation flag
	KhtSpan* spans = nullptr; KhtSpan* scratch = nullptr; KhtSubdivFrame* stack = nullptr; KhtKernel* kernelsDev = nullptr; size_t spansCap = 0;
	KhtKernel* kernelsHost = nullptr; size_t kernelsHostCap = 0;                       // pinned
	int32_t* counts = nullptr; size_t countsElems = 0;
	KhtVoteParams* params = nullptr; size_t paramsCap = 0; KhtVoteParams* paramsHost = nullptr; size_t paramsHostCap = 0;
	KhtCell* cells = nullptr; size_t cellsCap = 0; int* cellCount = nullptr; KhtCell* cellsHost = nullptr; size_t cellsHostCap = 0;
	std::vector<KhtBatchFrame> frames;
	hipEvent_t syncEv = nullptr;   // blocking-sync event: a controller that waits for a GPU stage SLEEPS (hipStreamSynchronize spins on a CPU of the quota the workers need)
	double stageMs[6] = {};   // of the groups this state handled in the current call
};

struct compvhip_ctx {
	int device = 0;
	std::string err;
	std::atomic<long> live{0};   // hipMalloc / hipFree balance; KHT workers of a plan allocate from their own threads
	hipStream_t stream = nullptr;      // stream of the host entry points
	compvhip_plan* hostPlan = nullptr; // single-frame plan cached for the host entry points
	uint8_t* dIn = nullptr;            // device staging of the host entry points
	uint8_t* dOut = nullptr;
	size_t dInBytes = 0, dOutBytes = 0;
	uint8_t* dPacked = nullptr; size_t dPackedBytes = 0; // packed-pixel staging of compvhip_grayscale_u8
	uint32_t* dHist = nullptr;                             // [256] histogram + 1 result word of compvhip_otsu_u8
	int32_t* dCounts = nullptr;
	int32_t* dAccOut = nullptr; size_t dAccOutElems = 0;
	KhtScratch kht;                    // KHT scratch of the host entry point (compvhip_houghkht_u8)
};

struct TimingEntry { const char* name; hipEvent_t a, b; };

// one step of the device-resident pipeline: [grayscale ->] Canny -> SHT [-> toCartesian] (compvhip_plan_pipeline{,_async,_ex})
struct StepParams {
	const uint8_t* d_in = nullptr; float tLow = 0.f, tHigh = 0.f; int threshold = 0, maxLines = 0;
	int ksize = 3, thresholdType = COMPVHIP_CANNY_THRESHOLD_COMPARE_TO_GRADIENT, pixfmt = COMPVHIP_FMT_Y;
	uint8_t* d_gray = nullptr; int32_t* d_otsu = nullptr; float* d_cart = nullptr;
	uint8_t* d_edges = nullptr; compvhip_line* d_lines = nullptr; size_t lineCap = 0; int32_t* d_counts = nullptr;
};

struct compvhip_plan {
	compvhip_ctx* ctx = nullptr;
	size_t W = 0, H = 0, S = 0, frames = 0;
	float thetaDeg = 1.f;
	// canny
	int tilesX = 0, tilesY = 0, wb = 0;
	size_t bitsFrameStride = 0;
	uint32_t* ebits = nullptr; uint32_t* ubits = nullptr;
	int* counters = nullptr;  // ONE device allocation zeroed by ONE memset per step: [edgeCounts frames][lineCounts frames][tileCounts frames*tiles][blockCounts frames*lineBlocks][frameTotals frames*kFrameSlot][lineTotal kFrameSlot][flags kMaxRounds]
	size_t nCounts = 0;       // ints in front of the flags
	int* flags = nullptr; int* hFlags = nullptr; // device (inside counters) / pinned host (kAsyncDepth + 1 slots)
	int* frameTotals = nullptr; unsigned int* lineTotal = nullptr; // device (inside counters): NMS survivors per frame (one per 128-byte line) / key slots in use
	unsigned int* hTotals = nullptr;             // pinned host (behind hFlags): lineTotal of the synchronous call (slot 0) and of the asynchronous steps (1 + ticket)
	// The line sort covers the key slots that exist.  A synchronous step reads their number before it enqueues the sort; an asynchronous step cannot, so it
	// sorts a range predicted from the totals of the plan's last steps (0 = none seen yet: the whole capacity) -- compvhip_plan_wait compares with the step's
	// real total and replays the step when the prediction was too small.
	unsigned int recentTotals[8] = {}; int recentN = 0;
	// speculative hysteresis rounds of a step: what the plan's last 8 asynchronous steps needed (the first round that changed nothing, inclusive), at least 2, at
	// most kSpecRounds; a step that needs more is replayed by compvhip_plan_wait and teaches the plan
	int specRounds = kSpecRounds; unsigned char recentRounds[8] = {}; int recentRoundsN = 0;
	int* hRoundsDev = nullptr; int* stepHostSlot = nullptr;   // hRounds as the device sees it / the slot of the asynchronous step being enqueued (nullptr otherwise)
	int* hRounds = nullptr;                      // pinned host, per ticket: [0] the step's line total (the last int of its counter slot ... see runStepAsync), [kFrameSlot .. +3] its first 4 round flags
	int roundsUsed = 0;
	int maxRounds = kMaxRounds; // flag slots in use (COMPVHIP_RESOLVE_WRAP lowers it: tests of the slot reuse)
	bool countersFresh = false; // the step's memset already zeroed the edge/line counts (no second fill in front of the SHT stage)
	int2* thrDev = nullptr; unsigned int* sums = nullptr;
	uint8_t* dirty = nullptr;  // per-workgroup change flags of the resolve rounds
	uint8_t* patchOut = nullptr; uint8_t* copyBack = nullptr; // byte map the tile kernel writes and the resolve rounds patch / in-place target of the last Canny call
	uint8_t* grayTmp = nullptr; // luma plane of a packed-input step when the caller does not want it (compvhip_plan_pipeline_ex)
	uint8_t* tmpOut = nullptr; // aliasing (in == out) scratch: a tile may still read the row halo a neighbour has overwritten
	bool bitsValid = false;
	// sht
	bool shtReady = false;
	size_t R = 0, T = 0; float thetaStep = 0.f; int accPitch = 0;
	uint8_t* blurTmp = nullptr;                       // u8 intermediate of the fixed-point convolution
	uint32_t* hist = nullptr; int32_t* otsu = nullptr; // pre-processing scratch: partial histograms, [frames] Otsu level
	float* cosT = nullptr; float* invSinT = nullptr; // toCartesian tables: cosf(theta_col), 1/sinf(theta_col)
	int32_t* sinQ = nullptr; int32_t* cosQ = nullptr;
	uint32_t* edges = nullptr; size_t edgeCap = 0; int* edgeCounts = nullptr;
	uint16_t* acc = nullptr; size_t accFrameStride = 0;
	uint32_t* keysA = nullptr; uint32_t* keysB = nullptr; uint32_t* valsA = nullptr; uint32_t* valsB = nullptr; size_t lineCap = 0; int* lineCounts = nullptr;
	int2* reach = nullptr;                       // [T] accumulator rows the windows of a theta cover
	int2* nmsRange = nullptr;                    // [column groups of the NMS] accumulator rows the windows can reach
	uint8_t* nmsFlags = nullptr;                 // NMS survivors (flag planes)
	int* blockCounts = nullptr; int lineBlocks = 0;   // NMS survivors per 64 accumulator rows (part of `counters`)
	void* sortTemp = nullptr; size_t sortTempBytes = 0;
	int strengthBits = 16, keyBits = 0;
	// the line sort sized on the device (sht_sort_kernels.hip): used when a strength has at most 13 bits and a frame at most 32 chunks of keys
	uint16_t* chunkHist = nullptr; uint32_t* strengthStart = nullptr; int sortChunks = 0; bool deviceSort = false;
	// voting over image tiles (planned at plan creation: the per-tile edge counters live in `counters`)
	bool voteTiles = false;                      // the tile grid exists
	ShtTileArgs vt = {};                         // geometry + device tables
	std::vector<int32_t> vtKt, vtRowBase;        // host copies of the [tiles][T] tables
	int32_t* dKt = nullptr; int32_t* dRowBase = nullptr; uint8_t* partLo = nullptr; uint8_t* partHi = nullptr; uint8_t* colFlag = nullptr; int* tileCounts = nullptr;
	// batched KHT (compvhip_plan_houghkht): one scratch set + stream per worker thread, stage clocks of the last call
	std::vector<KhtBatchState*> khtBatch;        // device / pinned buffers and per-frame host state of the batched call: one per group of frames in flight
	std::vector<std::unique_ptr<KhtPeaksWork>> khtWork;   // sort + sweep workspace (axes, 1.6 MB visited map at 4K) of WORKER w: it stays in that core's cache from frame to frame
	double khtStageMs[6] = {}; double khtWallMs = 0.0; int khtThreads = 0;
	// asynchronous steps (compvhip_plan_pipeline_async / compvhip_plan_wait)
	// seq: enqueue order; replay: an EARLIER step of the plan was replayed after this one ran -- its outputs may have been overwritten
	struct AsyncStep { bool used = false; bool replay = false; uint64_t seq = 0; hipEvent_t done = nullptr; hipStream_t stream = nullptr; StepParams sp; size_t sortN = 0; int rounds = 0; } steps[kAsyncDepth];
	uint64_t stepSeq = 0;
	// timing
	int timing = 0; // 0 off, 1 every kernel, 2 canny_tile + sht_vote, 3 sht_vote only, 4 canny_tile only
	std::vector<hipEvent_t> eventPool;
	std::vector<TimingEntry> timeline;
	std::vector<std::string> timingNames; std::vector<float> timingMs;
};

namespace {

int fail(compvhip_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess)
{
	if (ctx) {
		ctx->err = what ? what : "";
		if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
	}
	return code;
}

#define HIPCHK(ctx, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail((ctx), COMPVHIP_E_HIP, #call, e__); } while (0)

template <typename T>
hipError_t dmalloc(compvhip_ctx* ctx, T** p, size_t count)
{
	*p = nullptr;
	if (!count) return hipSuccess;
	hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
	if (e == hipSuccess && ctx) ctx->live++;
	return e;
}
template <typename T>
void dfree(compvhip_ctx* ctx, T*& p)
{
	if (p) { (void)hipFree(p); if (ctx) ctx->live--; p = nullptr; }
}

size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

void khtScratchFree(compvhip_ctx* ctx, KhtScratch& k)
{
	dfree(ctx, k.counts); dfree(ctx, k.params); dfree(ctx, k.cells); dfree(ctx, k.cellCount);
	dfree(ctx, k.pts); dfree(ctx, k.spans); dfree(ctx, k.kernelsDev);
	dfree(ctx, k.strings); dfree(ctx, k.counts32); dfree(ctx, k.scratch); dfree(ctx, k.stack);
	if (k.linked) { (void)hipHostFree(k.linked); k.linked = nullptr; k.linkedCap = 0; }
	if (k.ownStream && k.stream) { (void)hipStreamDestroy(k.stream); k.stream = nullptr; }
}

void khtBatchFree(compvhip_ctx* ctx, KhtBatchState* b)
{
	if (!b) return;
	dfree(ctx, b->dBits); dfree(ctx, b->pts); dfree(ctx, b->strings); dfree(ctx, b->counts32); dfree(ctx, b->totals);
	dfree(ctx, b->spans); dfree(ctx, b->scratch); dfree(ctx, b->stack); dfree(ctx, b->kernelsDev);
	dfree(ctx, b->counts); dfree(ctx, b->params); dfree(ctx, b->cells); dfree(ctx, b->cellCount);
	if (b->hostBits) (void)hipHostFree(b->hostBits);
	if (b->linked) (void)hipHostFree(b->linked);
	if (b->stringsHost) (void)hipHostFree(b->stringsHost);
	if (b->kernelsHost) (void)hipHostFree(b->kernelsHost);
	if (b->paramsHost) (void)hipHostFree(b->paramsHost);
	if (b->cellsHost) (void)hipHostFree(b->cellsHost);
	for (hipEvent_t e : b->ready) (void)hipEventDestroy(e);
	if (b->syncEv) (void)hipEventDestroy(b->syncEv);
	if (b->stream) (void)hipStreamDestroy(b->stream);
	delete b;
}


// ---- thresholds: core/features/edges/compv_core_feature_canny_dete.cxx:251-266 (COMPARE_TO_GRADIENT branch) ----
// cosf / sinf exactly as the reference's scalar calls resolve them (never merged into sincosf)
__attribute__((noinline, optnone)) float libmCosf(float x) { return cosf(x); }
__attribute__((noinline, optnone)) float libmSinf(float x) { return sinf(x); }

void gradientThresholds(float fLow, float fHigh, int* tLow, int* tHigh)
{
	const float l = fLow < 1.f ? 1.f : (fLow > 65535.f ? 65535.f : fLow);
	const float h = fHigh < 1.f ? 1.f : (fHigh > 65535.f ? 65535.f : fHigh);
	uint16_t lo = static_cast<uint16_t>(l), hi = static_cast<uint16_t>(h);
	lo = static_cast<uint16_t>(std::max<int>(1, lo));
	hi = static_cast<uint16_t>(std::max<int>(lo + 2, hi));
	*tLow = lo; *tHigh = hi;
}

// ---- column coverage of the reference's SIMD + scalar-remainder dispatch (quirk Q3) ----
// ...canny_dete.cxx:362-412 (nms_gather) and :483-527 (hysteresis): the row leaves cover col = 1, 1+mpw, ... while
// col < (W-1)-(mpw-1); the scalar remainder restarts at (W-1) & -(mpw-1).
void cannyCoverage(size_t W, int* simdEnd, int* cStart)
{
	const size_t maxCols = W - 1;
	size_t mpw = 1;
	if (maxCols >= 16) mpw = 16; else if (maxCols >= 8) mpw = 8;
	if (mpw == 1) { *simdEnd = 1; *cStart = 1; return; }
	size_t col = 1;
	while (col + (mpw - 1) < maxCols) col += mpw;
	*simdEnd = static_cast<int>(col);
	*cStart = static_cast<int>(maxCols & static_cast<size_t>(-static_cast<ptrdiff_t>(mpw - 1)));
}

// ---- SHT geometry/tables: core/features/hough/compv_core_feature_houghsht.cxx:42-52,318-348 ----
const float kPiF = 3.1415926535897932384626433f;   // kfMathTrigPi (base/math/compv_math.cxx:27)
float piOver180() { return kPiF / 180.f; }         // kfMathTrigPiOver180 (:30)

int shtDims(size_t W, size_t H, float thetaDeg, size_t* R, size_t* T, float* step)
{
	if (!W || !H || !(thetaDeg > 0.f)) return COMPVHIP_E_INVALID_PARAMETER;
	const float fTheta = thetaDeg * piOver180();
	const float fRho = 1.f;
	*R = static_cast<size_t>((static_cast<float>(((W + H) << 1) + 1) / fRho) + 0.5);
	*T = static_cast<size_t>((kPiF / fTheta) + 0.5);
	if (step) *step = fTheta;
	return COMPVHIP_OK;
}

void shtTables(float thetaDeg, size_t T, std::vector<int32_t>& sinQ, std::vector<int32_t>& cosQ)
{
	// float32 running angle and libm sinf/cosf on the HOST, exactly as initCoords (:335-339); never device sinf.
	const float fTheta = thetaDeg * piOver180();
	const float fRho = 1.f;
	sinQ.resize(T); cosQ.resize(T);
	float tt = 0.f;
	for (size_t t = 0; t < T; ++t, tt += fTheta) {
		sinQ[t] = static_cast<int32_t>((libmSinf(tt) * fRho) * 65535.f);
		cosQ[t] = static_cast<int32_t>((libmCosf(tt) * fRho) * 65535.f);
	}
}

// ---- tiles of the second-generation vote kernel (sht_tiles_kernels.hip) ----
// Grid of equal tiles (width a multiple of 32 px) whose rho windows -- per theta, the span of (lx cosQ + ly sinQ + Clo) >> 16 over the tile -- have at
// most kShtMaxWindow rows; fills the [tiles][T] tables K and rowBase (int64 arithmetic here, int32 on the device).  Grids are tried in order of tile
// count (split the dimension with the longer tile side) and the smallest one that fits is taken -- except when it leaves the chip nearly empty (fewer
// than 64 workgroups = frames x tiles x theta groups: the single-frame plans of the host entry points): then the next four grids are priced with a model
// calibrated on the 4K benchmark (a workgroup: 67 us per 960 x 720 pixels of tile + 9.3 us per 1208 window rows; the reduce kernel: 36 us per
// 32 x 12 x 1208 partial-window rows) and the cheapest is taken: 7 x 4 instead of 4 x 3 tiles for one 4K frame, 0.349 -> 0.320 ms per compvhip_houghsht_u8
// call.  Measured and NOT done: the same model for batches (round 4: 32 x 1080p on 4 x 2 instead of 2 x 2 tiles = three full rounds of the 256 CUs instead
// of one and a half: no change, that step is bound by the launch chain; 32 x 720p on 4 x 2 instead of 2 x 1: 0.180 against 0.170 ms per step).
// Lab knob (like COMPVHIP_RESOLVE_WRAP): COMPVHIP_VOTE_MAX_WINDOW=<rows> caps a voting workgroup's window below the 1264 rows the LDS holds, i.e. forces a
// finer tile grid that leaves LDS free beside a voting workgroup (co-residency experiments, profiles/r06/coresidency.md).  Results are identical for any
// grid (tests/test_gpu_parity.py::test_vote_window_knob_is_bit_exact); unset = the product's choice.
static size_t voteMaxWindow()   // read whenever a plan is made: one process can hold plans on different grids (tools/coresidency/grid_ab.py)
{
	const char* e = getenv("COMPVHIP_VOTE_MAX_WINDOW");
	const long n = e ? atol(e) : 0;
	return (n >= 64 && n <= kShtMaxWindow) ? static_cast<size_t>(n) : static_cast<size_t>(kShtMaxWindow);
}

static bool voteGridTables(size_t W, size_t H, const std::vector<int32_t>& sinQ, const std::vector<int32_t>& cosQ, int split, int& nx, int& ny, int& TW, int& TH, long long& worst,
                           std::vector<int32_t>* kt, std::vector<int32_t>* rowBase)
{
	const size_t T = sinQ.size();
	const long long barrier = static_cast<long long>(W + H);
	nx = 1; ny = 1;
	for (int k = 0; k < split; ++k) {
		const double tw = static_cast<double>(W) / nx, th = static_cast<double>(H) / ny;
		if (tw >= th) ++nx; else ++ny;
	}
	TW = static_cast<int>(alignUp((W + nx - 1) / nx, 32)); TH = static_cast<int>((H + ny - 1) / ny);
	nx = static_cast<int>((W + TW - 1) / TW); // the rounding to 32 columns may save a column of tiles
	const int tiles = nx * ny;
	if (kt) { kt->assign(static_cast<size_t>(tiles) * T, 0); rowBase->assign(static_cast<size_t>(tiles) * T, 0); }
	worst = 0;
	for (int ty = 0; ty < ny; ++ty) for (int tx = 0; tx < nx; ++tx) {
		const long long x0 = static_cast<long long>(tx) * TW, y0 = static_cast<long long>(ty) * TH;
		const long long mx = TW - 1, my = TH - 1; // largest local coordinates (tiles at the image border are not clipped: simpler, still exact)
		for (size_t t = 0; t < T; ++t) {
			const long long c = cosQ[t], sn = sinQ[t];
			const long long C = x0 * c + y0 * sn;
			long long Chi = C / 65536; if (C - Chi * 65536 < 0) --Chi; // floor
			const long long Clo = C - Chi * 65536;
			const long long corners[4] = { Clo, mx * c + Clo, my * sn + Clo, mx * c + my * sn + Clo };
			long long qmin = 0, qmax = 0;
			for (int k = 0; k < 4; ++k) {
				long long q = corners[k] / 65536; if (corners[k] - q * 65536 < 0) --q;
				if (k == 0 || q < qmin) qmin = q;
				if (k == 0 || q > qmax) qmax = q;
			}
			// the window starts on a multiple of 16 accumulator rows (d extra rows at its top): the reduce kernel adds whole 16-byte groups of count bytes
			const long long base = barrier - Chi - qmax;
			const long long d = ((base % 16) + 16) % 16;
			worst = std::max(worst, qmax - qmin + 1 + d);
			if (kt) {
				(*kt)[(static_cast<size_t>(ty) * nx + tx) * T + t] = static_cast<int32_t>((qmax + d) * 65536 + 65535 - Clo);
				(*rowBase)[(static_cast<size_t>(ty) * nx + tx) * T + t] = static_cast<int32_t>(base - d);
			}
		}
	}
	return alignUp(static_cast<size_t>(worst), 16) <= voteMaxWindow() && TW <= 1280 &&
	       static_cast<long long>(TW - 1) * 65535 + static_cast<long long>(TH - 1) * 65535 < 0x7f000000LL;
}

bool planVoteTiles(size_t W, size_t H, size_t frames, const std::vector<int32_t>& sinQ, const std::vector<int32_t>& cosQ, ShtTileArgs& v, std::vector<int32_t>& kt,
                   std::vector<int32_t>& rowBase)
{
	const size_t T = sinQ.size();
	const int groups = static_cast<int>((T + 63) / 64);
	int first = -1;
	for (int split = 0; split < 512 && first < 0; ++split) {
		int nx, ny, TW, TH; long long worst;
		if (voteGridTables(W, H, sinQ, cosQ, split, nx, ny, TW, TH, worst, nullptr, nullptr)) first = split;
	}
	if (first < 0) return false;
	int best = first; double bestCost = 0.0;
	int nx0, ny0, TW0, TH0; long long worst0;
	(void)voteGridTables(W, H, sinQ, cosQ, first, nx0, ny0, TW0, TH0, worst0, nullptr, nullptr);
	const bool starved = static_cast<double>(frames) * nx0 * ny0 * groups < 64.0;   // a quarter of the CUs at most: single frames (the host entry points)
	for (int split = first; starved && split <= first + 4; ++split) {
		int nx, ny, TW, TH; long long worst;
		if (!voteGridTables(W, H, sinQ, cosQ, split, nx, ny, TW, TH, worst, nullptr, nullptr)) continue;
		const double rw = static_cast<double>(alignUp(static_cast<size_t>(worst), 16));
		const double wgs = static_cast<double>(frames) * nx * ny * groups;
		const double rounds = wgs <= 1024.0 ? std::ceil(wgs / 256.0) : wgs / 256.0;   // a few rounds are quantised, many level out (the dispatcher hands workgroups out as CUs free up)
		const double perWg = 67.0 * (static_cast<double>(TW) * TH) / (960.0 * 720.0) + 9.3 * rw / 1208.0;
		const double reduce = 36.0 * (static_cast<double>(frames) * nx * ny * rw) / (32.0 * 12.0 * 1208.0);
		const double cost = rounds * perWg + reduce;
		if (split == first || cost < bestCost * 0.97) { best = split; bestCost = cost; }   // a finer grid has to win by 3 % (more partial windows, more memory)
	}
	int nx, ny, TW, TH; long long worst;
	if (!voteGridTables(W, H, sinQ, cosQ, best, nx, ny, TW, TH, worst, &kt, &rowBase)) return false;
	v.nx = nx; v.ny = ny; v.TW = TW; v.TH = TH; v.tiles = nx * ny;
	v.Rw = static_cast<int>(alignUp(static_cast<size_t>(worst), 16)); v.rwPitch = v.Rw;
	v.groups = groups; v.Tpad = v.groups * 64;
	v.tileCap = static_cast<size_t>(TW) * TH;
	return true;
}

// timing mode 1 = every kernel; 2 = only the two kernels bench.py prices against the roofline (an event pair costs a few
// microseconds of stream time, ~0.1 ms per step when wrapped around all ~13 launches of the pipeline)
static bool stampWanted(const compvhip_plan* p, const char* name)
{
	if (p->timing == 1) return true;
	if (p->timing == 2) return !strcmp(name, "canny_tile_kernel") || !strcmp(name, "sht_vote_kernel");
	if (p->timing == 3) return !strcmp(name, "sht_vote_kernel");
	if (p->timing == 4) return !strcmp(name, "canny_tile_kernel");
	return false;
}

static bool takeEvent(compvhip_plan* p, hipEvent_t* e)
{
	if (!p->eventPool.empty()) { *e = p->eventPool.back(); p->eventPool.pop_back(); return true; }
	return hipEventCreate(e) == hipSuccess;
}

struct Stamp {
	compvhip_plan* p; hipStream_t s; size_t idx; bool on;
	Stamp(compvhip_plan* plan, hipStream_t stream, const char* name) : p(plan), s(stream), idx(0), on(stampWanted(plan, name))
	{
		if (!on) return;
		TimingEntry t; t.name = name;
		if (!takeEvent(p, &t.a)) { on = false; return; }
		if (!takeEvent(p, &t.b)) { p->eventPool.push_back(t.a); on = false; return; }
		(void)hipEventRecord(t.a, s);
		p->timeline.push_back(t);
		idx = p->timeline.size() - 1;
	}
	~Stamp() { if (on) (void)hipEventRecord(p->timeline[idx].b, s); }
};

void timelineClear(compvhip_plan* p)
{
	for (auto& t : p->timeline) { p->eventPool.push_back(t.a); p->eventPool.push_back(t.b); } // events are reused, not re-created
	p->timeline.clear();
}

void timelineCollect(compvhip_plan* p)
{
	p->timingNames.clear(); p->timingMs.clear();
	for (auto& t : p->timeline) {
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) ms = -1.f;
		p->timingNames.push_back(t.name); p->timingMs.push_back(ms);
	}
	timelineClear(p);
}

int ensureSht(compvhip_plan* p)
{
	if (p->shtReady) return COMPVHIP_OK;
	compvhip_ctx* ctx = p->ctx;
	// a previous attempt may have failed half way (out of memory): start from a clean slate instead of leaking its buffers
	dfree(ctx, p->sinQ); dfree(ctx, p->cosQ); dfree(ctx, p->cosT); dfree(ctx, p->invSinT);
	dfree(ctx, p->edges); dfree(ctx, p->acc);
	size_t R, T; float step;
	int rc = shtDims(p->W, p->H, p->thetaDeg, &R, &T, &step);
	if (rc) return fail(ctx, rc, "invalid SHT geometry");
	if (T < 5) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "theta step too large (fewer than 5 theta bins)");
	if (!p->voteTiles) return fail(ctx, COMPVHIP_E_NOT_IMPLEMENTED, "no tile grid fits the vote windows of this geometry");
	p->R = R; p->T = T; p->thetaStep = step;
	p->accPitch = static_cast<int>(alignUp(R, 64));
	p->accFrameStride = static_cast<size_t>(p->accPitch) * T;
	std::vector<int32_t> s, c;
	shtTables(p->thetaDeg, T, s, c);
	HIPCHK(ctx, dmalloc(ctx, &p->sinQ, T));
	HIPCHK(ctx, dmalloc(ctx, &p->cosQ, T));
	HIPCHK(ctx, hipMemcpy(p->sinQ, s.data(), T * sizeof(int32_t), hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(p->cosQ, c.data(), T * sizeof(int32_t), hipMemcpyHostToDevice));
	{
		// toCartesian (houghsht.cxx:582): a = std::cos(theta), b = 1.f / std::sin(theta) with theta = col * thetaStep in f32
		// (separate libm calls: a compiler that fuses them into sincosf() changes cos by an ulp for some arguments)
		std::vector<float> ct(T), ist(T);
		for (size_t t = 0; t < T; ++t) {
			const float theta = static_cast<float>(t) * step;
			ct[t] = libmCosf(theta);
			ist[t] = 1.f / libmSinf(theta);
		}
		HIPCHK(ctx, dmalloc(ctx, &p->cosT, T));
		HIPCHK(ctx, dmalloc(ctx, &p->invSinT, T));
		HIPCHK(ctx, hipMemcpy(p->cosT, ct.data(), T * sizeof(float), hipMemcpyHostToDevice));
		HIPCHK(ctx, hipMemcpy(p->invSinT, ist.data(), T * sizeof(float), hipMemcpyHostToDevice));
	}
	{
		if (p->vt.tiles <= 0 || p->vtKt.size() != static_cast<size_t>(p->vt.tiles) * T) return fail(ctx, COMPVHIP_E_INVALID_STATE, "vote tiles were not planned");
		dfree(ctx, p->dKt); dfree(ctx, p->dRowBase); dfree(ctx, p->partLo); dfree(ctx, p->partHi); dfree(ctx, p->colFlag);
		{
			// accumulator rows the NMS has to look at, per group of 8 theta columns (+ the column either side): the union of the tiles' windows
			const int tiles = p->vt.tiles, Rw = p->vt.Rw, groups = sht_nms_groups(static_cast<int>(T));
			std::vector<int2> range(static_cast<size_t>(groups));
			for (int g = 0; g < groups; ++g) {
				int lo = INT32_MAX, hi = INT32_MIN;
				for (int c = 8 * g - 1; c <= 8 * g + 8; ++c) {
					if (c < 0 || c >= static_cast<int>(T)) continue;
					for (int i = 0; i < tiles; ++i) {
						const int b = p->vtRowBase[static_cast<size_t>(i) * T + c];
						lo = std::min(lo, b); hi = std::max(hi, b + Rw);
					}
				}
				range[g] = make_int2(lo - 1, hi + 1);
			}
			std::vector<int2> reach(T);
			for (size_t t = 0; t < T; ++t) {
				int lo = INT32_MAX, hi = INT32_MIN;
				for (int i = 0; i < tiles; ++i) {
					const int b = p->vtRowBase[static_cast<size_t>(i) * T + t];
					lo = std::min(lo, b); hi = std::max(hi, b + Rw);
				}
				reach[t] = make_int2(lo, hi);
			}
			dfree(ctx, p->reach);
			HIPCHK(ctx, dmalloc(ctx, &p->reach, reach.size()));
			HIPCHK(ctx, hipMemcpy(p->reach, reach.data(), reach.size() * sizeof(int2), hipMemcpyHostToDevice));
			dfree(ctx, p->nmsRange);
			HIPCHK(ctx, dmalloc(ctx, &p->nmsRange, range.size()));
			HIPCHK(ctx, hipMemcpy(p->nmsRange, range.data(), range.size() * sizeof(int2), hipMemcpyHostToDevice));
		}
		HIPCHK(ctx, dmalloc(ctx, &p->dKt, p->vtKt.size()));
		HIPCHK(ctx, dmalloc(ctx, &p->dRowBase, p->vtRowBase.size()));
		HIPCHK(ctx, hipMemcpy(p->dKt, p->vtKt.data(), p->vtKt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
		HIPCHK(ctx, hipMemcpy(p->dRowBase, p->vtRowBase.data(), p->vtRowBase.size() * sizeof(int32_t), hipMemcpyHostToDevice));
		{
			// partial windows: one byte plane for the low bytes of the counts, one for the high bytes (only written / read for the few columns
			// that hold a count >= 256: a tile's share of a strong line), one flag byte per column
			const size_t cols = p->frames * p->vt.tiles * static_cast<size_t>(p->vt.Tpad);
			HIPCHK(ctx, dmalloc(ctx, &p->partLo, cols * p->vt.rwPitch));
			HIPCHK(ctx, dmalloc(ctx, &p->partHi, cols * p->vt.rwPitch));
			HIPCHK(ctx, dmalloc(ctx, &p->colFlag, cols));
		}
		p->vt.kt = p->dKt; p->vt.rowBase = p->dRowBase; p->vt.partLo = p->partLo; p->vt.partHi = p->partHi; p->vt.colFlag = p->colFlag; p->vt.reach = p->reach; p->vt.tileCounts = p->tileCounts;
		p->edgeCap = static_cast<size_t>(p->vt.tiles) * p->vt.tileCap; // per frame: one list of TW * TH entries per tile
	}
	HIPCHK(ctx, dmalloc(ctx, &p->edges, p->edgeCap * p->frames));
	HIPCHK(ctx, dmalloc(ctx, &p->acc, p->accFrameStride * p->frames));
	HIPCHK(ctx, hipMemset(p->acc, 0, sizeof(uint16_t) * p->accFrameStride * p->frames)); // rows [Rp, accPitch) stay zero for ever
	// line key = frameTag | strength (strengthBits); the accumulator cell rides as the sort value (sht_nms / rank / emit kernels).  A cell of
	// column theta counts the pixels with (x*cosQ + y*sinQ) in one 65536-wide interval; max(|cosQ|,|sinQ|) >= 46340 so every x (or every y)
	// contributes at most 2 pixels: count <= 2*max(W,H).  Fewer key bits = fewer radix-sort passes.
	p->strengthBits = 1;
	while ((static_cast<size_t>(1) << p->strengthBits) <= 2 * (p->W > p->H ? p->W : p->H)) p->strengthBits++;
	if (p->strengthBits > 16) p->strengthBits = 16; // the accumulator itself is u16
	int frameBits = 0;
	while ((static_cast<size_t>(1) << frameBits) < p->frames) frameBits++;
	p->keyBits = frameBits + p->strengthBits;
	if (R * T >= (static_cast<size_t>(1) << 32)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "theta step too small: the accumulator has 2^32 cells or more"); // 32-bit cell values
	if (p->keyBits > 32) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "too many frames for a 32-bit line key");
	dfree(ctx, p->nmsFlags);
	{
		const size_t frows = sht_nms_rows(static_cast<int>(R)), fgroups = static_cast<size_t>(sht_nms_groups(static_cast<int>(T)));
		HIPCHK(ctx, dmalloc(ctx, &p->nmsFlags, frows * fgroups * p->frames));
		HIPCHK(ctx, hipMemset(p->nmsFlags, 0, frows * fgroups * p->frames)); // the NMS kernel skips the blocks no window reaches
		if (sht_lines_blocks(static_cast<int>(R)) != p->lineBlocks) return fail(ctx, COMPVHIP_E_INVALID_STATE, "row-block counters were sized for another accumulator");
	}
	p->shtReady = true;
	return COMPVHIP_OK;
}

int ensureLineCap(compvhip_plan* p, size_t cap)
{
	compvhip_ctx* ctx = p->ctx;
	cap = std::min(cap, p->R * p->T);
	if (cap <= p->lineCap) return COMPVHIP_OK;
	dfree(ctx, p->keysA); dfree(ctx, p->keysB); dfree(ctx, p->valsA); dfree(ctx, p->valsB); dfree(ctx, p->sortTemp); dfree(ctx, p->chunkHist); dfree(ctx, p->strengthStart);
	p->lineCap = 0; p->deviceSort = false;
	HIPCHK(ctx, dmalloc(ctx, &p->keysA, cap * p->frames));
	HIPCHK(ctx, dmalloc(ctx, &p->keysB, cap * p->frames));
	HIPCHK(ctx, dmalloc(ctx, &p->valsA, cap * p->frames));
	HIPCHK(ctx, dmalloc(ctx, &p->valsB, cap * p->frames));
	size_t tb = 0;
	hipError_t e = sht_sort_pairs(nullptr, tb, p->keysA, p->keysB, p->valsA, p->valsB, cap * p->frames, p->keyBits, nullptr);
	if (e != hipSuccess) return fail(ctx, COMPVHIP_E_HIP, "radix sort size query", e);
	p->sortTempBytes = tb;
	uint8_t* tmp = nullptr;
	HIPCHK(ctx, dmalloc(ctx, &tmp, std::max<size_t>(tb, 16)));
	p->sortTemp = tmp;
	p->sortChunks = static_cast<int>((cap + kShtSortChunk - 1) / kShtSortChunk);
	if (p->strengthBits <= kShtSortMaxStrengthBits && p->sortChunks <= kShtSortMaxChunks) {
		HIPCHK(ctx, dmalloc(ctx, &p->chunkHist, p->frames * static_cast<size_t>(p->sortChunks) << kShtSortMaxStrengthBits));
		HIPCHK(ctx, dmalloc(ctx, &p->strengthStart, p->frames << kShtSortMaxStrengthBits));
		p->deviceSort = true;
	}
	p->lineCap = cap;
	p->recentN = 0;   // totals clamped to another capacity
	return COMPVHIP_OK;
}

ShtArgs shtArgs(compvhip_plan* p, int threshold)
{
	ShtArgs a;
	a.ebits = p->ebits; a.edges = p->edges; a.edgeCounts = p->edgeCounts; a.acc = p->acc;
	a.sinQ = p->sinQ; a.cosQ = p->cosQ; a.lineKeys = p->keysA; a.lineVals = p->valsA; a.lineCounts = p->lineCounts;
	a.frameTotals = p->frameTotals; a.lineTotal = p->lineTotal; a.sortN = 0; a.outCounts = nullptr; a.stepFlags = nullptr; a.hostStep = nullptr;
	a.nmsRange = p->nmsRange; a.blockCounts = p->blockCounts; a.lineBlocks = p->lineBlocks; a.nmsFlags = p->nmsFlags; a.nmsRows = static_cast<int>(sht_nms_rows(static_cast<int>(p->R))); a.nmsGroups = sht_nms_groups(static_cast<int>(p->T));
	a.bitsFrameStride = p->bitsFrameStride; a.edgeCap = p->edgeCap; a.accFrameStride = p->accFrameStride; a.lineCap = p->lineCap;
	a.W = static_cast<int>(p->W); a.H = static_cast<int>(p->H); a.wb = p->wb;
	a.R = static_cast<int>(p->R); a.T = static_cast<int>(p->T); a.accPitch = p->accPitch; a.barrier = static_cast<int>(p->W + p->H);
	a.threshold = threshold;
	a.nmsLastCol = static_cast<int>((p->T - 1) & ~static_cast<size_t>(3)); // quirk Q2: NMS covers theta columns [1, (T-1)&~3]
	a.frames = static_cast<int>(p->frames);
	a.strengthBits = p->strengthBits;
	return a;
}

// Enqueue canny tiles + `rounds` speculative resolve rounds starting at p->roundsUsed.
int ensurePreproc(compvhip_plan* p)
{
	compvhip_ctx* ctx = p->ctx;
	if (!p->hist) HIPCHK(ctx, dmalloc(ctx, &p->hist, static_cast<size_t>(256) * kOtsuMaxChunks * p->frames));
	if (!p->otsu) HIPCHK(ctx, dmalloc(ctx, &p->otsu, p->frames));
	return COMPVHIP_OK;
}

// thrMode: COMPVHIP_CANNY_THRESHOLD_* (per-frame device thresholds for PERCENT_OF_MEAN and OTSU)
int enqueueCanny(compvhip_plan* p, const uint8_t* d_in, uint8_t* d_out, int tLow, int tHigh, int ksize, int thrMode, float fLow, float fHigh, hipStream_t st)
{
	compvhip_ctx* ctx = p->ctx;
	CannyArgs a;
	a.zero = nullptr; a.nZero = 0;
	a.in = d_in; a.out = d_out; a.ebits = p->ebits; a.ubits = p->ubits; a.thrDev = (thrMode != COMPVHIP_CANNY_THRESHOLD_COMPARE_TO_GRADIENT) ? p->thrDev : nullptr;
	a.inFrameStride = p->S * p->H; a.outFrameStride = p->S * p->H; a.bitsFrameStride = p->bitsFrameStride;
	a.W = static_cast<int>(p->W); a.H = static_cast<int>(p->H); a.S = static_cast<int>(p->S); a.So = static_cast<int>(p->S);
	a.wb = p->wb; a.tilesX = p->tilesX; a.tilesY = p->tilesY; a.tLow = tLow; a.tHigh = tHigh; a.ksize = ksize;
	cannyCoverage(p->W, &a.simdEnd, &a.cStart);
	// coverage [1,simdEnd) U [cStart,W-1) equals the whole interior unless the two pieces leave a hole (W = 1 mod 16 ...)
	const bool gap = !((a.simdEnd >= a.W - 1) || (a.cStart <= a.simdEnd));
	if (thrMode == COMPVHIP_CANNY_THRESHOLD_OTSU) {
		int rc = ensurePreproc(p);
		if (rc) return rc;
		Stamp s(p, st, "otsu_kernels");
		HIPCHK(ctx, launch_otsu(d_in, a.W, a.H, a.S, a.inFrameStride, static_cast<int>(p->frames), fLow, fHigh, p->hist, p->otsu, p->thrDev, st));
	}
	if (thrMode == COMPVHIP_CANNY_THRESHOLD_PERCENT_OF_MEAN) {
		Stamp s(p, st, "canny_mean_thresholds");
		HIPCHK(ctx, launch_mean_thresholds(d_in, a.W, a.H, a.S, a.inFrameStride, static_cast<int>(p->frames), fLow, fHigh, p->sums, p->thrDev, st));
	}
	// ONE fill per step -- edge counts, line counts and the hysteresis round flags live in one allocation --, done by the first workgroups of the tile kernel
	// (round 6: the hipMemsetAsync it replaces was a launch of its own on the lane's chain)
	a.zero = p->counters; a.nZero = static_cast<int>(p->nCounts + kMaxRounds);
	p->countersFresh = true;
	p->roundsUsed = 0;
	{
		Stamp s(p, st, "canny_tile_kernel");
		HIPCHK(ctx, launch_canny_tiles(a, static_cast<int>(p->frames), gap, st));
	}
	return COMPVHIP_OK;
}

// d_out: the byte map to patch with the promoted pixels
int enqueueResolve(compvhip_plan* p, uint8_t* d_out, int rounds, hipStream_t st)
{
	compvhip_ctx* ctx = p->ctx;
	ResolveArgs r;
	r.ebits = p->ebits; r.ubits = p->ubits; r.out = d_out; r.flags = p->flags; r.dirty = p->dirty;
	r.outFrameStride = p->S * p->H; r.bitsFrameStride = p->bitsFrameStride;
	r.H = static_cast<int>(p->H); r.So = static_cast<int>(p->S); r.wb = p->wb;
	for (int i = 0; i < rounds; ++i) {
		if (p->roundsUsed >= p->maxRounds) {
			// More border crossings than flag slots (a weak chain that zigzags across a band border needs one round per crossing): the
			// slots are reused.  The flood only ever adds pixels, so it terminates however many rounds that takes.  The sequence
			// continues as round 4 -- same dirty-flag generation as round maxRounds, a multiple of 4 -- behind a "changed" in slot 3.
			HIPCHK(ctx, hipMemsetAsync(p->flags + 3, 0, sizeof(int) * (p->maxRounds - 3), st));
			HIPCHK(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p->flags + 3), 1, 1, st));
			p->roundsUsed = 4;
		}
		r.round = p->roundsUsed++;
		Stamp s(p, st, "canny_resolve_kernel");
		HIPCHK(ctx, launch_canny_resolve(r, static_cast<int>(p->frames), st));
	}
	return COMPVHIP_OK;
}

// true when the last enqueued round changed nothing (fixed point reached). Synchronises the stream.
int resolveConverged(compvhip_plan* p, hipStream_t st, bool* done)
{
	compvhip_ctx* ctx = p->ctx;
	const int last = p->roundsUsed - 1;
	HIPCHK(ctx, hipMemcpyAsync(p->hFlags, p->flags + last, sizeof(int), hipMemcpyDeviceToHost, st));
	HIPCHK(ctx, hipStreamSynchronize(st));
	*done = (p->hFlags[0] == 0);
	return COMPVHIP_OK;
}

int validateCannyParams(compvhip_ctx* ctx, float tLow, float tHigh, int ksize, int type, int* lo, int* hi)
{
	if (ksize != 3 && ksize != 5) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "kernel size must be 3 or 5"); // canny_dete.cxx:101
	if (type != COMPVHIP_CANNY_THRESHOLD_COMPARE_TO_GRADIENT && type != COMPVHIP_CANNY_THRESHOLD_PERCENT_OF_MEAN && type != COMPVHIP_CANNY_THRESHOLD_OTSU)
		return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "invalid threshold type"); // :83
	if (type == COMPVHIP_CANNY_THRESHOLD_OTSU && !(tLow > 0.f)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "Otsu threshold factors must be > 0");
	if (tLow >= tHigh) return fail(ctx, COMPVHIP_E_INVALID_STATE, "tLow >= tHigh"); // :126
	*lo = 0; *hi = 0;
	if (type == COMPVHIP_CANNY_THRESHOLD_COMPARE_TO_GRADIENT) {
		gradientThresholds(tLow, tHigh, lo, hi);
		// the reference's SIMD leaves compare as signed int16 (intrin_avx2.cxx:144-147): thresholds above 32767 are an
		// artefact regime (g <= 24480) that this implementation rejects instead of reproducing
		if (*hi > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "thresholds above 32767 are not supported");
	}
	else if (type == COMPVHIP_CANNY_THRESHOLD_OTSU) { if (!(tHigh * 255.f < 32767.f)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "thresholds above 32767 are not supported"); }
	else if (!(tHigh * 255.f < 32767.f)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "thresholds above 32767 are not supported");
	return COMPVHIP_OK;
}

} // namespace

// ==================================================================================================================
namespace {
// The host workers of ONE compvhip_plan_houghkht call, shared by the controllers of all groups in flight: run(n, fn) queues fn(0) .. fn(n - 1) and blocks until
// every item has returned; the workers take items from the OLDEST job that still has some, so while one group waits for a GPU stage its controller sleeps
// and the workers link / sweep the frames of the other groups.  (Rounds 4-5 gave every controller a private pool of threads / controllers workers: all
// groups reached their GPU stages together and the workers of a waiting group idled -- 0.46 ms per 4K frame at 16 threads against 0.30 here.)
// The callers do not work along: `threads` workers are what runs, whatever the number of controllers.
thread_local int t_khtWorker = -1;   // index of the pool worker running the current item (-1: not a pool worker)
class KhtPool {
	struct Job {
		const std::function<void(size_t)>* fn; size_t n, next = 0;   // next: guarded by the pool's mutex
		std::atomic<size_t> left; char tag;
		Job(const std::function<void(size_t)>* f, size_t count, char t) : fn(f), n(count), left(count), tag(t) {}
	};
	struct Span { int worker; char tag; double t0, t1; };
public:
	explicit KhtPool(size_t threads)
	{
		trace_ = getenv("COMPVHIP_KHT_TRACE") != nullptr;   // lab: one line per item on stderr when the pool goes (worker, stage tag, start, end in ms)
		born_ = std::chrono::steady_clock::now();
		try { for (size_t t = 0; t < std::max<size_t>(1, threads); ++t) pool_.emplace_back([this, t] { t_khtWorker = static_cast<int>(t); loop(); }); }
		catch (...) { /* the system refused another thread: the ones that started share the work (none at all: run() works itself) */ }
	}
	~KhtPool()
	{
		{ std::lock_guard<std::mutex> g(m_); quit_ = true; }
		cv_.notify_all();
		for (std::thread& t : pool_) t.join();
		if (trace_) for (const Span& sp : spans_) fprintf(stderr, "khtpool w%02d %c %8.3f %8.3f\n", sp.worker, sp.tag, sp.t0, sp.t1);
	}
	size_t workers() const { return pool_.size(); }
	void run(size_t n, const std::function<void(size_t)>& fn, char tag = '?')
	{
		if (!n) return;
		if (pool_.empty()) { for (size_t i = 0; i < n; ++i) fn(i); return; }
		std::shared_ptr<Job> job = std::make_shared<Job>(&fn, n, tag);
		{
			// by stage priority (prio()), then by arrival
			std::lock_guard<std::mutex> g(m_);
			auto it = jobs_.begin();
			while (it != jobs_.end() && prio((*it)->tag) >= prio(tag)) ++it;
			jobs_.insert(it, job);
		}
		cv_.notify_all();
		if (tag == 'K') {
			// The prune items are short (0.2 ms) and gate their group's second GPU stage: when they are posted every worker is usually in the middle of a 3 ms
			// link of another group, and the group -- and, 4 ms later, the workers -- would wait for one to come free.  The posting controller works along.
			for (;;) {
				size_t i;
				{
					std::lock_guard<std::mutex> g(m_);
					if (job->next >= job->n) break;
					i = job->next++;
					if (job->next >= job->n) jobs_.erase(std::remove(jobs_.begin(), jobs_.end(), job), jobs_.end());
				}
				fn(i);
				job->left.fetch_sub(1);
			}
		}
		std::unique_lock<std::mutex> lk(m_);
		done_.wait(lk, [&] { return job->left.load() == 0; });   // every item has RETURNED: fn may go out of scope
	}
private:
	// the stage with the longest way to go first: a frame that is not linked yet still needs 3 ms of link + two GPU stages + 2 ms of sweep, a sweep is the end of its frame
	// and fills whatever gap is left (sweeps before links: 17.8 ms per 32 x 4K batch on 16 workers against 14.4)
	static int prio(char tag) { return tag == 'K' ? 3 : (tag == 'L' || tag == 'P') ? 2 : 1; }
	void loop()
	{
		for (;;) {
			std::shared_ptr<Job> job; size_t i = 0;
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_.wait(lk, [&] { return quit_ || !jobs_.empty(); });
				if (jobs_.empty()) return;   // quit_
				job = jobs_.front();
				i = job->next++;
				if (job->next >= job->n) jobs_.pop_front();
			}
			const auto t0 = std::chrono::steady_clock::now();
			(*job->fn)(i);
			if (trace_) {
				const auto t1 = std::chrono::steady_clock::now();
				std::lock_guard<std::mutex> g(m_);
				spans_.push_back({ t_khtWorker, job->tag, std::chrono::duration<double, std::milli>(t0 - born_).count(), std::chrono::duration<double, std::milli>(t1 - born_).count() });
			}
			if (job->left.fetch_sub(1) == 1) { std::lock_guard<std::mutex> g(m_); done_.notify_all(); }
		}
	}
	bool trace_ = false; std::chrono::steady_clock::time_point born_; std::vector<Span> spans_;
	std::vector<std::thread> pool_;
	std::mutex m_; std::condition_variable cv_, done_;
	std::deque<std::shared_ptr<Job>> jobs_;
	bool quit_ = false;
};

// CPUs this process may really use at once: min(affinity mask, cgroup CPU quota) -- a container can show 256 logical CPUs and own 16 (cpu.max "1600000 100000");
// twice as many busy threads as the quota only makes the kernel throttle all of them (round 5: 32 threads on a 16-CPU quota, sort + sweep 0.69 -> 1.97 ms per frame)
size_t hostCpuBudget()
{
	size_t n = std::thread::hardware_concurrency();
	if (!n) n = 4;
#if defined(__linux__)
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min<size_t>(n, static_cast<size_t>(c)); }
	if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {               // cgroup v2: "<quota|max> <period>"
		char q[64] = {}; long long period = 0;
		if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
			const long long quota = atoll(q);
			if (quota > 0) n = std::min<size_t>(n, static_cast<size_t>(std::max<long long>(1, quota / period)));
		}
		fclose(f);
	}
	else {
		long long quota = -1, period = 0;
		if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
		if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
		if (quota > 0 && period > 0) n = std::min<size_t>(n, static_cast<size_t>(std::max<long long>(1, quota / period)));
	}
#endif
	return std::max<size_t>(1, n);
}

template <typename T>
hipError_t growPinned(T*& ptr, size_t& cap, size_t want)
{
	if (cap >= want) return hipSuccess;
	if (ptr) (void)hipHostFree(ptr);
	ptr = nullptr; cap = 0;
	const size_t n = want + want / 4 + 1024;   // (frames of a stream resemble each other: no reallocation for a slightly denser batch)
	const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
	if (e == hipSuccess) cap = n;
	return e;
}
template <typename T>
hipError_t growDevice(compvhip_ctx* ctx, T*& ptr, size_t& cap, size_t want)
{
	if (cap >= want) return hipSuccess;
	dfree(ctx, ptr); cap = 0;
	const size_t n = want + want / 4 + 1024;
	const hipError_t e = dmalloc(ctx, &ptr, n);
	if (e == hipSuccess) cap = n;
	return e;
}
} // namespace


extern "C" {

int compvhip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int compvhip_ctx_create(compvhip_ctx** out, int device)
{
	if (!out) return COMPVHIP_E_INVALID_PARAMETER;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return COMPVHIP_E_NOT_INITIALIZED; // no GPU: fail loudly, no CPU fallback
	if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return COMPVHIP_E_HIP; }
	if (device >= n) return COMPVHIP_E_INVALID_PARAMETER;
	if (hipSetDevice(device) != hipSuccess) return COMPVHIP_E_HIP;
	compvhip_ctx* ctx = new (std::nothrow) compvhip_ctx();
	if (!ctx) return COMPVHIP_E_OUT_OF_MEMORY;
	ctx->device = device;
	if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return COMPVHIP_E_HIP; }
	*out = ctx;
	return COMPVHIP_OK;
}

void compvhip_ctx_destroy(compvhip_ctx* ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	if (ctx->hostPlan) compvhip_plan_destroy(ctx->hostPlan);
	dfree(ctx, ctx->dPacked); dfree(ctx, ctx->dHist);
	dfree(ctx, ctx->dIn); dfree(ctx, ctx->dOut); dfree(ctx, ctx->dCounts); dfree(ctx, ctx->dAccOut);
	khtScratchFree(ctx, ctx->kht);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

const char* compvhip_last_error(const compvhip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
long compvhip_live_allocations(const compvhip_ctx* ctx) { return ctx ? ctx->live.load() : 0; }

int compvhip_houghsht_to_cartesian(size_t W, size_t H, const compvhip_line* lines, size_t n, float* out)
{
	if (!W || !H || (n && (!lines || !out))) return COMPVHIP_E_INVALID_PARAMETER; // houghsht.cxx:268
	const float widthF = static_cast<float>(W), heightF = static_cast<float>(H);
	const float r = std::sqrt((widthF * widthF) + (heightF * heightF));
	for (size_t i = 0; i < n; ++i, out += 4) {
		const float rho = lines[i].rho, theta = lines[i].theta;
		if (theta == 0.f) { out[0] = rho; out[1] = r; out[2] = rho; out[3] = -r; continue; }
		const float c = libmCosf(theta), inv = 1.f / libmSinf(theta);
		out[0] = 0.f; out[1] = rho * inv;
		out[2] = widthF; out[3] = (rho - (widthF * c)) * inv;
	}
	return COMPVHIP_OK;
}

int compvhip_houghkht_to_cartesian(size_t W, size_t H, const compvhip_line* lines, size_t n, float* out)
{
	if (!W || !H || (n && (!lines || !out))) return COMPVHIP_E_INVALID_PARAMETER; // houghkht.cxx:453
	const float widthF = static_cast<float>(W), heightF = static_cast<float>(H);
	const float r = std::sqrt((widthF * widthF) + (heightF * heightF));
	const float halfW = widthF * 0.5f, halfH = heightF * 0.5f;
	for (size_t i = 0; i < n; ++i, out += 4) {
		const float rho = lines[i].rho, theta = lines[i].theta;
		if (theta == 0.f) { out[0] = rho + halfW; out[1] = r; out[2] = rho + halfW; out[3] = -r; continue; }
		const float c = libmCosf(theta) * halfW, inv = 1.f / libmSinf(theta);
		out[0] = 0.f; out[1] = ((rho + c) * inv) + halfH;
		out[2] = widthF; out[3] = ((rho - c) * inv) + halfH;
	}
	return COMPVHIP_OK;
}

int compvhip_houghsht_vote_grid(size_t W, size_t H, float thetaDeg, size_t frames, int* nx, int* ny, int* windowRows)
{
	size_t R = 0, T = 0; float st = 0.f;
	if (!nx || !ny || !windowRows || !frames) return COMPVHIP_E_INVALID_PARAMETER;
	const int rc = shtDims(W, H, thetaDeg, &R, &T, &st);
	if (rc) return rc;
	if (T < 5) return COMPVHIP_E_NOT_IMPLEMENTED;
	std::vector<int32_t> sq, cq, kt, rb;
	shtTables(thetaDeg, T, sq, cq);
	ShtTileArgs v = {};
	if (!planVoteTiles(W, H, frames, sq, cq, v, kt, rb)) return COMPVHIP_E_NOT_IMPLEMENTED;
	*nx = v.nx; *ny = v.ny; *windowRows = v.Rw;
	return COMPVHIP_OK;
}

int compvhip_host_cpu_budget(void)
{
	return static_cast<int>(std::min<size_t>(hostCpuBudget(), 0x7fffffff));
}

int compvhip_houghsht_dims(size_t W, size_t H, float thetaDeg, size_t* R, size_t* T, float* step)
{
	if (!R || !T) return COMPVHIP_E_INVALID_PARAMETER;
	return shtDims(W, H, thetaDeg, R, T, step);
}

int compvhip_houghkht_dims(size_t W, size_t H, float rho, float thetaDeg, size_t* rhoN, size_t* T)
{
	KhtAxes ax;
	if (!rhoN || !T || !khtAxes(W, H, rho, thetaDeg, ax)) return COMPVHIP_E_INVALID_PARAMETER;
	*rhoN = ax.rhoN; *T = ax.T;
	return COMPVHIP_OK;
}

// ---- plans -------------------------------------------------------------------------------------------------------
int compvhip_plan_create(compvhip_ctx* ctx, size_t W, size_t H, size_t S, size_t frames, float thetaDeg, compvhip_plan** out)
{
	if (!ctx || !out) return COMPVHIP_E_INVALID_PARAMETER;
	*out = nullptr;
	// W,H >= 3: the convolution rejects images smaller than the kernel (compv_math_convlt.h:100); int16 coordinates in
	// the reference's hysteresis stack bound W,H <= 32767 (canny_dete.cxx:617-621)
	if (W < 3 || H < 3 || W > 32767 || H > 32767 || S < W || (S & 7) || !frames || !(thetaDeg > 0.f))
		return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "plan geometry (need 3 <= W,H <= 32767, S >= W, S % 8 == 0, frames > 0)");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	compvhip_plan* p = new (std::nothrow) compvhip_plan();
	if (!p) return fail(ctx, COMPVHIP_E_OUT_OF_MEMORY, "plan");
	p->ctx = ctx; p->W = W; p->H = H; p->S = S; p->frames = frames; p->thetaDeg = thetaDeg;
	p->tilesX = static_cast<int>((W + 511) / 512);
	p->tilesY = static_cast<int>((H + 63) / 64);
	p->wb = p->tilesX * 16;
	p->bitsFrameStride = static_cast<size_t>(p->wb) * H;
	int rc = COMPVHIP_OK;
	do {
		if (dmalloc(ctx, &p->ebits, p->bitsFrameStride * frames) != hipSuccess) { rc = COMPVHIP_E_OUT_OF_MEMORY; break; }
		if (dmalloc(ctx, &p->ubits, p->bitsFrameStride * frames) != hipSuccess) { rc = COMPVHIP_E_OUT_OF_MEMORY; break; }
		// mask words past the last 496-column tile of a row are never written by the Canny kernel: they stay zero for ever
		if (hipMemset(p->ebits, 0, sizeof(uint32_t) * p->bitsFrameStride * frames) != hipSuccess) { rc = COMPVHIP_E_HIP; break; }
		if (hipMemset(p->ubits, 0, sizeof(uint32_t) * p->bitsFrameStride * frames) != hipSuccess) { rc = COMPVHIP_E_HIP; break; }
		// the fills run on the null stream; the plan's kernels may be enqueued on non-blocking streams that do not wait for it
		if (hipDeviceSynchronize() != hipSuccess) { rc = COMPVHIP_E_HIP; break; }
		{
			p->voteTiles = true;
			size_t R = 0, T = 0; float step = 0.f;
			if (p->voteTiles && shtDims(W, H, thetaDeg, &R, &T, &step) == COMPVHIP_OK && T >= 5) {
				std::vector<int32_t> sq, cq;
				shtTables(thetaDeg, T, sq, cq);
				if (!planVoteTiles(W, H, frames, sq, cq, p->vt, p->vtKt, p->vtRowBase)) p->voteTiles = false;
			}
			else p->voteTiles = false;
			p->lineBlocks = p->voteTiles ? sht_lines_blocks(static_cast<int>(R)) : 0;
			p->nCounts = (2 + static_cast<size_t>(p->voteTiles ? p->vt.tiles : 0) + static_cast<size_t>(p->lineBlocks)) * frames + (frames + 1) * kFrameSlot;
		}
		if (dmalloc(ctx, &p->counters, p->nCounts + kMaxRounds) != hipSuccess) { rc = COMPVHIP_E_OUT_OF_MEMORY; break; }
		p->edgeCounts = p->counters; p->lineCounts = p->counters + frames; p->tileCounts = p->counters + 2 * frames; p->blockCounts = p->counters + (2 + static_cast<size_t>(p->voteTiles ? p->vt.tiles : 0)) * frames; p->flags = p->counters + p->nCounts;
		p->frameTotals = p->blockCounts + static_cast<size_t>(p->lineBlocks) * frames; p->lineTotal = reinterpret_cast<unsigned int*>(p->frameTotals + frames * kFrameSlot);
		if (const char* e = getenv("COMPVHIP_RESOLVE_WRAP")) { const int v = atoi(e); if (v >= 8 && v <= kMaxRounds && (v & 3) == 0) p->maxRounds = v; }
		if (hipMemset(p->counters, 0, sizeof(int) * (p->nCounts + kMaxRounds)) != hipSuccess) { rc = COMPVHIP_E_HIP; break; }
		if (dmalloc(ctx, &p->thrDev, frames) != hipSuccess) { rc = COMPVHIP_E_OUT_OF_MEMORY; break; }
		if (dmalloc(ctx, &p->dirty, canny_resolve_dirty_bytes(static_cast<int>(H), p->wb, static_cast<int>(frames))) != hipSuccess) { rc = COMPVHIP_E_OUT_OF_MEMORY; break; }
		if (dmalloc(ctx, &p->sums, frames * kFrameSlot) != hipSuccess) { rc = COMPVHIP_E_OUT_OF_MEMORY; break; }
		if (hipHostMalloc(reinterpret_cast<void**>(&p->hFlags), sizeof(int) * 2 * (kAsyncDepth + 1)) != hipSuccess) { rc = COMPVHIP_E_OUT_OF_MEMORY; break; }
		p->hTotals = reinterpret_cast<unsigned int*>(p->hFlags + kAsyncDepth + 1);
		if (hipHostMalloc(reinterpret_cast<void**>(&p->hRounds), sizeof(int) * (kFrameSlot + 4) * kAsyncDepth, hipHostMallocMapped) != hipSuccess) { rc = COMPVHIP_E_OUT_OF_MEMORY; break; }
		if (hipHostGetDevicePointer(reinterpret_cast<void**>(&p->hRoundsDev), p->hRounds, 0) != hipSuccess) { rc = COMPVHIP_E_HIP; break; }
	} while (0);
	if (rc) { compvhip_plan_destroy(p); return fail(ctx, rc, "plan allocation"); }
	*out = p;
	return COMPVHIP_OK;
}

void compvhip_plan_destroy(compvhip_plan* p)
{
	if (!p) return;
	compvhip_ctx* ctx = p->ctx;
	(void)hipSetDevice(ctx->device);
	timelineClear(p);
	for (hipEvent_t e : p->eventPool) (void)hipEventDestroy(e);
	for (auto& stp : p->steps) if (stp.done) (void)hipEventDestroy(stp.done);
	dfree(ctx, p->dirty);
	dfree(ctx, p->ebits); dfree(ctx, p->ubits); dfree(ctx, p->counters); dfree(ctx, p->thrDev); dfree(ctx, p->sums); dfree(ctx, p->tmpOut);
	if (p->hFlags) (void)hipHostFree(p->hFlags);
	if (p->hRounds) (void)hipHostFree(p->hRounds);
	dfree(ctx, p->hist); dfree(ctx, p->otsu); dfree(ctx, p->blurTmp); dfree(ctx, p->grayTmp);
	for (KhtBatchState* b : p->khtBatch) khtBatchFree(ctx, b);
	p->khtBatch.clear();
	dfree(ctx, p->cosT); dfree(ctx, p->invSinT);
	dfree(ctx, p->dKt); dfree(ctx, p->dRowBase); dfree(ctx, p->partLo); dfree(ctx, p->partHi); dfree(ctx, p->colFlag);
	dfree(ctx, p->sinQ); dfree(ctx, p->cosQ); dfree(ctx, p->edges); dfree(ctx, p->acc);
	dfree(ctx, p->keysA); dfree(ctx, p->keysB); dfree(ctx, p->valsA); dfree(ctx, p->valsB); dfree(ctx, p->nmsFlags); dfree(ctx, p->chunkHist); dfree(ctx, p->strengthStart);
	dfree(ctx, p->nmsRange); dfree(ctx, p->reach); dfree(ctx, p->sortTemp);
	delete p;
}

int compvhip_plan_set_timing(compvhip_plan* p, int enabled)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	p->timing = (enabled >= 2 && enabled <= 4) ? enabled : (enabled != 0 ? 1 : 0);
	return COMPVHIP_OK;
}

int compvhip_plan_get_timing(compvhip_plan* p, const char** names, float* ms, int cap)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	(void)hipSetDevice(p->ctx->device);
	if (!p->timeline.empty()) {
		for (auto& t : p->timeline) (void)hipEventSynchronize(t.b);
		timelineCollect(p);
	}
	const int n = std::min<int>(cap, static_cast<int>(p->timingMs.size()));
	for (int i = 0; i < n; ++i) { if (names) names[i] = p->timingNames[i].c_str(); if (ms) ms[i] = p->timingMs[i]; }
	return n;
}

// The tile kernel writes E (edges so far), U (weak, unresolved) and the bytes of E; the resolve rounds finish the hysteresis on the
// masks and patch the promoted pixels into the byte map (p->patchOut; when in and out alias it is a scratch copy: a tile may still
// read the row halo a neighbour has overwritten).
static int planCannyImpl(compvhip_plan* p, const uint8_t* d_in, float tLow, float tHigh, int ksize, int type, uint8_t* d_edges, hipStream_t st,
                         bool waitConverged, bool clearTimeline = true)
{
	compvhip_ctx* ctx = p->ctx;
	if (!d_in || !d_edges) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null frame pointer");
	int lo, hi;
	int rc = validateCannyParams(ctx, tLow, tHigh, ksize, type, &lo, &hi);
	if (rc) return rc;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	if (p->timing && clearTimeline) timelineClear(p);
	uint8_t* out = d_edges;
	const size_t bytes = p->S * p->H * p->frames;
	const bool alias = (d_in < d_edges + bytes) && (d_edges < d_in + bytes);
	if (alias) {
		if (!p->tmpOut) HIPCHK(ctx, dmalloc(ctx, &p->tmpOut, bytes));
		out = p->tmpOut;
	}
	p->patchOut = out;
	p->copyBack = alias ? d_edges : nullptr;
	if (p->W < static_cast<size_t>(ksize) || p->H < static_cast<size_t>(ksize)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "image smaller than the kernel"); // compv_math_convlt.h:100
	rc = enqueueCanny(p, d_in, out, lo, hi, ksize, type, tLow, tHigh, st);
	if (rc) return rc;
	rc = enqueueResolve(p, p->patchOut, waitConverged ? kSpecRounds : p->specRounds, st);
	if (rc) return rc;
	if (waitConverged) {
		for (;;) {
			bool done = false;
			rc = resolveConverged(p, st, &done);
			if (rc) return rc;
			if (done) break;
			rc = enqueueResolve(p, p->patchOut, kSpecRounds, st);
			if (rc) return rc;
		}
	}
	if (p->copyBack) HIPCHK(ctx, hipMemcpyAsync(p->copyBack, out, bytes, hipMemcpyDeviceToDevice, st));
	p->bitsValid = true;
	return COMPVHIP_OK;
}

int compvhip_plan_canny(compvhip_plan* p, const uint8_t* d_in, float tLow, float tHigh, int ksize, int type, uint8_t* d_edges, void* stream)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	return planCannyImpl(p, d_in, tLow, tHigh, ksize, type, d_edges, static_cast<hipStream_t>(stream), true);
}

static int pixfmtBytes(int fmt)
{
	if (fmt < COMPVHIP_FMT_RGBA32 || fmt > COMPVHIP_FMT_Y) return 0;
	return fmt <= COMPVHIP_FMT_BGRA32 ? 4 : (fmt <= COMPVHIP_FMT_BGR24 ? 3 : (fmt == COMPVHIP_FMT_Y ? 1 : 2));
}

int compvhip_plan_grayscale(compvhip_plan* p, const uint8_t* d_in, int pixfmt, uint8_t* d_gray, void* stream)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	compvhip_ctx* ctx = p->ctx;
	if (!d_in || !d_gray) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null frame pointer");
	if (!pixfmtBytes(pixfmt)) return fail(ctx, COMPVHIP_E_NOT_IMPLEMENTED, "pixel format without a grayscale conversion"); // conv_to_grayscale.cxx:86-88
	if ((pixfmt == COMPVHIP_FMT_YUYV422 || pixfmt == COMPVHIP_FMT_UYVY422) && (p->W & 1))
		return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "packed 4:2:2 needs an even width");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (p->timing) timelineClear(p);
	GrayArgs a;
	a.in = d_in; a.out = d_gray; a.W = static_cast<int>(p->W); a.H = static_cast<int>(p->H); a.S = static_cast<int>(p->S); a.So = static_cast<int>(p->S);
	Stamp s(p, st, "gray_kernel");
	HIPCHK(ctx, launch_gray(a, pixfmt, static_cast<int>(p->frames), st));
	return COMPVHIP_OK;
}

int compvhip_plan_otsu(compvhip_plan* p, const uint8_t* d_gray, int32_t* d_thresholds, void* stream)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	compvhip_ctx* ctx = p->ctx;
	if (!d_gray || !d_thresholds) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null pointer");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	int rc = ensurePreproc(p);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (p->timing) timelineClear(p);
	Stamp s(p, st, "otsu_kernels");
	HIPCHK(ctx, launch_otsu(d_gray, static_cast<int>(p->W), static_cast<int>(p->H), static_cast<int>(p->S), p->S * p->H, static_cast<int>(p->frames), 0.5f, 1.f,
	                        p->hist, d_thresholds, nullptr, st));
	return COMPVHIP_OK;
}

static int checkFxpKernel(compvhip_ctx* ctx, size_t W, size_t H, const uint16_t* vt, const uint16_t* hz, size_t k)
{
	if (!vt || !hz || !(k & 1) || W < k || H < k) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "convolution kernel: null, even size or larger than the image"); // compv_math_convlt.h:100
	if (k < 3 || k > static_cast<size_t>(kFxpMaxTaps)) return fail(ctx, COMPVHIP_E_NOT_IMPLEMENTED, "fixed-point convolution supports kernel sizes 3..15");
	return COMPVHIP_OK;
}

int compvhip_plan_convlt1_fixedpoint(compvhip_plan* p, const uint8_t* d_in, const uint16_t* vtKern, const uint16_t* hzKern, size_t kernSize, uint8_t* d_out,
                                     void* stream)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	compvhip_ctx* ctx = p->ctx;
	if (!d_in || !d_out) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null frame pointer");
	int rc = checkFxpKernel(ctx, p->W, p->H, vtKern, hzKern, kernSize);
	if (rc) return rc;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const size_t span = p->S * p->H * p->frames;
	const bool alias = (d_in < d_out + span) && (d_out < d_in + span);
	if (alias && !p->blurTmp) HIPCHK(ctx, dmalloc(ctx, &p->blurTmp, span)); // only the in-place call needs the two-pass path
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (p->timing) timelineClear(p);
	Stamp s(p, st, "convlt_fxp_kernels");
	HIPCHK(ctx, launch_convlt_fxp(d_in, p->blurTmp, d_out, static_cast<int>(p->W), static_cast<int>(p->H), static_cast<int>(p->S), p->S * p->H,
	                              static_cast<int>(p->frames), vtKern, hzKern, static_cast<int>(kernSize), st));
	return COMPVHIP_OK;
}

int compvhip_plan_edge_dete(compvhip_plan* p, const uint8_t* d_in, int op, uint8_t* d_out, void* stream)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	compvhip_ctx* ctx = p->ctx;
	if (!d_in || !d_out) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null frame pointer");
	if (op != COMPVHIP_OP_SOBEL && op != COMPVHIP_OP_SCHARR && op != COMPVHIP_OP_PREWITT)
		return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "invalid detector id"); // edge_dete.cxx:246
	const size_t bytes = p->S * p->H * p->frames;
	if ((d_in < d_out + bytes) && (d_out < d_in + bytes)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "d_in and d_out must not alias");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (p->timing) timelineClear(p);
	EdgeDeteArgs a;
	a.in = d_in; a.out = d_out; a.gmax = p->sums;
	a.inFrameStride = p->S * p->H; a.outFrameStride = p->S * p->H;
	a.W = static_cast<int>(p->W); a.H = static_cast<int>(p->H); a.S = static_cast<int>(p->S); a.So = static_cast<int>(p->S);
	a.tilesX = p->tilesX; a.tilesY = p->tilesY;
	Stamp s(p, st, "edge_dete_kernels");
	HIPCHK(ctx, launch_edge_dete(a, op, static_cast<int>(p->frames), st));
	return COMPVHIP_OK;
}

// How many key slots the line sort covers (the reference sorts lines.size() elements, houghsht.cxx:241-249):
//   kSortAll   the whole capacity, unused slots zeroed by sht_lines_kernel -- the stream-ordered entry point, which may not wait for the device;
//   kSortExact the slots in use, read back behind sht_lines_kernel (one stream synchronisation) -- the synchronous step, which ends in one anyway;
//   otherwise  that many slots (a prediction: the asynchronous step; the caller checks it against the real total later).
constexpr size_t kSortAll = ~static_cast<size_t>(0), kSortExact = kSortAll - 1;

static int planShtImpl(compvhip_plan* p, const uint8_t* d_edges, int threshold, int maxLines, compvhip_line* d_lines, size_t lineCap, int32_t* d_counts,
                       hipStream_t st, bool clearTimeline, bool pairsOnly = false, size_t sortN = kSortAll)
{
	compvhip_ctx* ctx = p->ctx;
	if (threshold <= 0) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "threshold must be > 0"); // houghsht.cxx:82
	if (!d_edges && !p->bitsValid) return fail(ctx, COMPVHIP_E_INVALID_STATE, "no edge masks: run compvhip_plan_canny first or pass d_edges");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	int rc = ensureSht(p);
	if (rc) return rc;
	rc = ensureLineCap(p, std::max(lineCap, kMinLineCap));
	if (rc) return rc;
	if (p->timing && clearTimeline) timelineClear(p);
	const int frames = static_cast<int>(p->frames);
	if (d_edges) {
		Stamp s(p, st, "bytes_to_bits_kernel");
		HIPCHK(ctx, launch_bytes_to_bits(d_edges, static_cast<int>(p->W), static_cast<int>(p->H), static_cast<int>(p->S), p->S * p->H, p->ebits, p->wb,
		                                 p->bitsFrameStride, frames, st));
		p->bitsValid = false; // U masks no longer match
	}
	ShtArgs a = shtArgs(p, threshold);
	a.outCounts = d_counts;   // written by sht_lines_kernel together with the plan's own counts (was a device copy behind the sort)
	a.hostStep = p->stepHostSlot; a.stepFlags = p->flags;   // asynchronous steps: line total + round flags straight into the ticket's pinned slot
	const size_t capAll = p->lineCap * p->frames;
	const bool deviceSort = p->deviceSort && !pairsOnly;   // every size read on the device: nothing to predict, nothing to pad
	if (sortN != kSortExact) sortN = std::min(sortN, capAll);
	a.sortN = (pairsOnly || deviceSort || sortN == kSortExact) ? 0 : sortN;
	if (!p->countersFresh) HIPCHK(ctx, hipMemsetAsync(p->counters, 0, sizeof(int) * p->nCounts, st)); // edge, line and tile counts
	p->countersFresh = false;
	{ Stamp s(p, st, "sht_compact_kernel"); HIPCHK(ctx, launch_sht_compact_tiles(a, p->vt, frames, st)); }
	{ Stamp s(p, st, "sht_vote_kernel"); HIPCHK(ctx, launch_sht_vote_tiles(a, p->vt, frames, st)); }
	{ Stamp s(p, st, "sht_reduce_kernel"); HIPCHK(ctx, launch_sht_reduce_tiles(a, p->vt, frames, st)); }
	{ Stamp s(p, st, "sht_lines_kernel"); HIPCHK(ctx, launch_sht_lines(a, frames, st)); }
	// pairsOnly (the host entry point): stop at the (key, cell) pairs in emission order -- the caller orders them itself (referenceLineOrder)
	if (pairsOnly) {
		return COMPVHIP_OK;
	}
	if (deviceSort) {
		if (d_lines && lineCap) {
			Stamp s(p, st, "sht_sort_lines");
			ShtSortArgs q;
			q.sortedKeys = p->keysB; q.sortedVals = p->valsB; q.chunkHist = p->chunkHist; q.strengthStart = p->strengthStart; q.chunks = p->sortChunks;
			HIPCHK(ctx, launch_sht_sort_lines(a, q, frames, p->thetaStep, maxLines, d_lines, lineCap, st));
		}
		return COMPVHIP_OK;
	}
	// Invariant of the library-sort fallback: nothing reads keysB / valsB beyond lineTotal (the decode kernel stops at the clamped per-frame counts), so the
	// slots [sortN, capAll) are neither sorted nor cleared and may hold a previous step's pairs.
	if (sortN == kSortExact) {
		HIPCHK(ctx, hipMemcpyAsync(p->hTotals, p->lineTotal, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
		sortN = std::min(static_cast<size_t>(p->hTotals[0]), capAll);
	}
	if (sortN) {
		Stamp s(p, st, "sht_sort_lines");
		size_t tb = p->sortTempBytes;
		hipError_t e = sht_sort_pairs(p->sortTemp, tb, p->keysA, p->keysB, p->valsA, p->valsB, sortN, p->keyBits, st);
		if (e != hipSuccess) return fail(ctx, COMPVHIP_E_HIP, "radix sort", e);
	}
	if (d_lines && lineCap) {
		Stamp s(p, st, "sht_decode_kernel");
		HIPCHK(ctx, launch_sht_decode(p->keysB, p->valsB, p->lineCounts, p->lineCap, frames, static_cast<int>(p->T), static_cast<int>(p->W + p->H), p->thetaStep, maxLines,
		                              p->strengthBits, d_lines, lineCap, st));
	}
	return COMPVHIP_OK;
}

int compvhip_plan_houghsht(compvhip_plan* p, const uint8_t* d_edges, int threshold, int maxLines, compvhip_line* d_lines, size_t lineCap,
                           int32_t* d_counts, void* stream)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	return planShtImpl(p, d_edges, threshold, maxLines, d_lines, lineCap, d_counts, static_cast<hipStream_t>(stream), true);
}

// ---- the step: [grayscale ->] Canny (any kernel size / threshold mode) -> SHT [-> toCartesian], one enqueue ----------------------
static int checkStep(compvhip_plan* p, const StepParams& sp)
{
	compvhip_ctx* ctx = p->ctx;
	if (!sp.d_in || !sp.d_edges) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null frame pointer");
	if (sp.pixfmt != COMPVHIP_FMT_Y) {
		if (!pixfmtBytes(sp.pixfmt)) return fail(ctx, COMPVHIP_E_NOT_IMPLEMENTED, "pixel format without a grayscale conversion"); // conv_to_grayscale.cxx:86-88
		if ((sp.pixfmt == COMPVHIP_FMT_YUYV422 || sp.pixfmt == COMPVHIP_FMT_UYVY422) && (p->W & 1))
			return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "packed 4:2:2 needs an even width");
	}
	if (sp.d_cart && (!sp.d_lines || !sp.d_counts)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "toCartesian needs the line and count buffers");
	return COMPVHIP_OK;
}

// everything up to (not including) the convergence check, on `st`
static int enqueueStep(compvhip_plan* p, const StepParams& sp, hipStream_t st, bool clearTimeline)
{
	compvhip_ctx* ctx = p->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));   // before the scratch allocation below: the caller's current device may be another GPU
	const uint8_t* luma = sp.d_in;
	if (sp.pixfmt != COMPVHIP_FMT_Y) {
		// samples/hough_lines/main.cxx:102: CompVImage::convertGrayscale in front of everything else
		uint8_t* gray = sp.d_gray;
		if (!gray) {
			if (!p->grayTmp) HIPCHK(ctx, dmalloc(ctx, &p->grayTmp, p->S * p->H * p->frames));
			gray = p->grayTmp;
		}
		if (p->timing && clearTimeline) timelineClear(p);
		clearTimeline = false;
		GrayArgs a;
		a.in = sp.d_in; a.out = gray; a.W = static_cast<int>(p->W); a.H = static_cast<int>(p->H); a.S = static_cast<int>(p->S); a.So = static_cast<int>(p->S);
		Stamp s(p, st, "gray_kernel");
		HIPCHK(ctx, launch_gray(a, sp.pixfmt, static_cast<int>(p->frames), st));
		luma = gray;
	}
	int rc = planCannyImpl(p, luma, sp.tLow, sp.tHigh, sp.ksize, sp.thresholdType, sp.d_edges, st, false, clearTimeline);
	if (rc) return rc;
	if (sp.d_otsu && sp.thresholdType == COMPVHIP_CANNY_THRESHOLD_OTSU)
		HIPCHK(ctx, hipMemcpyAsync(sp.d_otsu, p->otsu, sizeof(int32_t) * p->frames, hipMemcpyDeviceToDevice, st));
	return COMPVHIP_OK;
}

static int toCartesianImpl(compvhip_plan* p, const compvhip_line* d_lines, const int32_t* d_counts, size_t lineCap, int maxLines, float* d_cart, void* stream);

static int enqueueStepTail(compvhip_plan* p, const StepParams& sp, hipStream_t st, size_t sortN)
{
	int rc = planShtImpl(p, nullptr, sp.threshold, sp.maxLines, sp.d_lines, sp.lineCap, sp.d_counts, st, false, false, sortN);
	if (rc) return rc;
	if (sp.d_cart) {
		rc = toCartesianImpl(p, sp.d_lines, sp.d_counts, sp.lineCap, sp.maxLines, sp.d_cart, st);   // d_counts = the UNCUT counts; only min(count, lineCap, maxLines) slots were decoded
		if (rc) return rc;
	}
	return COMPVHIP_OK;
}

static int runStepSync(compvhip_plan* p, const StepParams& sp, hipStream_t st)
{
	// Everything is enqueued back to back (speculative resolve rounds included); the convergence flag is checked once at the end and,
	// in the rare case the hysteresis needed more rounds, the tail is replayed.
	int rc = enqueueStep(p, sp, st, true);
	if (rc) return rc;
	const size_t bytes = p->S * p->H * p->frames;
	for (;;) {
		rc = enqueueStepTail(p, sp, st, kSortExact);
		if (rc) return rc;
		bool done = false;
		rc = resolveConverged(p, st, &done);
		if (rc) return rc;
		if (done) break;
		do {
			rc = enqueueResolve(p, p->patchOut, kSpecRounds, st);
			if (rc) return rc;
			rc = resolveConverged(p, st, &done);
			if (rc) return rc;
		} while (!done);
		if (p->copyBack) HIPCHK(p->ctx, hipMemcpyAsync(p->copyBack, p->patchOut, bytes, hipMemcpyDeviceToDevice, st));
		// replay of the Hough stage on the now final masks
	}
	return COMPVHIP_OK;
}

// The step without its host round trip: the hysteresis flag of the last speculative round travels to a pinned host slot behind
// the step's kernels and is looked at by compvhip_plan_wait(), normally while the NEXT step is already running.
static int runStepAsync(compvhip_plan* p, const StepParams& sp, hipStream_t st, int* ticket)
{
	compvhip_ctx* ctx = p->ctx;
	*ticket = -1;
	int slot = -1;
	for (int i = 0; i < kAsyncDepth; ++i) if (!p->steps[i].used) { slot = i; break; }
	if (slot < 0) return fail(ctx, COMPVHIP_E_INVALID_STATE, "too many steps in flight: call compvhip_plan_wait first");
	const size_t bytes = p->S * p->H * p->frames;
	const size_t inBytes = bytes * static_cast<size_t>(sp.pixfmt == COMPVHIP_FMT_Y ? 1 : pixfmtBytes(sp.pixfmt));
	if ((sp.d_in < sp.d_edges + bytes) && (sp.d_edges < sp.d_in + inBytes))
		return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "the asynchronous step needs distinct input and edge buffers"); // a replay re-reads d_in
	compvhip_plan::AsyncStep& stp = p->steps[slot];
	HIPCHK(ctx, hipSetDevice(ctx->device));
	if (!stp.done) HIPCHK(ctx, hipEventCreateWithFlags(&stp.done, hipEventDisableTiming));
	// timing events of asynchronous steps accumulate until compvhip_plan_get_timing reads them; beyond kMaxTimeline entries the oldest go
	if (p->timeline.size() > kMaxTimeline) timelineClear(p);
	int rc = enqueueStep(p, sp, st, false);
	if (rc) return rc;
	// (capacities first: ensureLineCap may grow the key buffers and reset recentN -- totals clamped to the old capacity predict nothing)
	rc = ensureSht(p);
	if (rc) return rc;
	rc = ensureLineCap(p, std::max(sp.lineCap, kMinLineCap));
	if (rc) return rc;
	// the sorted range: the largest total of the plan's recent steps + 1/16 + 4096 slots (nothing seen yet, or just reset: everything)
	size_t sortN = kSortAll;
	if (p->recentN > 0) {
		unsigned int m = 0;
		for (int i = 0; i < std::min(p->recentN, 8); ++i) m = std::max(m, p->recentTotals[i]);
		sortN = static_cast<size_t>(m) + (m >> 4) + 4096;
	}
	if (p->deviceSort) sortN = kSortAll;   // sized on the device: no prediction to check
	sortN = std::min(sortN, p->lineCap * p->frames);
	// What compvhip_plan_wait needs -- the step's line total and its first four round flags -- is written by sht_lines_kernel straight into the ticket's pinned,
	// device-mapped slot: no copy behind the step's kernels (rounds 2-5: three device-to-host copies, a device copy of the counts and a fill per step -- 30 us
	// of stream operations between the last kernel of a step and the first of the next on a lane, 5 now)
	static_assert(kMaxRounds >= 4 && kSpecRounds <= 4, "the round flags compvhip_plan_wait looks at");
	if (p->roundsUsed > 4) return fail(ctx, COMPVHIP_E_INVALID_STATE, "more speculative rounds than flags the step reports");
	p->stepHostSlot = p->hRoundsDev + (kFrameSlot + 4) * slot;
	rc = enqueueStepTail(p, sp, st, sortN);
	p->stepHostSlot = nullptr;
	if (rc) return rc;
	HIPCHK(ctx, hipEventRecord(stp.done, st));
	stp.used = true; stp.replay = false; stp.seq = ++p->stepSeq; stp.stream = st; stp.sp = sp; stp.sortN = sortN; stp.rounds = p->roundsUsed;
	*ticket = slot;
	return COMPVHIP_OK;
}

static StepParams classicStep(const uint8_t* d_in, float tLow, float tHigh, int threshold, int maxLines, uint8_t* d_edges, compvhip_line* d_lines, size_t lineCap,
                              int32_t* d_counts)
{
	StepParams sp;
	sp.d_in = d_in; sp.tLow = tLow; sp.tHigh = tHigh; sp.threshold = threshold; sp.maxLines = maxLines; sp.d_edges = d_edges; sp.d_lines = d_lines;
	sp.lineCap = lineCap; sp.d_counts = d_counts;
	return sp;
}

int compvhip_plan_pipeline(compvhip_plan* p, const uint8_t* d_in, float tLow, float tHigh, int threshold, int maxLines, uint8_t* d_edges,
                           compvhip_line* d_lines, size_t lineCap, int32_t* d_counts, void* stream)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	const StepParams sp = classicStep(d_in, tLow, tHigh, threshold, maxLines, d_edges, d_lines, lineCap, d_counts);
	int rc = checkStep(p, sp);
	if (rc) return rc;
	return runStepSync(p, sp, static_cast<hipStream_t>(stream));
}

int compvhip_plan_pipeline_async(compvhip_plan* p, const uint8_t* d_in, float tLow, float tHigh, int threshold, int maxLines, uint8_t* d_edges,
                                 compvhip_line* d_lines, size_t lineCap, int32_t* d_counts, void* stream, int* ticket)
{
	if (!p || !ticket) return COMPVHIP_E_INVALID_PARAMETER;
	const StepParams sp = classicStep(d_in, tLow, tHigh, threshold, maxLines, d_edges, d_lines, lineCap, d_counts);
	*ticket = -1;
	int rc = checkStep(p, sp);
	if (rc) return rc;
	return runStepAsync(p, sp, static_cast<hipStream_t>(stream), ticket);
}

int compvhip_plan_pipeline_ex(compvhip_plan* p, const uint8_t* d_in, const compvhip_pipeline_opts* o, uint8_t* d_edges, compvhip_line* d_lines, size_t lineCap,
                              int32_t* d_counts, void* stream, int* ticket)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	if (ticket) *ticket = -1;
	if (!o) return fail(p->ctx, COMPVHIP_E_INVALID_PARAMETER, "null options");
	StepParams sp = classicStep(d_in, o->tLow, o->tHigh, o->threshold, o->maxLines, d_edges, d_lines, lineCap, d_counts);
	sp.ksize = o->ksize ? o->ksize : 3; sp.thresholdType = o->thresholdType; sp.pixfmt = o->pixfmt;
	sp.d_gray = o->d_gray; sp.d_otsu = o->d_otsu; sp.d_cart = o->d_cart;
	int rc = checkStep(p, sp);
	if (rc) return rc;
	return ticket ? runStepAsync(p, sp, static_cast<hipStream_t>(stream), ticket) : runStepSync(p, sp, static_cast<hipStream_t>(stream));
}

int compvhip_plan_wait(compvhip_plan* p, int ticket)
{
	if (!p || ticket < 0 || ticket >= kAsyncDepth || !p->steps[ticket].used) return COMPVHIP_E_INVALID_PARAMETER;
	compvhip_ctx* ctx = p->ctx;
	compvhip_plan::AsyncStep& stp = p->steps[ticket];
	HIPCHK(ctx, hipSetDevice(ctx->device));
	HIPCHK(ctx, hipEventSynchronize(stp.done));
	stp.used = false;
	const int* stepOut = p->hRounds + (kFrameSlot + 4) * ticket;   // what the step's one device-to-host copy brought: line total, then (kFrameSlot ints on) the round flags
	const unsigned int total = static_cast<unsigned int>(stepOut[0]);
	p->recentTotals[p->recentN++ & 7] = total;
	if (p->recentN >= 16) p->recentN -= 8;   // the ring index keeps counting, "entries seen" saturates at 8
	const bool sorted = static_cast<size_t>(total) <= stp.sortN;   // the predicted range covered every line of the step
	{
		// rounds this step needed = the first round that changed nothing, inclusive (more than were enqueued: one more than that, at least)
		const int* rf = stepOut + kFrameSlot;
		int needed = std::min(stp.rounds, 4) + 1;
		for (int i = 0; i < std::min(stp.rounds, 4); ++i) if (rf[i] == 0) { needed = i + 1; break; }
		p->recentRounds[p->recentRoundsN++ & 7] = static_cast<unsigned char>(std::min(needed, 255));
		if (p->recentRoundsN >= 16) p->recentRoundsN -= 8;
		int m = 2;
		for (int i = 0; i < std::min(p->recentRoundsN, 8); ++i) m = std::max<int>(m, p->recentRounds[i]);
		p->specRounds = (p->recentRoundsN >= 4) ? std::min(m, kSpecRounds) : kSpecRounds;   // a few steps first, then as many as they needed
		if (getenv("COMPVHIP_TRACE_ROUNDS")) fprintf(stderr, "plan %p ticket %d: rounds enqueued %d, flags %d %d %d %d, needed %d -> next %d\n", (void*)p, ticket, stp.rounds, rf[0], rf[1], rf[2], rf[3], needed, p->specRounds);
	}
	const int lastFlag = stepOut[kFrameSlot + std::max(0, std::min(stp.rounds, 4) - 1)];   // the flag of the step's last speculative round
	if (lastFlag == 0 && sorted && !stp.replay) return COMPVHIP_OK; // the speculative rounds reached the fixed point and the sort covered the lines (the usual case)
	// Rare: the hysteresis of this step needed more rounds than were enqueued (or it produced more lines than the sorted range held), and a later step may
	// already have reused the plan's masks.  Let the stream drain and run the step again, synchronously, from its (unmodified) input.  The replay writes this
	// step's output buffers AFTER the later steps of the plan ran: if they share those buffers (a caller that only consumes the newest
	// result), their results are gone -- every step enqueued after this one is therefore replayed as well when it is waited for, in order.
	HIPCHK(ctx, hipStreamSynchronize(stp.stream));
	for (int i = 0; i < kAsyncDepth; ++i)
		if (p->steps[i].used && p->steps[i].seq > stp.seq) p->steps[i].replay = true;
	stp.replay = false;
	return runStepSync(p, stp.sp, stp.stream);
}

int compvhip_plan_to_cartesian(compvhip_plan* p, const compvhip_line* d_lines, const int32_t* d_counts, size_t lineCap, float* d_cart, void* stream)
{
	return toCartesianImpl(p, d_lines, d_counts, lineCap, 0, d_cart, stream);
}

// maxLines > 0: d_counts holds the uncut line counts of a step that decoded only the first maxLines lines of every frame
static int toCartesianImpl(compvhip_plan* p, const compvhip_line* d_lines, const int32_t* d_counts, size_t lineCap, int maxLines, float* d_cart, void* stream)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	compvhip_ctx* ctx = p->ctx;
	if (!d_lines || !d_counts || !d_cart) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null pointer");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	int rc = ensureSht(p);
	if (rc) return rc;
	const float widthF = static_cast<float>(p->W), heightF = static_cast<float>(p->H);
	const float r = std::sqrt((widthF * widthF) + (heightF * heightF)); // houghsht.cxx:570
	HIPCHK(ctx, launch_sht_cartesian(d_lines, d_counts, lineCap, maxLines, static_cast<int>(p->frames), static_cast<int>(p->T), p->cosT, p->invSinT, widthF, r, d_cart,
	                                 static_cast<hipStream_t>(stream)));
	return COMPVHIP_OK;
}

int compvhip_plan_acc(compvhip_plan* p, size_t frame, const uint16_t** d_acc, size_t* R, size_t* T, size_t* accPitch)
{
	if (!p || !p->shtReady || frame >= p->frames) return COMPVHIP_E_INVALID_PARAMETER;
	if (d_acc) *d_acc = p->acc + frame * p->accFrameStride;
	if (R) *R = p->R;
	if (T) *T = p->T;
	if (accPitch) *accPitch = static_cast<size_t>(p->accPitch);
	return COMPVHIP_OK;
}

int compvhip_plan_acc_export(compvhip_plan* p, size_t frame, int32_t* d_out, size_t outStride, void* stream)
{
	if (!p || !p->shtReady || frame >= p->frames || !d_out || outStride < p->T) return COMPVHIP_E_INVALID_PARAMETER;
	compvhip_ctx* ctx = p->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	HIPCHK(ctx, launch_sht_acc_transpose(p->acc + frame * p->accFrameStride, static_cast<int>(p->R), static_cast<int>(p->T), p->accPitch, d_out, outStride,
	                                     static_cast<hipStream_t>(stream)));
	return COMPVHIP_OK;
}

int compvhip_plan_edge_counts(compvhip_plan* p, const int32_t** d_edge_counts)
{
	if (!p || !p->shtReady || !d_edge_counts) return COMPVHIP_E_INVALID_PARAMETER;
	*d_edge_counts = p->edgeCounts;
	return COMPVHIP_OK;
}

// ---- host entry points -------------------------------------------------------------------------------------------
static int hostPlan(compvhip_ctx* ctx, size_t W, size_t H, float thetaDeg, compvhip_plan** out)
{
	const size_t S = alignUp(W, 64);
	compvhip_plan* p = ctx->hostPlan;
	if (p && (p->W != W || p->H != H || p->thetaDeg != thetaDeg)) { compvhip_plan_destroy(p); ctx->hostPlan = p = nullptr; }
	if (!p) {
		int rc = compvhip_plan_create(ctx, W, H, S, 1, thetaDeg, &p);
		if (rc) return rc;
		ctx->hostPlan = p;
	}
	const size_t bytes = S * H;
	if (ctx->dInBytes < bytes) { dfree(ctx, ctx->dIn); HIPCHK(ctx, dmalloc(ctx, &ctx->dIn, bytes)); ctx->dInBytes = bytes; }
	if (ctx->dOutBytes < bytes) { dfree(ctx, ctx->dOut); HIPCHK(ctx, dmalloc(ctx, &ctx->dOut, bytes)); ctx->dOutBytes = bytes; }
	*out = p;
	return COMPVHIP_OK;
}

static int checkImage(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S, const void* out, size_t So)
{
	if (!ctx) return COMPVHIP_E_INVALID_PARAMETER;
	if (!in || !out || S < W || So < W) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null image or stride < width");
	if (W < 3 || H < 3 || W > 32767 || H > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "image size out of range (3..32767)");
	return COMPVHIP_OK;
}

int compvhip_canny_u8(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S, float tLow, float tHigh, int ksize, int type,
                      uint8_t* out, size_t So)
{
	int rc = checkImage(ctx, in, W, H, S, out, So);
	if (rc) return rc;
	int lo, hi;
	rc = validateCannyParams(ctx, tLow, tHigh, ksize, type, &lo, &hi);
	if (rc) return rc;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	compvhip_plan* p = nullptr;
	rc = hostPlan(ctx, W, H, ctx->hostPlan ? ctx->hostPlan->thetaDeg : 1.f, &p);
	if (rc) return rc;
	HIPCHK(ctx, hipMemcpy2DAsync(ctx->dIn, p->S, in, S, W, H, hipMemcpyHostToDevice, ctx->stream));
	rc = compvhip_plan_canny(p, ctx->dIn, tLow, tHigh, ksize, type, ctx->dOut, ctx->stream);
	if (rc) return rc;
	HIPCHK(ctx, hipMemcpy2DAsync(out, So, ctx->dOut, p->S, W, H, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return COMPVHIP_OK;
}

int compvhip_edge_dete_u8(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S, int op, uint8_t* out, size_t So)
{
	int rc = checkImage(ctx, in, W, H, S, out, So);
	if (rc) return rc;
	if (op != COMPVHIP_OP_SOBEL && op != COMPVHIP_OP_SCHARR && op != COMPVHIP_OP_PREWITT)
		return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "invalid detector id"); // edge_dete.cxx:246
	HIPCHK(ctx, hipSetDevice(ctx->device));
	compvhip_plan* p = nullptr;
	rc = hostPlan(ctx, W, H, ctx->hostPlan ? ctx->hostPlan->thetaDeg : 1.f, &p);
	if (rc) return rc;
	HIPCHK(ctx, hipMemcpy2DAsync(ctx->dIn, p->S, in, S, W, H, hipMemcpyHostToDevice, ctx->stream));
	EdgeDeteArgs a;
	a.in = ctx->dIn; a.out = ctx->dOut; a.gmax = p->sums;
	a.inFrameStride = p->S * H; a.outFrameStride = p->S * H;
	a.W = static_cast<int>(W); a.H = static_cast<int>(H); a.S = static_cast<int>(p->S); a.So = static_cast<int>(p->S);
	a.tilesX = p->tilesX; a.tilesY = p->tilesY;
	HIPCHK(ctx, launch_edge_dete(a, op, 1, ctx->stream));
	HIPCHK(ctx, hipMemcpy2DAsync(out, So, ctx->dOut, p->S, W, H, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return COMPVHIP_OK;
}

int compvhip_gauss_kernel_fixedpoint(size_t size, float sigma, uint16_t* kernel)
{
	// compv_math_gauss.h:24-55 with T = float (note the float/double mix), then compv_math_convlt.h:88
	if (!kernel || !(size & 1) || size > 255 || !(sigma > 0.f)) return COMPVHIP_E_INVALID_PARAMETER;
	float f[255];
	const size_t half = size >> 1;
	const float sigma2_times2 = static_cast<float>(2 * (sigma * sigma));
	const float a = static_cast<float>(1 / std::sqrt(3.14159265358979323846 * sigma2_times2));
	float sum = a;
	f[half] = a;
	for (size_t x = 1; x <= half; ++x) {
		const float k = static_cast<float>(a * std::exp(-static_cast<double>((x * x) / sigma2_times2)));
		f[x + half] = k; f[half - x] = k;
		sum += (k + k);
	}
	sum = 1 / sum;
	for (size_t x = 0; x < size; ++x) { f[x] *= sum; kernel[x] = static_cast<uint16_t>(f[x] * 0xffff); }
	return COMPVHIP_OK;
}

int compvhip_convlt1_fixedpoint_u8(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S, const uint16_t* vtKern, const uint16_t* hzKern,
                                   size_t kernSize, uint8_t* out, size_t So)
{
	if (!ctx) return COMPVHIP_E_INVALID_PARAMETER;
	if (!in || !out || S < W || So < W || !W || !H || W > 32767 || H > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null image, stride < width or size out of range");
	int rc = checkFxpKernel(ctx, W, H, vtKern, hzKern, kernSize);
	if (rc) return rc;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const size_t Sd = alignUp(W, 64), bytes = Sd * H;
	if (ctx->dInBytes < bytes) { dfree(ctx, ctx->dIn); HIPCHK(ctx, dmalloc(ctx, &ctx->dIn, bytes)); ctx->dInBytes = bytes; }
	if (ctx->dOutBytes < bytes) { dfree(ctx, ctx->dOut); HIPCHK(ctx, dmalloc(ctx, &ctx->dOut, bytes)); ctx->dOutBytes = bytes; }
	HIPCHK(ctx, hipMemcpy2DAsync(ctx->dIn, Sd, in, S, W, H, hipMemcpyHostToDevice, ctx->stream));
	// dIn -> dOut with the fused kernel (no intermediate: the two staging buffers never alias)
	HIPCHK(ctx, launch_convlt_fxp(ctx->dIn, nullptr, ctx->dOut, static_cast<int>(W), static_cast<int>(H), static_cast<int>(Sd), bytes, 1, vtKern, hzKern,
	                              static_cast<int>(kernSize), ctx->stream));
	HIPCHK(ctx, hipMemcpy2DAsync(out, So, ctx->dOut, Sd, W, H, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return COMPVHIP_OK;
}

// CompVMathConvlt::convlt1<uint8_t | int16_t, int16_t, int16_t> (compv_math_convlt.h:26-28,37-39,98-292): the separable integer correlation the
// gradient is made of, stand-alone.  The device buffers are private to the call (the operator is not on the per-frame hot path: there it is
// fused into the tile kernels); S, So in elements.
static int convlt1I16(compvhip_ctx* ctx, const void* in, bool inIsU8, size_t W, size_t H, size_t S, const int16_t* vtKern, const int16_t* hzKern, size_t kernSize,
                      int16_t* out, size_t So)
{
	if (!ctx) return COMPVHIP_E_INVALID_PARAMETER;
	if (!in || !out || !vtKern || !hzKern || S < W || So < W || !(kernSize & 1) || W < kernSize || H < kernSize)
		return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "convolution: null pointer, stride < width, even kernel size or image smaller than the kernel"); // compv_math_convlt.h:100
	if (kernSize > static_cast<size_t>(kFxpMaxTaps)) return fail(ctx, COMPVHIP_E_NOT_IMPLEMENTED, "integer convolution supports kernel sizes 1..15");
	if (W > 32767 || H > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "image size out of range");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const size_t es = inIsU8 ? 1 : 2;
	const size_t Sd = alignUp(W, 64);
	uint8_t* dIn = nullptr; int16_t* dTmp = nullptr; int16_t* dOut = nullptr;
	int rc = COMPVHIP_OK;
	do {
		if (dmalloc(ctx, &dIn, Sd * H * es) != hipSuccess || dmalloc(ctx, &dTmp, Sd * H) != hipSuccess || dmalloc(ctx, &dOut, Sd * H) != hipSuccess) { rc = fail(ctx, COMPVHIP_E_OUT_OF_MEMORY, "convolution buffers"); break; }
		hipError_t e = hipMemcpy2DAsync(dIn, Sd * es, in, S * es, W * es, H, hipMemcpyHostToDevice, ctx->stream);
		if (e == hipSuccess) e = launch_convlt_i16(dIn, inIsU8, dTmp, dOut, static_cast<int>(W), static_cast<int>(H), static_cast<int>(Sd), static_cast<int>(Sd), vtKern, hzKern,
		                                           static_cast<int>(kernSize), ctx->stream);
		if (e == hipSuccess) e = hipMemcpy2DAsync(out, So * 2, dOut, Sd * 2, W * 2, H, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) rc = fail(ctx, COMPVHIP_E_HIP, "integer convolution", e);
	} while (0);
	dfree(ctx, dIn); dfree(ctx, dTmp); dfree(ctx, dOut);
	return rc;
}

int compvhip_convlt1_8u16s16s(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S, const int16_t* vtKern, const int16_t* hzKern, size_t kernSize,
                              int16_t* out, size_t So)
{
	return convlt1I16(ctx, in, true, W, H, S, vtKern, hzKern, kernSize, out, So);
}

int compvhip_convlt1_16s16s16s(compvhip_ctx* ctx, const int16_t* in, size_t W, size_t H, size_t S, const int16_t* vtKern, const int16_t* hzKern, size_t kernSize,
                               int16_t* out, size_t So)
{
	return convlt1I16(ctx, in, false, W, H, S, vtKern, hzKern, kernSize, out, So);
}

int compvhip_grayscale_u8(compvhip_ctx* ctx, const uint8_t* in, int pixfmt, size_t W, size_t H, size_t S, uint8_t* out, size_t So)
{
	if (!ctx) return COMPVHIP_E_INVALID_PARAMETER;
	if (!in || !out || S < W || So < W || !W || !H || W > 32767 || H > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null image, stride < width or size out of range");
	const int bpp = pixfmtBytes(pixfmt);
	if (!bpp) return fail(ctx, COMPVHIP_E_NOT_IMPLEMENTED, "pixel format without a grayscale conversion"); // conv_to_grayscale.cxx:86-88
	if ((pixfmt == COMPVHIP_FMT_YUYV422 || pixfmt == COMPVHIP_FMT_UYVY422) && (W & 1)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "packed 4:2:2 needs an even width");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const size_t Sd = alignUp(W, 64);
	const size_t inBytes = Sd * H * bpp, outBytes = Sd * H;
	if (ctx->dPackedBytes < inBytes) { dfree(ctx, ctx->dPacked); HIPCHK(ctx, dmalloc(ctx, &ctx->dPacked, inBytes)); ctx->dPackedBytes = inBytes; }
	if (ctx->dOutBytes < outBytes) { dfree(ctx, ctx->dOut); HIPCHK(ctx, dmalloc(ctx, &ctx->dOut, outBytes)); ctx->dOutBytes = outBytes; }
	HIPCHK(ctx, hipMemcpy2DAsync(ctx->dPacked, Sd * bpp, in, S * bpp, W * bpp, H, hipMemcpyHostToDevice, ctx->stream));
	GrayArgs a;
	a.in = ctx->dPacked; a.out = ctx->dOut; a.W = static_cast<int>(W); a.H = static_cast<int>(H); a.S = static_cast<int>(Sd); a.So = static_cast<int>(Sd);
	HIPCHK(ctx, launch_gray(a, pixfmt, 1, ctx->stream));
	HIPCHK(ctx, hipMemcpy2DAsync(out, So, ctx->dOut, Sd, W, H, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return COMPVHIP_OK;
}

int compvhip_otsu_u8(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S, double* threshold)
{
	if (!ctx) return COMPVHIP_E_INVALID_PARAMETER;
	if (!in || !threshold || S < W || !W || !H || W > 32767 || H > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null image, stride < width or size out of range"); // threshold.cxx:54
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const size_t Sd = alignUp(W, 64);
	const size_t bytes = Sd * H;
	if (ctx->dInBytes < bytes) { dfree(ctx, ctx->dIn); HIPCHK(ctx, dmalloc(ctx, &ctx->dIn, bytes)); ctx->dInBytes = bytes; }
	if (!ctx->dHist) HIPCHK(ctx, dmalloc(ctx, &ctx->dHist, static_cast<size_t>(256) * kOtsuMaxChunks + 1));
	HIPCHK(ctx, hipMemcpy2DAsync(ctx->dIn, Sd, in, S, W, H, hipMemcpyHostToDevice, ctx->stream));
	int32_t* dT = reinterpret_cast<int32_t*>(ctx->dHist + static_cast<size_t>(256) * kOtsuMaxChunks);
	HIPCHK(ctx, launch_otsu(ctx->dIn, static_cast<int>(W), static_cast<int>(H), static_cast<int>(Sd), bytes, 1, 0.5f, 1.f, ctx->dHist, dT, nullptr, ctx->stream));
	int32_t t = 0;
	HIPCHK(ctx, hipMemcpyAsync(&t, dT, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	*threshold = static_cast<double>(t);
	return COMPVHIP_OK;
}

// CompVHoughSht::process ends with std::sort(lines, strength >) and keeps the first maxLines (houghsht.cxx:241-249).  std::sort is
// unstable, but deterministic for one libstdc++ and one input order, and the input order is nms_apply's emission order: accumulator
// rows ascending, columns ascending (:546-562; per-thread vectors are concatenated in row order, :228-234).  Re-creating that order
// and calling the same std::sort gives the reference's list element by element -- callers such as CompVCalibCamera (line grouping,
// core/calib/compv_core_calib_camera.cxx:200-) depend on the order inside equal-strength groups.  The permutation only depends on
// the strengths, so it is computed on (strength, index) pairs.
static void referenceLineOrder(const std::vector<uint32_t>& keys, const std::vector<uint32_t>& cells, uint32_t strengthMask, size_t T, long long barrier, float thetaStep,
                               std::vector<compvhip_line>& lines)
{
	// keys / cells arrive in emission order (ascending cell): exactly the array the reference sorts
	const size_t n = keys.size();
	struct Item { int32_t strength; uint32_t idx; };
	std::vector<Item> items(n);
	for (size_t i = 0; i < n; ++i) { items[i].strength = static_cast<int32_t>(keys[i] & strengthMask); items[i].idx = static_cast<uint32_t>(i); }
	std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.strength > b.strength; });
	lines.resize(n);
	for (size_t i = 0; i < n; ++i) {
		const uint32_t cell = cells[items[i].idx];
		const uint32_t row = cell / static_cast<uint32_t>(T), col = cell - row * static_cast<uint32_t>(T);
		compvhip_line& l = lines[i];
		l.rho = static_cast<float>(barrier - static_cast<long long>(row));   // houghsht.cxx:661
		l.theta = static_cast<float>(col) * thetaStep;                       // houghsht.cxx:662 (one rounded f32 product: -ffp-contract=off)
		l.strength = items[i].strength; l.row = static_cast<int32_t>(row); l.col = static_cast<int32_t>(col);
	}
}

int compvhip_houghsht_u8(compvhip_ctx* ctx, const uint8_t* edges, size_t W, size_t H, size_t S, float rho, float thetaDeg, int threshold, int maxLines,
                         compvhip_line* lines, size_t cap, size_t* n, int32_t* acc, size_t accStride)
{
	if (!ctx) return COMPVHIP_E_INVALID_PARAMETER;
	if (!edges || !n || (cap && !lines) || S < W || !W || !H) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null/invalid argument"); // houghsht.cxx:98
	if (rho != 1.f) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "SHT requires rho == 1 (use KHT for fractional rho)"); // :306-316
	if (!(thetaDeg > 0.f) || threshold <= 0) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "theta and threshold must be > 0");
	if (W < 3 || H < 3 || W > 32767 || H > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "image size out of range (3..32767)");
	*n = 0;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	compvhip_plan* p = nullptr;
	int rc = hostPlan(ctx, W, H, thetaDeg, &p);
	if (rc) return rc;
	rc = ensureSht(p);
	if (rc) return rc;
	if (acc && accStride < p->T) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "accStride < theta bins");
	HIPCHK(ctx, hipMemcpy2DAsync(ctx->dIn, p->S, edges, S, W, H, hipMemcpyHostToDevice, ctx->stream));
	if (!ctx->dCounts) HIPCHK(ctx, dmalloc(ctx, &ctx->dCounts, 1));
	// ALL candidate lines of the frame come back as (strength, cell) pairs in emission order -- the sort and the decode kernel are skipped:
	// the order the reference returns them in, and which equal-strength lines survive maxLines, is decided by its unstable std::sort
	int32_t count = 0;
	for (int attempt = 0; attempt < 2; ++attempt) {
		rc = planShtImpl(p, ctx->dIn, threshold, 0, nullptr, 0, ctx->dCounts, ctx->stream, true, true);
		if (rc) return rc;
		HIPCHK(ctx, hipMemcpyAsync(&count, ctx->dCounts, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
		if (static_cast<size_t>(count) <= p->lineCap) break;
		// more candidate lines than the device key buffer holds: grow it and redo the line stage
		rc = ensureLineCap(p, static_cast<size_t>(count));
		if (rc) return rc;
	}
	std::vector<uint32_t> hk(static_cast<size_t>(count)), hv(static_cast<size_t>(count));
	if (count) {
		HIPCHK(ctx, hipMemcpyAsync(hk.data(), p->keysA, hk.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
		HIPCHK(ctx, hipMemcpyAsync(hv.data(), p->valsA, hv.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	}
	std::vector<compvhip_line> all;
	referenceLineOrder(hk, hv, (1u << p->strengthBits) - 1u, p->T, static_cast<long long>(p->W + p->H), p->thetaStep, all);
	size_t found = all.size();
	if (maxLines > 0 && found > static_cast<size_t>(maxLines)) found = static_cast<size_t>(maxLines);
	*n = found;
	const size_t ncopy = std::min(found, cap);
	if (ncopy) memcpy(lines, all.data(), ncopy * sizeof(compvhip_line));
	if (acc) {
		const size_t elems = p->R * p->T;
		if (ctx->dAccOutElems < elems) { dfree(ctx, ctx->dAccOut); HIPCHK(ctx, dmalloc(ctx, &ctx->dAccOut, elems)); ctx->dAccOutElems = elems; }
		HIPCHK(ctx, launch_sht_acc_transpose(p->acc, static_cast<int>(p->R), static_cast<int>(p->T), p->accPitch, ctx->dAccOut, p->T, ctx->stream));
		HIPCHK(ctx, hipMemcpy2DAsync(acc, accStride * sizeof(int32_t), ctx->dAccOut, p->T * sizeof(int32_t), p->T * sizeof(int32_t), p->R,
		                             hipMemcpyDeviceToHost, ctx->stream));
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	}
	if (found > cap) return fail(ctx, COMPVHIP_E_OUT_OF_BOUND, "line buffer too small");
	return COMPVHIP_OK;
}

// ---- KHT -----------------------------------------------------------------------------------------------------------------------
// All of it works on ONE KhtScratch (its stream, its device buffers) and reports failures through K.err: the batched entry point runs
// several of these at the same time on worker threads, so nothing below touches ctx->err or any other shared state (ctx->live is atomic).
#define KCHK(K, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (K).err = std::string(#call) + ": " + hipGetErrorString(e__); return COMPVHIP_E_HIP; } } while (0)

// the reference's AVX (4) / SSE2 (2) kernel-height loops take n & ~(pack - 1) clusters of a frame; the rest go through the C code (other operation order)
static int khtSimdEnd(size_t n)
{
	const size_t pack = n >= 4 ? 4 : (n >= 2 ? 2 : 1);
	return static_cast<int>(pack > 1 ? (n & ~(pack - 1)) : 0);
}

// host linking (on K.plane, which it destroys), then cluster subdivision (kht_subdivide_kernel) and per-cluster statistics (kht_stats_kernel) on the GPU; kernels in cluster order
static int khtBuildKernels(compvhip_ctx* ctx, KhtScratch& K, size_t W, size_t H, double clusterMinDeviation, size_t clusterMinSize,
                           std::vector<KhtKernel>& kernels, double& hmax)
{
	using clk = std::chrono::steady_clock;
	auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	kernels.clear(); hmax = 0.0;
	const auto t0 = clk::now();
	std::vector<KhtRange> strings;
	const size_t most = khtPlaneCount(K.plane);
	if (most > 0x7fffffffull) { K.err = "too many edge pixels"; return COMPVHIP_E_INVALID_PARAMETER; }
	if (K.linkedCap < most) {
		KCHK(K, hipSetDevice(ctx->device));
		if (K.linked) (void)hipHostFree(K.linked);
		K.linked = nullptr; K.linkedCap = 0;
		const size_t want = most + most / 4 + 4096;   // (frames of a stream resemble each other: no reallocation for a slightly denser one)
		KCHK(K, hipHostMalloc(reinterpret_cast<void**>(&K.linked), want * sizeof(KhtPoint)));
		K.linkedCap = want;
	}
	const size_t nPts = khtLink(K.plane, clusterMinSize, K.linked, strings);
	const auto t1 = clk::now();
	K.stageMs[0] += ms(t0, t1);
	if (strings.empty()) return COMPVHIP_OK;

	// device: cluster subdivision (one wave per string), per-cluster statistics (one thread per cluster)
	std::vector<KhtStringDesc> descs(strings.size());
	size_t slots = 0;
	for (size_t i = 0; i < strings.size(); ++i) {
		descs[i].begin = static_cast<uint32_t>(strings[i].begin); descs[i].end = static_cast<uint32_t>(strings[i].end);
		descs[i].slot = static_cast<uint32_t>(slots);
		slots += khtSubdivSlots(strings[i].end - strings[i].begin, clusterMinSize);
	}
	KCHK(K, hipSetDevice(ctx->device));
	if (K.ptsCap < nPts) { dfree(ctx, K.pts); K.ptsCap = 0; KCHK(K, dmalloc(ctx, &K.pts, K.linkedCap)); K.ptsCap = K.linkedCap; }
	if (K.stringsCap < descs.size()) {
		dfree(ctx, K.strings); dfree(ctx, K.counts32); K.stringsCap = 0;
		KCHK(K, dmalloc(ctx, &K.strings, descs.size())); KCHK(K, dmalloc(ctx, &K.counts32, descs.size() + 2)); K.stringsCap = descs.size();
	}
	if (K.spansCap < slots) {
		dfree(ctx, K.spans); dfree(ctx, K.scratch); dfree(ctx, K.stack); dfree(ctx, K.kernelsDev); K.spansCap = 0;
		KCHK(K, dmalloc(ctx, &K.spans, slots)); KCHK(K, dmalloc(ctx, &K.scratch, slots)); KCHK(K, dmalloc(ctx, &K.stack, slots));
		KCHK(K, dmalloc(ctx, &K.kernelsDev, slots)); K.spansCap = slots;
	}
	hipStream_t st = K.stream;
	KCHK(K, hipMemcpyAsync(K.pts, K.linked, nPts * sizeof(KhtPoint), hipMemcpyHostToDevice, st));
	KCHK(K, hipMemcpyAsync(K.strings, descs.data(), descs.size() * sizeof(KhtStringDesc), hipMemcpyHostToDevice, st));
	KhtSubdivArgs sv;
	sv.pts = K.pts; sv.strings = K.strings; sv.nStrings = static_cast<int>(descs.size());
	sv.minSize = static_cast<int>(std::min<size_t>(clusterMinSize, 0x7fffffff)); sv.minDev = clusterMinDeviation;
	sv.scratch = K.scratch; sv.stack = K.stack; sv.counts = K.counts32; sv.clusters = K.spans; sv.total = K.counts32 + descs.size(); sv.flagIndex = 1;
	KCHK(K, hipMemsetAsync(sv.total, 0, 2 * sizeof(uint32_t), st));   // [0] cluster total, [1] "recursion truncated" flag
	KhtBatchStrings one{};   // a batch of one frame
	one.frames = 1; one.stringBegin[0] = 0; one.stringBegin[1] = static_cast<uint32_t>(descs.size()); one.clusterBase[0] = 0;
	KCHK(K, launch_kht_subdivide(sv, one, st));
	uint32_t tot[2] = { 0, 0 };
	KCHK(K, hipMemcpyAsync(tot, sv.total, sizeof(tot), hipMemcpyDeviceToHost, st));
	KCHK(K, hipStreamSynchronize(st));
	if (tot[1]) { K.err = "cluster subdivision ran out of recursion slots"; return COMPVHIP_E_INVALID_STATE; } // cannot happen: clusterMinSize >= 2 is enforced and khtSubdivSlots bounds the depth for it
	const uint32_t nClusters = tot[0];
	const auto t2 = clk::now();
	K.stageMs[1] += ms(t1, t2);
	if (!nClusters) return COMPVHIP_OK;
	const size_t n = nClusters;
	KhtStatsArgs sa;
	sa.pts = K.pts; sa.clusters = K.spans;
	sa.hw = static_cast<double>(W) * 0.5; sa.hh = static_cast<double>(H) * 0.5;
	sa.out = K.kernelsDev;
	KhtBatchStats ones{};
	ones.frames = 1; ones.clusterBase[0] = 0; ones.n[0] = static_cast<int>(n); ones.simdEnd[0] = khtSimdEnd(n);
	KCHK(K, launch_kht_stats(sa, ones, st));
	kernels.resize(n);
	KCHK(K, hipMemcpyAsync(kernels.data(), K.kernelsDev, n * sizeof(KhtKernel), hipMemcpyDeviceToHost, st));
	KCHK(K, hipStreamSynchronize(st));
	khtFinishKernels(kernels, hmax);
	K.stageMs[2] += ms(t2, clk::now());
	return COMPVHIP_OK;
}

// one frame, host edge map -> lines in the reference's order (the body of CompVHoughKht::process, houghkht.cxx:208-447)
// (the frame's edge map is K.plane: packed by the caller, destroyed by the linker)
static int khtFrame(compvhip_ctx* ctx, KhtScratch& K, size_t W, size_t H, const KhtAxes& ax, int threshold, int maxLines,
                    double clusterMinDeviation, size_t clusterMinSize, double kernelMinHeight, std::vector<KhtLine>& out, double* gs)
{
	using clk = std::chrono::steady_clock;
	auto msSince = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
	out.clear();
	std::vector<KhtKernel> kernels;
	double hmax = 0.0;
	const int rck = khtBuildKernels(ctx, K, W, H, clusterMinDeviation, clusterMinSize, kernels, hmax);
	if (rck) return rck;
	if (kernels.empty()) return COMPVHIP_OK;
	auto t3 = clk::now();
	const double GS = khtPruneAndScale(kernels, hmax, kernelMinHeight);
	if (kernels.empty()) return COMPVHIP_OK;
	if (gs) *gs = GS;
	std::vector<KhtVoteParams> params;
	khtVoteParams(ax, kernels, params);

	// device: Gaussian voting + smoothing/threshold
	KCHK(K, hipSetDevice(ctx->device));
	const int stride = static_cast<int>(alignUp(ax.rhoN + 2, 16));
	const size_t countsElems = (ax.T + 2) * static_cast<size_t>(stride);
	if (K.countsElems < countsElems) { dfree(ctx, K.counts); K.countsElems = 0; KCHK(K, dmalloc(ctx, &K.counts, countsElems)); K.countsElems = countsElems; }
	if (K.paramsCap < params.size()) { dfree(ctx, K.params); K.paramsCap = 0; KCHK(K, dmalloc(ctx, &K.params, params.size())); K.paramsCap = params.size(); }
	const size_t cellCap = ax.T * ax.rhoN;
	if (K.cellsCap < cellCap) { dfree(ctx, K.cells); K.cellsCap = 0; KCHK(K, dmalloc(ctx, &K.cells, cellCap)); K.cellsCap = cellCap; }
	if (!K.cellCount) KCHK(K, dmalloc(ctx, &K.cellCount, 1));
	hipStream_t st = K.stream;
	K.stageMs[3] += msSince(t3);
	t3 = clk::now();
	KCHK(K, hipMemsetAsync(K.counts, 0, countsElems * sizeof(int32_t), st));
	KCHK(K, hipMemsetAsync(K.cellCount, 0, sizeof(int), st));
	KCHK(K, hipMemcpyAsync(K.params, params.data(), params.size() * sizeof(KhtVoteParams), hipMemcpyHostToDevice, st));
	KhtGpuArgs a;
	a.params = K.params; a.nKernels = static_cast<int>(params.size()); a.counts = K.counts; a.stride = stride;
	a.rhoN = static_cast<int>(ax.rhoN); a.T = static_cast<int>(ax.T); a.dRho = ax.dRho; a.dThetaDeg = ax.dThetaDeg; a.gs = GS;
	a.threshold = threshold; a.cells = K.cells; a.cellCount = K.cellCount; a.cellCap = static_cast<int>(cellCap);
	KhtBatchVote onev{};
	onev.frames = 1; onev.paramsBase[0] = 0; onev.nKernels[0] = a.nKernels; onev.gs[0] = GS; onev.mapElems = countsElems; onev.cellCap = cellCap;
	KCHK(K, launch_kht_vote(a, onev, st));
	KCHK(K, launch_kht_peaks(a, onev, st));
	int cellCount = 0;
	KCHK(K, hipMemcpyAsync(&cellCount, K.cellCount, sizeof(int), hipMemcpyDeviceToHost, st));
	KCHK(K, hipStreamSynchronize(st));
	std::vector<KhtCell>& cells = K.cellsHost;
	cells.resize(static_cast<size_t>(std::min<int>(cellCount, static_cast<int>(cellCap))));
	if (!cells.empty()) {
		KCHK(K, hipMemcpyAsync(cells.data(), K.cells, cells.size() * sizeof(KhtCell), hipMemcpyDeviceToHost, st));
		KCHK(K, hipStreamSynchronize(st));
	}
	K.stageMs[4] += msSince(t3);
	t3 = clk::now();
	// host: sort + sweep (order dependent, :1195-1247)
	khtPeaks(ax, cells, maxLines, out, K.peaks);
	K.stageMs[5] += msSince(t3);
	return COMPVHIP_OK;
}

static int khtCheckParams(compvhip_ctx* ctx, size_t W, size_t H, float rho, float thetaDeg, int threshold, size_t clusterMinSize, double kernelMinHeight, KhtAxes& ax)
{
	if (!(rho > 0.f) || rho > 1.f || !(thetaDeg > 0.f) || threshold <= 0) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "rho in (0,1], theta > 0, threshold > 0"); // :146-163,491
	if (!clusterMinSize || !(kernelMinHeight >= 0.0)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "invalid KHT knob"); // :169-186 (the deviation is unchecked there)
	// Defined deviation: the reference's set() accepts a cluster size of 1 and its clusters_subdivision then recurses without bound on the first
	// collinear string (max_index stays at start_index, both "halves" hold >= 1 point: houghkht.cxx:795-821) -- a stack overflow, not a result.
	if (clusterMinSize < 2) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "clusterMinSize must be >= 2 (the reference's recursion does not terminate for 1)");
	if (!W || !H || W > 32767 || H > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "image size out of range");
	if (!khtAxes(W, H, rho, thetaDeg, ax)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "degenerate KHT parameter space");
	// the peak stage identifies a vote cell by the 32-bit key theta * 2 (rhoN + 2) + rho (KhtCell::order) and indexes the vote map with ints
	if (static_cast<uint64_t>(ax.T + 2) * 2u * (ax.rhoN + 2) >= (1ull << 32) || static_cast<uint64_t>(ax.T + 2) * (ax.rhoN + 2) > 0x7fffffffull)
		return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "KHT parameter space too fine for this image size ((T + 2) * 2 (rhoN + 2) must stay below 2^32)");
	return COMPVHIP_OK;
}

static void khtCopyLines(const std::vector<KhtLine>& out, compvhip_line* lines, size_t cap)
{
	const size_t ncopy = std::min(out.size(), cap);
	for (size_t i = 0; i < ncopy; ++i) {
		lines[i].rho = out[i].rho; lines[i].theta = out[i].theta; lines[i].strength = out[i].strength;
		lines[i].row = out[i].rhoIndex; lines[i].col = out[i].thetaIndex;
	}
}

int compvhip_houghkht_kernels_u8(compvhip_ctx* ctx, const uint8_t* edges, size_t W, size_t H, size_t S, double clusterMinDeviation, size_t clusterMinSize,
                                 double* kernels7, size_t cap, size_t* n, double* hmax)
{
	if (!ctx) return COMPVHIP_E_INVALID_PARAMETER;
	if (!edges || !n || (cap && !kernels7) || S < W || !W || !H || clusterMinSize < 2) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null/invalid argument (clusterMinSize >= 2)");
	if (W > 32767 || H > 32767) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "image size out of range");
	std::vector<KhtKernel> kernels; double hm = 0.0;
	ctx->kht.stream = ctx->stream;
	memset(ctx->kht.stageMs, 0, sizeof(ctx->kht.stageMs));
	{
		const auto tp = std::chrono::steady_clock::now();
		khtPackBytes(edges, W, H, S, ctx->kht.plane);
		ctx->kht.stageMs[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count();
	}
	const int rc = khtBuildKernels(ctx, ctx->kht, W, H, clusterMinDeviation, clusterMinSize, kernels, hm);
	if (rc) return fail(ctx, rc, ctx->kht.err.c_str());
	*n = kernels.size();
	if (hmax) *hmax = hm;
	for (size_t i = 0; i < std::min(kernels.size(), cap); ++i) {
		const KhtKernel& k = kernels[i];
		const double v[7] = { k.rho, k.theta, k.h, k.sigmaThetaSquare, k.sigmaRhoSquare, k.m2, k.sigmaRhoTimesTheta };
		memcpy(kernels7 + i * 7, v, sizeof(v));
	}
	if (kernels.size() > cap) return fail(ctx, COMPVHIP_E_OUT_OF_BOUND, "kernel buffer too small");
	return COMPVHIP_OK;
}

int compvhip_houghkht_link_u8(const uint8_t* edges, size_t W, size_t H, size_t S, size_t clusterMinSize, int32_t* xy, size_t cap, size_t* nPoints,
                              uint32_t* stringEnds, size_t stringCap, size_t* nStrings)
{
	if (!edges || !nPoints || !nStrings || (cap && !xy) || (stringCap && !stringEnds) || S < W || !W || !H || !clusterMinSize || W > 32767 || H > 32767)
		return COMPVHIP_E_INVALID_PARAMETER;
	try {
		KhtBitPlane plane;
		std::vector<KhtRange> strings;
		khtPackBytes(edges, W, H, S, plane);
		std::unique_ptr<KhtPoint[]> pts(new KhtPoint[khtPlaneCount(plane) + 1]);
		const size_t n = khtLink(plane, clusterMinSize, pts.get(), strings);
		*nPoints = n; *nStrings = strings.size();
		if (n > cap || strings.size() > stringCap) return COMPVHIP_E_OUT_OF_BOUND;
		for (size_t i = 0; i < n; ++i) { xy[2 * i] = pts[i].x; xy[2 * i + 1] = pts[i].y; }
		for (size_t i = 0; i < strings.size(); ++i) stringEnds[i] = static_cast<uint32_t>(strings[i].end);
	}
	catch (...) { return COMPVHIP_E_OUT_OF_MEMORY; }
	return COMPVHIP_OK;
}

int compvhip_houghkht_stage_ms(compvhip_ctx* ctx, double* ms6)
{
	if (!ctx || !ms6) return COMPVHIP_E_INVALID_PARAMETER;
	memcpy(ms6, ctx->kht.stageMs, sizeof(ctx->kht.stageMs));
	return COMPVHIP_OK;
}

int compvhip_houghkht_u8(compvhip_ctx* ctx, const uint8_t* edges, size_t W, size_t H, size_t S, float rho, float thetaDeg, int threshold, int maxLines,
                         double clusterMinDeviation, size_t clusterMinSize, double kernelMinHeight, compvhip_line* lines, size_t cap, size_t* n, double* gs)
{
	if (!ctx) return COMPVHIP_E_INVALID_PARAMETER;
	if (!edges || !n || (cap && !lines) || S < W || !W || !H) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null/invalid argument"); // houghkht.cxx:210-211
	KhtAxes ax;
	int rc = khtCheckParams(ctx, W, H, rho, thetaDeg, threshold, clusterMinSize, kernelMinHeight, ax);
	if (rc) return rc;
	*n = 0;
	ctx->kht.stream = ctx->stream;
	memset(ctx->kht.stageMs, 0, sizeof(ctx->kht.stageMs));
	std::vector<KhtLine> out;
	{
		const auto tp = std::chrono::steady_clock::now();
		khtPackBytes(edges, W, H, S, ctx->kht.plane);   // host bytes -> the linker's bit plane (the linker never touches the caller's map)
		ctx->kht.stageMs[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count();
	}
	rc = khtFrame(ctx, ctx->kht, W, H, ax, threshold, maxLines, clusterMinDeviation, clusterMinSize, kernelMinHeight, out, gs);
	if (rc) return fail(ctx, rc, ctx->kht.err.c_str());
	*n = out.size();
	khtCopyLines(out, lines, cap);
	if (out.size() > cap) return fail(ctx, COMPVHIP_E_OUT_OF_BOUND, "line buffer too small");
	return COMPVHIP_OK;
}

// CompVHoughKht::process on the plan's `frames` device edge maps.  The chain walk of the linker (Appendix A) is sequential per frame and
// stays on the host -- but frames are independent: a pool of host threads takes them in turn, each with its own HIP stream and scratch
// buffers.  A worker downloads its frame (pinned buffer, asynchronous copy on its stream), links it, and drives the GPU stages of that
// frame (subdivision, statistics, voting, peaks); while one worker links, the kernels and copies of the others run, so the GPU work
// and the PCIe transfers of the batch hide under the host stage that bounds it.
// ---- batched KHT (compvhip_plan_houghkht) ------------------------------------------------------------------------------------------------------
// The frames of a batch go through the stages TOGETHER: the host stages (bit-plane linking, prune / Gmin, sort + sweep: sequential per frame, independent
// between frames) run as parallel loops over the frames on a pool of host threads, the GPU stages are ONE launch each over the strings / clusters /
// kernels / vote maps of all frames (kht.hpp: the per-frame tables travel in the kernel arguments), with one upload and one download per stage.
// (Rounds 3-4 gave every worker thread its own stream and let it drive its frame's five small launches and four synchronisations: with 32 workers the
// GPU-touching stages took 5-9 x their single-frame time -- a launch / synchronisation pile-up, not compute.)
// one group of up to kKhtBatch frames
static int khtBatchGroup(compvhip_plan* p, KhtBatchState& B, KhtPool& pool, const uint8_t* d_edges, size_t G, const KhtAxes& ax, int threshold, int maxLines,
                         double clusterMinDeviation, size_t clusterMinSize, double kernelMinHeight, compvhip_line* lines, size_t cap, size_t* counts, double* gs, bool* overflow,
                         std::string& err)
{
	using clk = std::chrono::steady_clock;
	auto msSince = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
	compvhip_ctx* ctx = p->ctx;
	const size_t W = p->W, H = p->H, S = p->S;
	const size_t wpr = (W + 31) / 32, words = wpr * H;
	hipStream_t st = B.stream;
	auto sleepSync = [&]() -> hipError_t {   // the stream's work so far, waited for without spinning
		hipError_t e = hipEventRecord(B.syncEv, st);
		return e != hipSuccess ? e : hipEventSynchronize(B.syncEv);
	};
	// (several groups run at the same time, each on its own controller thread: errors travel back as (code, text), only the caller touches ctx->err)
// (an early return must not leave asynchronous copies in flight towards this frame's stack arrays or the pinned state: drain the stream first, result ignored)
#define BCHK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e__); (void)hipStreamSynchronize(st); return COMPVHIP_E_HIP; } } while (0)
	auto firstError = [&]() -> int {
		for (size_t f = 0; f < G; ++f)
			if (B.frames[f].code) { err = "frame " + std::to_string(f) + " of its group: " + B.frames[f].err; (void)hipStreamSynchronize(st); return B.frames[f].code; }
		return COMPVHIP_OK;
	};
	if (hipSetDevice(ctx->device) != hipSuccess) { err = "hipSetDevice"; return COMPVHIP_E_HIP; }
	auto guarded = [&](size_t f, const std::function<void(KhtBatchFrame&)>& body) {   // nothing may leave a pool thread (or an extern "C" entry point) as an exception
		KhtBatchFrame& fr = B.frames[f];
		if (fr.code) return;
		try { body(fr); }
		catch (const std::exception& ex) { fr.code = COMPVHIP_E_OUT_OF_MEMORY; fr.err = std::string("exception in a KHT stage: ") + ex.what(); }
		catch (...) { fr.code = COMPVHIP_E_OUT_OF_MEMORY; fr.err = "exception in a KHT stage"; }
	};
	for (size_t f = 0; f < G; ++f) {
		KhtBatchFrame& fr = B.frames[f];
		fr.code = COMPVHIP_OK; fr.err.clear(); fr.nClusters = 0; fr.kernels.clear(); fr.params.clear(); fr.cells.clear(); fr.cellCount = 0; fr.out.clear(); fr.haveGS = false;
		memset(fr.ms, 0, sizeof(fr.ms));
	}

	// ---- A. the edge maps leave the device as bit-mask rows (1/8 of the bytes over PCIe; the linker works on bits anyway): one kernel, one copy per frame ----
	BCHK(launch_bytes_to_bits(d_edges, static_cast<int>(W), static_cast<int>(H), static_cast<int>(S), S * H, B.dBits, static_cast<int>(wpr), words, static_cast<int>(G), st));
	for (size_t f = 0; f < G; ++f) {
		BCHK(hipMemcpyAsync(B.hostBits + f * words, B.dBits + f * words, words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		BCHK(hipEventRecord(B.ready[f], st));
	}
	pool.run(G, [&](size_t f) { guarded(f, [&](KhtBatchFrame& fr) {
		const auto t0 = clk::now();
		if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(B.ready[f]) != hipSuccess) { fr.code = COMPVHIP_E_HIP; fr.err = "frame download"; return; }
		khtPlaneFromWords(B.hostBits + f * words, wpr, W, H, fr.plane);
		fr.most = khtPlaneCount(fr.plane);
		fr.ms[0] += msSince(t0);
	}); }, 'P');
	int rc = firstError();
	if (rc) return rc;
	size_t total = 0;
	for (size_t f = 0; f < G; ++f) { B.frames[f].ptsOff = total; total += B.frames[f].most; }
	if (total > 0x7fffffffull) { err = "too many edge pixels in the batch"; return COMPVHIP_E_INVALID_PARAMETER; }
	BCHK(growPinned(B.linked, B.linkedCap, total + 1));
	// ---- B. linking (Appendix A): sequential inside a frame, the frames in parallel; every frame's points go straight into its slice of the pinned arena ----
	// (longest first: the items are milliseconds long and few, the last ones decide when the group moves on)
	std::vector<size_t> byWork(G);
	for (size_t f = 0; f < G; ++f) byWork[f] = f;
	std::sort(byWork.begin(), byWork.end(), [&](size_t a, size_t b) { return B.frames[a].most > B.frames[b].most; });
	pool.run(G, [&](size_t i) { const size_t f = byWork[i]; guarded(f, [&](KhtBatchFrame& fr) {
		const auto t0 = clk::now();
		fr.nPts = khtLink(fr.plane, clusterMinSize, B.linked + fr.ptsOff, fr.strings);
		fr.ms[0] += msSince(t0);
	}); }, 'L');
	rc = firstError();
	if (rc) return rc;

	// ---- C. cluster subdivision of ALL strings: one upload per frame slice, one launch, one download ----
	auto tc = clk::now();
	size_t nStrings = 0, slots = 0;
	for (size_t f = 0; f < G; ++f) nStrings += B.frames[f].strings.size();
	KhtBatchStrings tabS{};
	tabS.frames = static_cast<int>(G);
	if (nStrings) {
		BCHK(growPinned(B.stringsHost, B.stringsHostCap, nStrings));
		size_t si = 0;
		for (size_t f = 0; f < G; ++f) {
			KhtBatchFrame& fr = B.frames[f];
			tabS.stringBegin[f] = static_cast<uint32_t>(si); tabS.clusterBase[f] = static_cast<uint32_t>(slots);
			fr.slotBase = slots;
			for (const KhtRange& r : fr.strings) {
				KhtStringDesc& d = B.stringsHost[si++];
				d.begin = static_cast<uint32_t>(fr.ptsOff + r.begin); d.end = static_cast<uint32_t>(fr.ptsOff + r.end); d.slot = static_cast<uint32_t>(slots);
				slots += khtSubdivSlots(r.end - r.begin, clusterMinSize);
			}
			fr.slots = slots - fr.slotBase;
		}
		for (size_t f = G; f <= static_cast<size_t>(kKhtBatch); ++f) tabS.stringBegin[f] = static_cast<uint32_t>(nStrings);
		tabS.stringBegin[G] = static_cast<uint32_t>(nStrings);
		if (slots > 0xffffffffull) { err = "too many cluster slots in the batch"; return COMPVHIP_E_INVALID_PARAMETER; }
		BCHK(growDevice(ctx, B.pts, B.ptsCap, total + 1));
		if (B.stringsCap < nStrings) {
			dfree(ctx, B.strings); dfree(ctx, B.counts32); B.stringsCap = 0;
			const size_t n = nStrings + nStrings / 4 + 1024;
			BCHK(dmalloc(ctx, &B.strings, n)); BCHK(dmalloc(ctx, &B.counts32, n)); B.stringsCap = n;
		}
		if (B.spansCap < slots) {
			dfree(ctx, B.spans); dfree(ctx, B.scratch); dfree(ctx, B.stack); dfree(ctx, B.kernelsDev); B.spansCap = 0;
			const size_t n = slots + slots / 4 + 1024;
			BCHK(dmalloc(ctx, &B.spans, n)); BCHK(dmalloc(ctx, &B.scratch, n)); BCHK(dmalloc(ctx, &B.stack, n)); BCHK(dmalloc(ctx, &B.kernelsDev, n)); B.spansCap = n;
		}
		BCHK(growPinned(B.kernelsHost, B.kernelsHostCap, slots));
		for (size_t f = 0; f < G; ++f) {
			const KhtBatchFrame& fr = B.frames[f];
			if (fr.nPts) BCHK(hipMemcpyAsync(B.pts + fr.ptsOff, B.linked + fr.ptsOff, fr.nPts * sizeof(KhtPoint), hipMemcpyHostToDevice, st));
		}
		BCHK(hipMemcpyAsync(B.strings, B.stringsHost, nStrings * sizeof(KhtStringDesc), hipMemcpyHostToDevice, st));
		KhtSubdivArgs sv;
		sv.pts = B.pts; sv.strings = B.strings; sv.nStrings = static_cast<int>(nStrings);
		sv.minSize = static_cast<int>(std::min<size_t>(clusterMinSize, 0x7fffffff)); sv.minDev = clusterMinDeviation;
		sv.scratch = B.scratch; sv.stack = B.stack; sv.counts = B.counts32; sv.clusters = B.spans; sv.total = B.totals; sv.flagIndex = kKhtBatch;
		BCHK(hipMemsetAsync(B.totals, 0, (kKhtBatch + 1) * sizeof(uint32_t), st));
		BCHK(launch_kht_subdivide(sv, tabS, st));
		uint32_t tot[kKhtBatch + 1];
		BCHK(hipMemcpyAsync(tot, B.totals, sizeof(tot), hipMemcpyDeviceToHost, st));
		BCHK(sleepSync());
		if (tot[kKhtBatch]) { err = "cluster subdivision ran out of recursion slots"; return COMPVHIP_E_INVALID_STATE; }   // cannot happen: clusterMinSize >= 2 is enforced and khtSubdivSlots bounds the depth for it
		for (size_t f = 0; f < G; ++f) B.frames[f].nClusters = B.frames[f].strings.empty() ? 0u : tot[f];
	}
	B.stageMs[1] += msSince(tc);

	// ---- D. per-cluster statistics of ALL clusters: one launch, one download per frame slice; acos / hmax, prune, Gmin and the vote parameters on the pool ----
	tc = clk::now();
	{
		KhtBatchStats tab{};
		tab.frames = static_cast<int>(G);
		bool any = false;
		for (size_t f = 0; f < G; ++f) {
			const KhtBatchFrame& fr = B.frames[f];
			tab.clusterBase[f] = static_cast<uint32_t>(fr.slotBase); tab.n[f] = static_cast<int>(fr.nClusters); tab.simdEnd[f] = khtSimdEnd(fr.nClusters);
			any = any || fr.nClusters;
		}
		if (any) {
			KhtStatsArgs sa;
			sa.pts = B.pts; sa.clusters = B.spans; sa.hw = static_cast<double>(W) * 0.5; sa.hh = static_cast<double>(H) * 0.5; sa.out = B.kernelsDev;
			BCHK(launch_kht_stats(sa, tab, st));
			for (size_t f = 0; f < G; ++f) {
				const KhtBatchFrame& fr = B.frames[f];
				if (fr.nClusters) BCHK(hipMemcpyAsync(B.kernelsHost + fr.slotBase, B.kernelsDev + fr.slotBase, fr.nClusters * sizeof(KhtKernel), hipMemcpyDeviceToHost, st));
			}
			BCHK(sleepSync());
		}
	}
	B.stageMs[2] += msSince(tc);
	pool.run(G, [&](size_t f) { guarded(f, [&](KhtBatchFrame& fr) {
		if (!fr.nClusters) return;
		auto t0 = clk::now();
		fr.kernels.assign(B.kernelsHost + fr.slotBase, B.kernelsHost + fr.slotBase + fr.nClusters);
		khtFinishKernels(fr.kernels, fr.hmax);
		fr.ms[2] += msSince(t0);
		t0 = clk::now();
		fr.GS = khtPruneAndScale(fr.kernels, fr.hmax, kernelMinHeight);
		if (!fr.kernels.empty()) { fr.haveGS = true; khtVoteParams(ax, fr.kernels, fr.params); }
		fr.ms[3] += msSince(t0);
	}); }, 'K');
	rc = firstError();
	if (rc) return rc;

	// ---- E. Gaussian voting + smoothing / threshold of ALL frames' vote maps: one launch each, the cell counts, then the cells ----
	tc = clk::now();
	const int stride = static_cast<int>(alignUp(ax.rhoN + 2, 16));
	const size_t mapElems = (ax.T + 2) * static_cast<size_t>(stride), cellCap = ax.T * ax.rhoN;
	KhtBatchVote tabV{};
	tabV.frames = static_cast<int>(G); tabV.mapElems = mapElems; tabV.cellCap = cellCap;
	size_t nParams = 0;
	for (size_t f = 0; f < G; ++f) {
		KhtBatchFrame& fr = B.frames[f];
		fr.paramsBase = nParams; nParams += fr.params.size();
		tabV.paramsBase[f] = static_cast<uint32_t>(fr.paramsBase); tabV.nKernels[f] = static_cast<int>(fr.params.size()); tabV.gs[f] = fr.GS;
	}
	if (nParams) {
		if (B.countsElems < mapElems * G) { dfree(ctx, B.counts); B.countsElems = 0; BCHK(dmalloc(ctx, &B.counts, mapElems * G)); B.countsElems = mapElems * G; }
		if (B.cellsCap < cellCap * G) { dfree(ctx, B.cells); B.cellsCap = 0; BCHK(dmalloc(ctx, &B.cells, cellCap * G)); B.cellsCap = cellCap * G; }
		BCHK(growPinned(B.paramsHost, B.paramsHostCap, nParams));
		BCHK(growDevice(ctx, B.params, B.paramsCap, nParams));
		for (size_t f = 0; f < G; ++f) { const KhtBatchFrame& fr = B.frames[f]; if (!fr.params.empty()) memcpy(B.paramsHost + fr.paramsBase, fr.params.data(), fr.params.size() * sizeof(KhtVoteParams)); }
		BCHK(hipMemcpyAsync(B.params, B.paramsHost, nParams * sizeof(KhtVoteParams), hipMemcpyHostToDevice, st));
		BCHK(hipMemsetAsync(B.counts, 0, mapElems * G * sizeof(int32_t), st));
		BCHK(hipMemsetAsync(B.cellCount, 0, kKhtBatch * sizeof(int), st));
		KhtGpuArgs a;
		a.params = B.params; a.nKernels = 0; a.counts = B.counts; a.stride = stride;
		a.rhoN = static_cast<int>(ax.rhoN); a.T = static_cast<int>(ax.T); a.dRho = ax.dRho; a.dThetaDeg = ax.dThetaDeg; a.gs = 1.0;
		a.threshold = threshold; a.cells = B.cells; a.cellCount = B.cellCount; a.cellCap = static_cast<int>(cellCap);
		BCHK(launch_kht_vote(a, tabV, st));
		BCHK(launch_kht_peaks(a, tabV, st));
		int cc[kKhtBatch];
		BCHK(hipMemcpyAsync(cc, B.cellCount, sizeof(cc), hipMemcpyDeviceToHost, st));
		BCHK(sleepSync());
		size_t nCells = 0;
		for (size_t f = 0; f < G; ++f) {
			KhtBatchFrame& fr = B.frames[f];
			fr.cellCount = fr.params.empty() ? 0 : std::min<int>(cc[f], static_cast<int>(cellCap));
			fr.cellOff = nCells; nCells += static_cast<size_t>(fr.cellCount);
		}
		if (nCells) {
			BCHK(growPinned(B.cellsHost, B.cellsHostCap, nCells));
			for (size_t f = 0; f < G; ++f) {
				const KhtBatchFrame& fr = B.frames[f];
				if (fr.cellCount) BCHK(hipMemcpyAsync(B.cellsHost + fr.cellOff, B.cells + f * cellCap, static_cast<size_t>(fr.cellCount) * sizeof(KhtCell), hipMemcpyDeviceToHost, st));
			}
			BCHK(sleepSync());
		}
	}
	B.stageMs[4] += msSince(tc);

	// ---- F. sort + sweep with the visited map (order dependent, :1195-1247): per frame, on the pool ----
	for (size_t f = 0; f < G; ++f) byWork[f] = f;
	std::sort(byWork.begin(), byWork.end(), [&](size_t a, size_t b) { return B.frames[a].cellCount > B.frames[b].cellCount; });
	pool.run(G, [&](size_t i) { const size_t f = byWork[i]; guarded(f, [&](KhtBatchFrame& fr) {
		if (fr.haveGS && gs) gs[f] = fr.GS;
		if (!fr.cellCount) { counts[f] = 0; return; }
		const auto t0 = clk::now();
		fr.cells.assign(B.cellsHost + fr.cellOff, B.cellsHost + fr.cellOff + fr.cellCount);
		// the WORKER's workspace, not the frame's: 32 frames x 1.6 MB of visited maps cycled through the caches (sort + sweep 0.69 ms for a frame alone, 1.9 in a batch)
		const int w = t_khtWorker;
		KhtPeaksWork& wk = (w >= 0 && static_cast<size_t>(w) < p->khtWork.size() && p->khtWork[w]) ? *p->khtWork[w] : fr.peaks;
		khtPeaks(ax, fr.cells, maxLines, fr.out, wk);
		counts[f] = fr.out.size();
		if (lines) khtCopyLines(fr.out, lines + f * cap, cap);
		fr.ms[5] += msSince(t0);
	}); }, 'S');
	rc = firstError();
	if (rc) return rc;
	for (size_t f = 0; f < G; ++f) {
		const KhtBatchFrame& fr = B.frames[f];
		B.stageMs[0] += fr.ms[0]; B.stageMs[2] += fr.ms[2]; B.stageMs[3] += fr.ms[3]; B.stageMs[5] += fr.ms[5];
		if (fr.out.size() > cap) *overflow = true;
	}
#undef BCHK
	return COMPVHIP_OK;
}

int compvhip_plan_houghkht(compvhip_plan* p, const uint8_t* d_edges, float rho, float thetaDeg, int threshold, int maxLines, double clusterMinDeviation,
                           size_t clusterMinSize, double kernelMinHeight, compvhip_line* lines, size_t cap, size_t* counts, double* gs, int hostThreads)
{
	if (!p) return COMPVHIP_E_INVALID_PARAMETER;
	compvhip_ctx* ctx = p->ctx;
	if (!d_edges || !counts || (cap && !lines)) return fail(ctx, COMPVHIP_E_INVALID_PARAMETER, "null/invalid argument");
	const size_t W = p->W, H = p->H, S = p->S, F = p->frames;
	KhtAxes ax;
	int rc = khtCheckParams(ctx, W, H, rho, thetaDeg, threshold, clusterMinSize, kernelMinHeight, ax);
	if (rc) return rc;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	unsigned hw = std::thread::hardware_concurrency();
	if (!hw) hw = 4;
	// default: what the host really grants (affinity mask, cgroup quota), at most 32, at most half the hardware threads
	size_t T = hostThreads > 0 ? static_cast<size_t>(hostThreads) : std::min<size_t>(std::min<size_t>(32, hostCpuBudget()), std::max<size_t>(1, hw / 2));
	T = std::min(T, F);
	// The frames go through the stages in GROUPS of kKhtGroup, up to four groups at a time, each with its own controller thread, stream, buffers and share
	// of the host threads: inside a group the stages are batched (one launch, one transfer per stage), and while one group is in a GPU stage the host
	// threads of the others link or sweep.  (One group of 32 frames: every stage waits for the slowest frame and the GPU stages -- 100 MB over PCIe per
	// 4K batch -- wait for all of them: 18-20 ms per batch against 10.9 ms for the thread-per-frame pipeline of round 4; measured)
	size_t group = kKhtGroup;
	if (const char* e = getenv("COMPVHIP_KHT_GROUP")) { const long v = atol(e); if (v >= 1 && v <= kKhtBatch) group = static_cast<size_t>(v); }   // lab knob
	const size_t nGroups = (F + group - 1) / group;
	// controllers = groups in flight.  A controller only enqueues GPU work, sleeps on it and posts its group's host stages to the shared workers, so there are
	// enough of them to keep the workers fed while some groups are on the GPU: all groups of a 32-frame batch, two groups per four workers otherwise.
	const size_t K = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(std::max<size_t>(2, T / 2), kKhtMaxInFlight), nGroups));
	// The producer of d_edges may still be running on the caller's stream; the groups use private streams: drain the device first (the call is
	// synchronous and takes milliseconds -- the drain is not what bounds it)
	HIPCHK(ctx, hipDeviceSynchronize());
	const size_t G0 = std::min<size_t>(F, group), words = ((W + 31) / 32) * H;
	while (p->khtBatch.size() < K) {
		KhtBatchState* b = new (std::nothrow) KhtBatchState();
		if (!b) return fail(ctx, COMPVHIP_E_OUT_OF_MEMORY, "KHT batch state");
		p->khtBatch.push_back(b);
	}
	for (size_t k = 0; k < K; ++k) {
		KhtBatchState& B = *p->khtBatch[k];
		if (!B.stream) HIPCHK(ctx, hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking));
		if (!B.syncEv) HIPCHK(ctx, hipEventCreateWithFlags(&B.syncEv, hipEventBlockingSync | hipEventDisableTiming));
		if (B.bitsWords < words * G0) {
			dfree(ctx, B.dBits); if (B.hostBits) (void)hipHostFree(B.hostBits);
			B.hostBits = nullptr; B.bitsWords = 0;
			HIPCHK(ctx, dmalloc(ctx, &B.dBits, words * G0));
			if (hipHostMalloc(reinterpret_cast<void**>(&B.hostBits), words * G0 * sizeof(uint32_t)) != hipSuccess) return fail(ctx, COMPVHIP_E_OUT_OF_MEMORY, "pinned bit planes");
			B.bitsWords = words * G0;
		}
		while (B.ready.size() < G0) { hipEvent_t e; HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); B.ready.push_back(e); }
		if (!B.totals) HIPCHK(ctx, dmalloc(ctx, &B.totals, kKhtBatch + 1));
		if (!B.cellCount) HIPCHK(ctx, dmalloc(ctx, &B.cellCount, kKhtBatch));
		try { if (B.frames.size() < G0) B.frames.resize(G0); }
		catch (...) { return fail(ctx, COMPVHIP_E_OUT_OF_MEMORY, "KHT batch state"); }
		memset(B.stageMs, 0, sizeof(B.stageMs));
	}
	for (size_t f = 0; f < F; ++f) counts[f] = 0;
	try {   // nothing may leave an extern "C" entry point as an exception (the vectors below allocate)
	const auto wall0 = std::chrono::steady_clock::now();
	std::atomic<size_t> nextGroup{0};
	std::vector<int> codes(K, COMPVHIP_OK);
	std::vector<std::string> errs(K);
	std::vector<size_t> badGroup(K, 0);
	std::atomic<int> overflowAny{0};
	while (p->khtWork.size() < T) p->khtWork.emplace_back(new KhtPeaksWork());
	KhtPool pool(T);   // the workers of this call, shared by every group in flight
	auto controller = [&](size_t k) {
		try {
			for (;;) {
				const size_t g = nextGroup.fetch_add(1);
				if (g >= nGroups) break;
				const size_t g0 = g * group, G = std::min<size_t>(group, F - g0);
				bool overflow = false;
				const int r = khtBatchGroup(p, *p->khtBatch[k], pool, d_edges + g0 * S * H, G, ax, threshold, maxLines, clusterMinDeviation, clusterMinSize, kernelMinHeight,
				                            lines ? lines + g0 * cap : nullptr, cap, counts + g0, gs ? gs + g0 : nullptr, &overflow, errs[k]);
				if (overflow) overflowAny.store(1);
				if (r) { codes[k] = r; badGroup[k] = g0; nextGroup.store(nGroups); break; }   // the other controllers finish the group they are in and stop
			}
		}
		catch (const std::exception& ex) { codes[k] = COMPVHIP_E_OUT_OF_MEMORY; errs[k] = std::string("exception in the batched KHT: ") + ex.what(); }   // nothing may leave a thread
		catch (...) { codes[k] = COMPVHIP_E_OUT_OF_MEMORY; errs[k] = "exception in the batched KHT"; }                                                   // (or an extern "C" entry point) as an exception
	};
	{
		std::vector<std::thread> ctl;
		try { for (size_t k = 1; k < K; ++k) ctl.emplace_back(controller, k); }
		catch (...) { /* the system refused a thread: the controllers that did start take all the groups */ }
		controller(0);
		for (std::thread& t : ctl) t.join();
	}
	p->khtWallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
	p->khtThreads = static_cast<int>(T);
	memset(p->khtStageMs, 0, sizeof(p->khtStageMs));
	for (size_t k = 0; k < K; ++k) for (int i = 0; i < 6; ++i) p->khtStageMs[i] += p->khtBatch[k]->stageMs[i];
	for (size_t k = 0; k < K; ++k)
		if (codes[k]) return fail(ctx, codes[k], ("frames from " + std::to_string(badGroup[k]) + ": " + errs[k]).c_str());
	if (overflowAny.load()) return fail(ctx, COMPVHIP_E_OUT_OF_BOUND, "line buffer too small");
	}
	catch (const std::exception& ex) { return fail(ctx, COMPVHIP_E_OUT_OF_MEMORY, (std::string("exception in the batched KHT: ") + ex.what()).c_str()); }
	catch (...) { return fail(ctx, COMPVHIP_E_OUT_OF_MEMORY, "exception in the batched KHT"); }
	return COMPVHIP_OK;
}

int compvhip_plan_houghkht_stage_ms(compvhip_plan* p, double* ms6, double* wallMs, int* threads)
{
	if (!p || !ms6) return COMPVHIP_E_INVALID_PARAMETER;
	memcpy(ms6, p->khtStageMs, sizeof(p->khtStageMs));
	if (wallMs) *wallMs = p->khtWallMs;
	if (threads) *threads = p->khtThreads;
	return COMPVHIP_OK;
}

} // extern "C"
