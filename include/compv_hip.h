/* compv_hip.h -- C ABI of the MI355X (gfx950) implementation of CompV's Sobel -> Canny -> Hough hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ types, no ownership transfer.  It is what a
 * CompV maintainer binds from the replacement factories registered with CompVFeature::addFactory()
 * (reference: base/include/compv/base/compv_features.h:36-44 registry, base/compv_features.cxx:30-40 replace-by-id);
 * the reference-side binding is shown in INTEGRATION.md and implemented in integration/compv_hip_plugin.cxx.
 * Precedent for a function-pointer GPU hook in the reference: gpu/include/compv/gpu/base/math/compv_gpu_math_convlt.h.
 *
 * All file:line citations are relative to the CompV source tree.
 *
 * Conventions
 *   - return value: 0 = success (COMPV_ERROR_CODE_S_OK), negative = COMPVHIP_E_* (mapping to COMPV_ERROR_CODE in
 *     INTEGRATION.md).  No exceptions cross this boundary.
 *   - W,H = columns/rows, S = row stride in elements (bytes for u8).  Images are single-plane uint8.
 *   - "host" entry points take HOST pointers, are synchronous and leave results valid in host memory on return
 *     (what CompVEdgeDete::process / CompVHough::process promise their callers).
 *   - "dev" entry points take DEVICE pointers to frames resident in HBM and enqueue work on a HIP stream.
 *   - an instance (ctx / plan) is not re-entrant, exactly like a CompVEdgeDeteCanny / CompVHoughSht object
 *     (persistent scratch sized on first use: core/features/edges/compv_core_feature_canny_dete.cxx:133-147).
 */
#ifndef COMPV_HIP_H
#define COMPV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#	define COMPVHIP_API __attribute__((visibility("default")))
#else
#	define COMPVHIP_API
#endif

/* ---- error codes (negative) ------------------------------------------------------------------------------- */
enum {
	COMPVHIP_OK = 0,
	COMPVHIP_E_NOT_IMPLEMENTED = -1,    /* COMPV_ERROR_CODE_E_NOT_IMPLEMENTED */
	COMPVHIP_E_NOT_INITIALIZED = -2,    /* COMPV_ERROR_CODE_E_NOT_INITIALIZED */
	COMPVHIP_E_INVALID_STATE = -3,      /* COMPV_ERROR_CODE_E_INVALID_STATE   (tLow >= tHigh, canny_dete.cxx:126) */
	COMPVHIP_E_INVALID_PARAMETER = -4,  /* COMPV_ERROR_CODE_E_INVALID_PARAMETER */
	COMPVHIP_E_OUT_OF_MEMORY = -5,      /* COMPV_ERROR_CODE_E_OUT_OF_MEMORY */
	COMPVHIP_E_OUT_OF_BOUND = -6,       /* COMPV_ERROR_CODE_E_OUT_OF_BOUND    (caller's line buffer too small) */
	COMPVHIP_E_HIP = -7                 /* a hipError_t; class of COMPV_ERROR_CODE_E_CUDA / _E_OPENCL (compv_common.h:265-266) */
};

/* ---- operator / mode ids ---------------------------------------------------------------------------------- */
enum { /* gradient operator of the edge detector; ids mirror COMPV_SOBEL_ID / _SCHARR_ID / _PREWITT_ID
          (compv_features.h:84-90), kernels compv_features.h:124-133 */
	COMPVHIP_OP_SOBEL = 0,
	COMPVHIP_OP_SCHARR = 2,
	COMPVHIP_OP_PREWITT = 3
};
enum { /* COMPV_CANNY_THRESHOLD_TYPE_* (compv_features.h:80-81) */
	COMPVHIP_CANNY_THRESHOLD_COMPARE_TO_GRADIENT = 0,
	COMPVHIP_CANNY_THRESHOLD_PERCENT_OF_MEAN = 1,
	/* plan API only -- the sample's per-frame sequence fused on the device (samples/hough_lines/main.cxx:103-105):
	 * t = thresholdOtsu(frame); LOW = (float)(t * tLow); HIGH = (float)(t * tHigh); then COMPARE_TO_GRADIENT.
	 * The sample uses the factors tLow = 0.5, tHigh = 1.0. */
	COMPVHIP_CANNY_THRESHOLD_OTSU = 2
};
typedef enum compvhip_pixfmt { /* packed input formats of CompVImage::convertGrayscale (COMPV_SUBTYPE_PIXELS_*, compv_common.h:347-367) */
	COMPVHIP_FMT_RGBA32 = 0, COMPVHIP_FMT_ARGB32 = 1, COMPVHIP_FMT_BGRA32 = 2, COMPVHIP_FMT_RGB24 = 3, COMPVHIP_FMT_BGR24 = 4,
	COMPVHIP_FMT_RGB565LE = 5, COMPVHIP_FMT_RGB565BE = 6, COMPVHIP_FMT_BGR565LE = 7, COMPVHIP_FMT_BGR565BE = 8,
	COMPVHIP_FMT_YUYV422 = 9, COMPVHIP_FMT_UYVY422 = 10,
	COMPVHIP_FMT_Y = 11 /* the Y plane of Y / NV12 / NV21 / YUV420P / YVU420P / YUV422P / YUV444P: a copy */
} compvhip_pixfmt;

/* One Hough line.  rho/theta/strength are CompVHoughLine's fields (compv_common.h:686-693); row/col are the
 * accumulator cell (rho = barrier - row, theta = col * thetaStepRad) and define the canonical tie order of the plan API. */
typedef struct compvhip_line {
	float rho;
	float theta;
	int32_t strength;
	int32_t row;
	int32_t col;
} compvhip_line;

typedef struct compvhip_ctx compvhip_ctx;     /* one GPU + its scratch; one per host thread / CompV object */
typedef struct compvhip_plan compvhip_plan;   /* batched device-resident pipeline for a fixed geometry */

/* ---- life cycle -------------------------------------------------------------------------------------------- */
COMPVHIP_API int compvhip_device_count(void);
/* device < 0: use the current HIP device. Replaces nothing in CompV: it is what CompVGpu::init() would call
 * (gpu/compv_gpu.cxx:53-58). */
COMPVHIP_API int compvhip_ctx_create(compvhip_ctx** ctx, int device);
COMPVHIP_API void compvhip_ctx_destroy(compvhip_ctx* ctx);
/* text of the last failure on this ctx (HIP error string included); never NULL */
COMPVHIP_API const char* compvhip_last_error(const compvhip_ctx* ctx);
/* hipMalloc/hipFree balance of this ctx -- the analogue of COMPV_DEBUG_CHECK_FOR_MEMORY_LEAKS (compv_api.h:148-155) */
COMPVHIP_API long compvhip_live_allocations(const compvhip_ctx* ctx);

/* ---- host entry points (drop-in for process()) ------------------------------------------------------------- */

/* CompVCornerDeteEdgeBase::process (core/features/edges/compv_core_feature_edge_dete.cxx:55-206):
 * out = sat_u8(trunc(float(|gx|+|gy|) * (255.f / gmax))).  in and out may alias. */
COMPVHIP_API int compvhip_edge_dete_u8(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S, int op,
                                       uint8_t* out, size_t So);

/* CompVEdgeDeteCanny::process (core/features/edges/compv_core_feature_canny_dete.cxx:123-331).
 * tLow/tHigh are the detector's float thresholds (resolved to uint16 exactly as :251-266), ksize 3 or 5,
 * thresholdType COMPVHIP_CANNY_THRESHOLD_*.  out = {0,0xff}; in and out may alias (samples/edges_canny/main.cxx:72). */
COMPVHIP_API int compvhip_canny_u8(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S,
                                   float tLow, float tHigh, int ksize, int thresholdType, uint8_t* out, size_t So);

/* CompVImage::convertGrayscale (base/image/compv_image_conv_to_grayscale.cxx:35-90, called at samples/hough_lines/main.cxx:102):
 * packed RGB-family / YUV 4:2:2 frame -> luma plane, Y = ((33 R + 65 G + 13 B) >> 7) + 16 (compv_image_conv_rgbfamily.cxx:108).
 * in: H rows of S samples (S * bytes-per-sample bytes per row, S >= W as in CompVMat); out: H rows of W bytes at stride So. */
COMPVHIP_API int compvhip_grayscale_u8(compvhip_ctx* ctx, const uint8_t* in, int pixfmt, size_t W, size_t H, size_t S,
                                       uint8_t* out, size_t So);

/* CompVImage::thresholdOtsu (base/image/compv_image_threshold.cxx:52-114, called at samples/hough_lines/main.cxx:103):
 * 256-bin histogram of the W x H plane + the reference's f32 scan; *threshold = the integer level as a double. */
COMPVHIP_API int compvhip_otsu_u8(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S, double* threshold);

/* CompVMathGauss::kernelDim1FixedPoint (base/math/compv_math_gauss.cxx:11-17: kernelDim1<float> of compv_math_gauss.h:24-55, then
 * CompVMathConvlt::fixedPointKernel, compv_math_convlt.h:77-92): `size` (odd) Q16 weights of a normalised 1-D Gaussian.  Host
 * arithmetic only (libm exp/sqrt in the reference's float/double mix); no GPU involved. */
COMPVHIP_API int compvhip_gauss_kernel_fixedpoint(size_t size, float sigma, uint16_t* kernel);

/* CompVMathConvlt::convlt1FixedPoint (base/include/compv/base/math/compv_math_convlt.h:31-33,98-173,386-405): separable Q16
 * convolution u8 -> u8, horizontal pass with hzKern then vertical pass with vtKern through a u8 temporary, each
 * out = min(255, sum_k ((in[k] * kern[k]) >> 16)), zero OUTPUT border of kernSize/2.  kernSize odd, 3..15 (larger:
 * COMPVHIP_E_NOT_IMPLEMENTED), W,H >= kernSize.  The optional Gaussian pre-blur in front of Canny.  in and out may alias. */
COMPVHIP_API int compvhip_convlt1_fixedpoint_u8(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S,
                                                const uint16_t* vtKern, const uint16_t* hzKern, size_t kernSize, uint8_t* out, size_t So);

/* CompVMathConvlt::convlt1<uint8_t, int16_t, int16_t> and <int16_t, int16_t, int16_t> (base/include/compv/base/math/compv_math_convlt.h:26-28,
 * 37-39; driver :98-173, passes :176-292): separable integer CORRELATION (no kernel flip), horizontal pass with hzKern then vertical pass
 * with vtKern through an int16 temporary, int32 sums saturated to int16 after each pass, zero OUTPUT border of kernSize/2 columns and
 * rows.  The operator the Sobel / Scharr / Prewitt gradients are made of (gx: vt = smoothing, hz = derivative; canny_dete.cxx:237-241);
 * on the hot path it is fused into the tile kernels, this is its stand-alone form (reference known-answer vectors:
 * unittests/math_convlt.cxx:24-25).  kernSize odd, 1..15 (larger: COMPVHIP_E_NOT_IMPLEMENTED), W,H >= kernSize; S, So in ELEMENTS. */
COMPVHIP_API int compvhip_convlt1_8u16s16s(compvhip_ctx* ctx, const uint8_t* in, size_t W, size_t H, size_t S,
                                           const int16_t* vtKern, const int16_t* hzKern, size_t kernSize, int16_t* out, size_t So);
COMPVHIP_API int compvhip_convlt1_16s16s16s(compvhip_ctx* ctx, const int16_t* in, size_t W, size_t H, size_t S,
                                            const int16_t* vtKern, const int16_t* hzKern, size_t kernSize, int16_t* out, size_t So);

/* CompVHoughSht::process (core/features/hough/compv_core_feature_houghsht.cxx:96-262).  rho must be 1 (:306-316),
 * thetaDeg in degrees, threshold > 0 is the NMS/line threshold, maxLines <= 0 keeps every line.
 * lines: caller-allocated, capacity cap; *n receives the number of lines found (after the maxLines cut); if *n > cap
 * only cap lines are written and COMPVHIP_E_OUT_OF_BOUND is returned.  Lines come back in the REFERENCE's order: sorted by strength
 * descending, and inside equal-strength groups -- also at the maxLines cut -- exactly as the reference's unstable std::sort (:241-249)
 * leaves them when it is built with this C++ runtime (the whole list is brought to the host in the (row, col) emission order of
 * nms_apply and put through the same std::sort; callers such as CompVCalibCamera's line grouping depend on that order).  The
 * device-resident plan API keeps the canonical order instead (strength, then (row, col) ascending), see compvhip_plan_houghsht.
 * acc (optional): int32 accumulator in the reference layout, R rows of accStride elements, R = 2(W+H)+1. */
COMPVHIP_API int compvhip_houghsht_u8(compvhip_ctx* ctx, const uint8_t* edges, size_t W, size_t H, size_t S,
                                      float rho, float thetaDeg, int threshold, int maxLines,
                                      compvhip_line* lines, size_t cap, size_t* n,
                                      int32_t* acc, size_t accStride);

/* CompVHoughKht::process (core/features/hough/compv_core_feature_houghkht.cxx:208-447): kernel-based Hough transform.
 * rho in (0,1], thetaDeg in degrees, threshold on the 3x3-smoothed vote count, maxLines <= 0 keeps every line;
 * clusterMinDeviation / clusterMinSize (> 0) / kernelMinHeight (>= 0) are the COMPV_HOUGHKHT_SET_* knobs (defaults 2.0, 10, 0.002,
 * houghkht.cxx:38-40; ranges as CompVHoughKht::set checks them, :169-186).  Lines come back in the reference's order (descending smoothed count, the reference's own
 * std::sort tie order); rho is measured from the image centre (toCartesian, :1249-1280); row/col of compvhip_line hold
 * the rho/theta indices.  *gs receives COMPV_HOUGHKHT_GET_FLT64_GS when kernels survive (left untouched otherwise, like
 * the reference's m_dGS).  Hybrid: edge linking (Appendix A follows chains pixel by pixel in raster order and erases what it visits)
 * and the final sweep are order dependent and run on the host inside this library; cluster subdivision (one thread per string), the
 * per-cluster statistics (Algorithm 2), Gaussian voting (Algorithm 4) and vote-map smoothing run on the GPU. */
COMPVHIP_API int compvhip_houghkht_u8(compvhip_ctx* ctx, const uint8_t* edges, size_t W, size_t H, size_t S,
                                      float rho, float thetaDeg, int threshold, int maxLines,
                                      double clusterMinDeviation, size_t clusterMinSize, double kernelMinHeight,
                                      compvhip_line* lines, size_t cap, size_t* n, double* gs);

/* Stage inspection (tests, profiling): the elliptical kernels of voting_Algorithm2_Kernels (houghkht.cxx:885-1026) BEFORE the height
 * pruning, in cluster order, 7 doubles each in the field order of CompVHoughKhtKernel (houghkht.h:52-62): rho, theta (degrees), h,
 * sigma_theta_square, sigma_rho_square, m2, sigma_rho_times_theta; *hmax = the largest height.  COMPVHIP_E_OUT_OF_BOUND (with *n = the
 * number of kernels) when cap is too small.  compvhip_houghkht_stage_ms: wall-clock milliseconds of the six stages of this context's
 * last compvhip_houghkht_u8 call -- linking, subdivision (upload + GPU), statistics (GPU + download), pruning + Gmin, voting + peaks (GPU),
 * sort + sweep. */
COMPVHIP_API int compvhip_houghkht_kernels_u8(compvhip_ctx* ctx, const uint8_t* edges, size_t W, size_t H, size_t S,
                                              double clusterMinDeviation, size_t clusterMinSize,
                                              double* kernels7, size_t cap, size_t* n, double* hmax);
COMPVHIP_API int compvhip_houghkht_stage_ms(compvhip_ctx* ctx, double* ms6);
/* The host stage of KHT on its own -- linking_AppendixA (houghkht.cxx:544-760) on the bit-plane linker, no device involved (ctx is not needed):
 * xy receives the (x, y) pairs of the points of all strings, string after string (2 * *nPoints int32; cap = capacity in POINTS), stringEnds[i] the
 * index one past the last point of string i (string i = points [stringEnds[i-1], stringEnds[i])).  COMPVHIP_E_OUT_OF_BOUND when a capacity is too
 * small (*nPoints / *nStrings hold what is needed).  What CPU-only hosts test the linker with. */
COMPVHIP_API int compvhip_houghkht_link_u8(const uint8_t* edges, size_t W, size_t H, size_t S, size_t clusterMinSize, int32_t* xy, size_t cap, size_t* nPoints,
                                           uint32_t* stringEnds, size_t stringCap, size_t* nStrings);

/* CompVHoughSht::toCartesian (core/features/hough/compv_core_feature_houghsht.cxx:264-304,566-589) and
 * CompVHoughKht::toCartesian (core/features/hough/compv_core_feature_houghkht.cxx:449-489,1249-1280) for caller-held polar lines:
 * out[4 i ..] = {a.x, a.y, b.x, b.y} of line i (a.z = b.z = 1 in CompVLineFloat32).  SHT: rho from the image origin, endpoints at
 * x = 0 and x = W; KHT: rho from the image centre.  theta == 0 is the perfect vertical line (x = rho, y = +-sqrt(W^2 + H^2)).
 * Host float32 arithmetic in the reference's operation order (libm cosf / sinf called separately); no GPU involved.  Only the
 * rho / theta fields of `lines` are read. */
COMPVHIP_API int compvhip_houghsht_to_cartesian(size_t W, size_t H, const compvhip_line* lines, size_t n, float* out);
COMPVHIP_API int compvhip_houghkht_to_cartesian(size_t W, size_t H, const compvhip_line* lines, size_t n, float* out);

/* Geometry helper: R (rho rows), T (theta bins) and the float32 theta step for a W x H image
 * (initCoords, houghsht.cxx:318-348). */
COMPVHIP_API int compvhip_houghsht_dims(size_t W, size_t H, float thetaDeg, size_t* R, size_t* T, float* thetaStepRad);
/* Geometry helper of the KHT: rho bins and theta bins of its vote map for a W x H image (initCoords,
 * core/features/hough/compv_core_feature_houghkht.cxx:501-541); the map holds (T + 2) x (rhoN + 2) int32 cells. */
COMPVHIP_API int compvhip_houghkht_dims(size_t W, size_t H, float rho, float thetaDeg, size_t* rhoN, size_t* T);
/* Geometry helper: the grid of image tiles (nx x ny) and the rho-window rows per tile a plan of `frames` W x H frames votes with
 * (acc_gather, houghsht.cxx:350-481, runs as one workgroup per frame, tile and 64 theta bins).  Host arithmetic only: what
 * compvhip_plan_create would choose, for tests and capacity planning.  COMPVHIP_E_NOT_IMPLEMENTED when no grid fits. */
COMPVHIP_API int compvhip_houghsht_vote_grid(size_t W, size_t H, float thetaDeg, size_t frames, int* nx, int* ny, int* windowRows);
/* The CPUs this process may really use at once: min(hardware threads, affinity mask, cgroup CPU quota) -- what compvhip_plan_houghkht sizes its default
 * worker pool by (hostThreads = 0), and what the reference's CompVBase::init(-1) (base/compv_base.cxx:62: one thread per logical CPU) does not look at: a
 * container may show 256 logical CPUs and own 16.  Host arithmetic only; always >= 1. */
COMPVHIP_API int compvhip_host_cpu_budget(void);

/* ---- device-resident batched pipeline (frames already in HBM) ---------------------------------------------- */

/* A plan owns every scratch buffer for `frames` frames of W x H (stride S, S % 8 == 0, frame stride S*H) and the
 * Q16 sin/cos tables for thetaDeg.  Frames are independent units: one plan per GPU, shard frames across GPUs. */
COMPVHIP_API int compvhip_plan_create(compvhip_ctx* ctx, size_t W, size_t H, size_t S, size_t frames, float thetaDeg,
                                      compvhip_plan** plan);
/* A plan must be destroyed BEFORE its context (it keeps a pointer to it); compvhip_ctx_destroy only releases the private
 * single-frame plan of the host entry points. */
COMPVHIP_API void compvhip_plan_destroy(compvhip_plan* plan);

/* Canny on `frames` device frames: d_in -> d_edges (both frames*S*H bytes, may alias).  Asynchronous on `stream`
 * (a hipStream_t; NULL = default stream) except for the hysteresis convergence check, which polls a device flag. */
COMPVHIP_API int compvhip_plan_canny(compvhip_plan* plan, const uint8_t* d_in, float tLow, float tHigh, int ksize,
                                     int thresholdType, uint8_t* d_edges, void* stream);

/* compvhip_grayscale_u8 on `frames` device frames: d_in = [frames][H][S samples], d_gray = [frames][H][S].  Asynchronous. */
COMPVHIP_API int compvhip_plan_grayscale(compvhip_plan* plan, const uint8_t* d_in, int pixfmt, uint8_t* d_gray, void* stream);

/* compvhip_otsu_u8 on `frames` device frames: d_thresholds[f] = Otsu level of frame f (device array).  Asynchronous. */
COMPVHIP_API int compvhip_plan_otsu(compvhip_plan* plan, const uint8_t* d_gray, int32_t* d_thresholds, void* stream);

/* compvhip_convlt1_fixedpoint_u8 on `frames` device frames (vtKern/hzKern are HOST arrays of kernSize weights); d_in and
 * d_out may alias.  Asynchronous. */
COMPVHIP_API int compvhip_plan_convlt1_fixedpoint(compvhip_plan* plan, const uint8_t* d_in, const uint16_t* vtKern, const uint16_t* hzKern,
                                                  size_t kernSize, uint8_t* d_out, void* stream);

/* CompVHoughSht::toCartesian (core/features/hough/compv_core_feature_houghsht.cxx:264-304,566-589) on the device line arrays a
 * compvhip_plan_houghsht / _pipeline call produced: d_cart[f][i] = {a.x, a.y, b.x, b.y} of line i of frame f (a.z = b.z = 1), for
 * i < min(d_counts[f], lineCap).  Asynchronous. */
COMPVHIP_API int compvhip_plan_to_cartesian(compvhip_plan* plan, const compvhip_line* d_lines, const int32_t* d_counts, size_t lineCap,
                                            float* d_cart, void* stream);

/* Sobel / Scharr / Prewitt detector (compvhip_edge_dete_u8 semantics) on `frames` device frames; d_in and d_out must
 * not alias.  Fully asynchronous on `stream`. */
COMPVHIP_API int compvhip_plan_edge_dete(compvhip_plan* plan, const uint8_t* d_in, int op, uint8_t* d_out, void* stream);

/* SHT on the edge maps produced by the last compvhip_plan_canny() of this plan (uses its 1-bit edge masks, no byte
 * re-read) or, when d_edges != NULL, on arbitrary device edge maps.  Results stay on the device:
 * d_lines: frames * lineCap compvhip_line (strength descending, ties by accumulator (row, col) ascending -- the canonical order, not the
 * reference's unstable-sort tie order the host entry point reproduces), d_counts: frames int32 (lines found,
 * before clipping to lineCap).  The call is asynchronous and cannot grow its buffers after the fact: the device key buffer holds
 * min(R*T, max(lineCap, 65536)) candidates per frame, so whatever lineCap is, the lineCap STRONGEST lines are returned as long as
 * d_counts[f] <= max(lineCap, 65536); beyond that the key buffer overflowed and frame f's lines are an arbitrary subset -- call
 * again with lineCap >= d_counts[f] (the host entry point compvhip_houghsht_u8 does that by itself). */
COMPVHIP_API int compvhip_plan_houghsht(compvhip_plan* plan, const uint8_t* d_edges, int threshold, int maxLines,
                                        compvhip_line* d_lines, size_t lineCap, int32_t* d_counts, void* stream);

/* Sobel -> Canny -> HoughSHT in one call (the benchmark's "step"). */
COMPVHIP_API int compvhip_plan_pipeline(compvhip_plan* plan, const uint8_t* d_in, float tLow, float tHigh,
                                        int threshold, int maxLines, uint8_t* d_edges,
                                        compvhip_line* d_lines, size_t lineCap, int32_t* d_counts, void* stream);

/* compvhip_plan_pipeline without its host round trip.  The synchronous call reads the hysteresis convergence flag before it
 * returns (one stream synchronisation per step); this one enqueues the step, lets the flag travel to pinned host memory behind
 * the kernels and returns a ticket.  compvhip_plan_wait(plan, ticket) blocks until that step has finished and, in the rare case
 * its hysteresis needed more resolve rounds than were enqueued speculatively, drains the stream and runs the step again
 * synchronously -- so d_in must stay unmodified, and distinct from d_edges, until the step was waited for.  Up to 4 steps may be
 * in flight (further calls return COMPVHIP_E_INVALID_STATE); steps of one plan must use one stream.  Typical use:
 * t1 = async(batch k+1); wait(t0) -- the GPU never idles between steps.
 * Output buffers and the replay: a replayed step writes its d_edges / d_lines / d_counts again, AFTER later steps of the plan have run.  Steps in
 * flight may share output buffers (a caller that only consumes the newest result): every step enqueued after a replayed one is then replayed too
 * when it is waited for, in enqueue order, so after compvhip_plan_wait(t) the buffers of step t always hold step t's results.  Wait for the tickets
 * of a plan in the order they were issued.  Results of step t are only guaranteed to still be there until the next step that shares its buffers
 * starts -- give steps their own buffers to read them later.
 * Speculative rounds: a step enqueues as many hysteresis rounds as the plan's last 8 asynchronous steps needed (the first round that changed nothing,
 * inclusive): 3 for a new plan, 2 once four steps have needed no more (the benchmark's frames: round 0 does the work, round 1 confirms), never more than 3; a
 * step that needs more is the replay described above, and the plan enqueues 3 again for the steps after it.
 * A second replay cause exists only on the library-sort fallback (max(W, H) > 4095, or more than 32 chunks of 4096 lines per frame): there the step
 * sorts a PREDICTED range of the line keys -- the largest line total of the plan's last 8 steps + 1/16 + 4096 -- and compvhip_plan_wait replays the
 * step when its real total exceeded the prediction.  Content whose line count jumps from step to step therefore replays often on such plans, and each
 * replay drains the stream and cascades to the later tickets that share output buffers; plans on the device-sized sort (every size up to 4095 x 4095
 * with at most 131 072 lines per frame, i.e. all BASELINE configurations) never replay for this reason. */
COMPVHIP_API int compvhip_plan_pipeline_async(compvhip_plan* plan, const uint8_t* d_in, float tLow, float tHigh,
                                              int threshold, int maxLines, uint8_t* d_edges,
                                              compvhip_line* d_lines, size_t lineCap, int32_t* d_counts, void* stream, int* ticket);
COMPVHIP_API int compvhip_plan_wait(compvhip_plan* plan, int ticket);

/* The general step: what samples/hough_lines/main.cxx:102-109 does per camera frame -- convertGrayscale -> thresholdOtsu ->
 * Canny(Otsu * tLow, Otsu * tHigh) -> HoughSHT -> toCartesian -- as ONE enqueue over the plan's frames.  Zero-initialise the struct and set what
 * is needed; compvhip_plan_pipeline(_async) is this call with ksize 3, COMPARE_TO_GRADIENT, FMT_Y and no optional outputs. */
typedef struct compvhip_pipeline_opts {
	float tLow, tHigh;      /* Canny thresholds (threshold factors in the PERCENT_OF_MEAN / OTSU modes) */
	int threshold, maxLines;/* SHT threshold (> 0) / line cut (<= 0: every line) */
	int ksize;              /* Sobel kernel size of the gradient: 3 or 5 (0 = 3) */
	int thresholdType;      /* COMPVHIP_CANNY_THRESHOLD_* */
	int pixfmt;             /* compvhip_pixfmt of d_in; COMPVHIP_FMT_Y (luma plane, stride S) or a packed format ([frames][H][S samples]): converted first */
	uint8_t* d_gray;        /* packed input: receives the luma planes [frames][H][S] (NULL: plan-owned scratch) */
	int32_t* d_otsu;        /* OTSU mode: receives the per-frame Otsu levels (NULL: not wanted) */
	float* d_cart;          /* receives toCartesian's endpoints [frames][lineCap][4] (NULL: not wanted) */
} compvhip_pipeline_opts;
/* ticket == NULL: synchronous like compvhip_plan_pipeline; otherwise asynchronous like compvhip_plan_pipeline_async (same rules: wait with
 * compvhip_plan_wait, d_in unmodified and distinct from d_edges until then). */
COMPVHIP_API int compvhip_plan_pipeline_ex(compvhip_plan* plan, const uint8_t* d_in, const compvhip_pipeline_opts* opts, uint8_t* d_edges,
                                           compvhip_line* d_lines, size_t lineCap, int32_t* d_counts, void* stream, int* ticket);

/* CompVHoughKht::process (compvhip_houghkht_u8 semantics, same knobs) on the plan's `frames` DEVICE edge maps d_edges = [frames][H][S]
 * ({0, non-zero} bytes, e.g. the edge maps a compvhip_plan_canny / _pipeline call produced); results in HOST memory: lines[f * cap ..] /
 * counts[f] / gs[f] (gs optional; gs[f] is left untouched for a frame without surviving kernels, like the reference's m_dGS).  Synchronous,
 * and it drains the device first (hipDeviceSynchronize): whatever stream produced d_edges has finished before a worker reads them.
 * The edge-linking stage is a sequential chain walk per frame and runs on the host (on a bit plane: the edge maps leave the device as bit masks);
 * frames are independent, so they go through the stages in groups of 8: inside a group the host stages (linking, prune / Gmin, sort + sweep) are
 * parallel loops over its frames and every GPU stage is ONE launch over the strings / clusters / kernels / vote maps of all its frames; up to eight
 * groups are in flight, each with its own stream and a controller thread that only enqueues, SLEEPS on its GPU stages (blocking-sync events) and
 * posts its group's host stages to the hostThreads workers all groups share (it works along on the short prune items only) -- so while one group is on the
 * GPU the workers link / sweep the frames of the others.  hostThreads = 0: min(32, hardware threads / 2, the CPUs this process may really use: affinity mask and cgroup CPU quota).  COMPVHIP_E_OUT_OF_BOUND when a frame has more
 * than cap lines (counts[f] tells); on any other failure the error text names the frame.  clusterMinSize must be >= 2 (for 1 the reference's
 * cluster subdivision does not terminate: a defined deviation, also of compvhip_houghkht_u8 / compvhip_houghkht_kernels_u8).
 * compvhip_plan_houghkht_stage_ms: the six stage clocks of the last call summed over its frames (compvhip_houghkht_stage_ms order), the wall
 * time of the call and the number of workers. */
COMPVHIP_API int compvhip_plan_houghkht(compvhip_plan* plan, const uint8_t* d_edges, float rho, float thetaDeg, int threshold, int maxLines,
                                        double clusterMinDeviation, size_t clusterMinSize, double kernelMinHeight,
                                        compvhip_line* lines, size_t cap, size_t* counts, double* gs, int hostThreads);
COMPVHIP_API int compvhip_plan_houghkht_stage_ms(compvhip_plan* plan, double* ms6, double* wallMs, int* threads);

/* Device accumulator of frame f after compvhip_plan_houghsht: uint16 (a cell never exceeds the pixels of a 1-px band),
 * theta-major [T][accPitch] (pitch >= R).  compvhip_plan_acc_export gives the reference's int32 rho-major layout. */
COMPVHIP_API int compvhip_plan_acc(compvhip_plan* plan, size_t frame, const uint16_t** d_acc, size_t* R, size_t* T, size_t* accPitch);
/* Copies the accumulator of frame f into a caller DEVICE buffer in the reference layout: int32 [R][outStride]
 * (outStride >= T), i.e. acc[(barrier - rho) * outStride + t] (houghsht.cxx:430-431). */
COMPVHIP_API int compvhip_plan_acc_export(compvhip_plan* plan, size_t frame, int32_t* d_out, size_t outStride, void* stream);
/* Number of edge pixels per frame found by the last canny/houghsht of this plan (device int32[frames]). */
COMPVHIP_API int compvhip_plan_edge_counts(compvhip_plan* plan, const int32_t** d_edge_counts);

/* Per-kernel timing of the last plan call, measured with hipEvents on the stream the kernels were launched on.
 * names/ms: caller arrays of capacity cap; returns the number of entries (<= cap). compvhip_plan_set_timing(plan, mode):
 * 0 = off, 1 = every kernel, 2 = only canny_tile_kernel and sht_vote_kernel, 3 = only sht_vote_kernel, 4 = only canny_tile_kernel
 * (an event pair costs ~10-20 us of stream time and lengthens the bracketed kernel: the narrow modes keep that out of a
 * throughput measurement). */
COMPVHIP_API int compvhip_plan_set_timing(compvhip_plan* plan, int enabled);
COMPVHIP_API int compvhip_plan_get_timing(compvhip_plan* plan, const char** names, float* ms, int cap);

#ifdef __cplusplus
}
#endif
#endif /* COMPV_HIP_H */
