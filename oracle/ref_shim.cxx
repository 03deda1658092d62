// TEST INFRASTRUCTURE ONLY.  Flat C wrappers over the *real* CompV C++ API (compiled from
// /root/reference by oracle/build_ref.sh into oracle/_ref/).  Used to (1) validate the C restatement
// in oracle/compv_oracle.c, (2) generate the golden fixtures under tests/golden/, and (3) serve as the
// "reference" CPU baseline in bench.py (cpu_baseline.kind == "reference").  Never linked into, loaded
// by or called from the product library (compv_amd/).
//
// This file contains no reference source: it only *calls* the public API
//   CompVEdgeDete::newObj / process      base/include/compv/base/compv_features.h:207-215
//   CompVHough::newObj / process         base/include/compv/base/compv_features.h:218-227
//   CompVMathConvlt::convlt1             base/include/compv/base/math/compv_math_convlt.h:26-28
// and, for the SHT accumulator (which process() keeps private), the exported SIMD leaves
//   CompVHoughShtRowTimesSinRho_Intrin_SSE41 / CompVHoughShtAccGatherRow_8mpd_Intrin_AVX2
//   (core/features/hough/intrin/x86/*.h) in the same order acc_gather() calls them
//   (core/features/hough/compv_core_feature_houghsht.cxx:350-481).
#include <compv/base/compv_base.h>
#include <compv/base/compv_cpu.h>
#include <compv/base/compv_features.h>
#include <compv/base/compv_debug.h>
#include <compv/base/compv_mem.h>
#include <compv/base/image/compv_image.h>
#include <compv/base/math/compv_math_convlt.h>
#include <compv/base/math/compv_math_gauss.h>
#include <compv/base/parallel/compv_parallel.h>
#include <compv/core/compv_core.h>
#include <compv/core/features/hough/intrin/x86/compv_core_feature_houghsht_intrin_sse41.h>
#include <compv/core/features/hough/intrin/x86/compv_core_feature_houghsht_intrin_avx2.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace compv;

namespace {
struct State {
	bool inited = false;
	int threads = 0;
	CompVEdgeDetePtr sobel, canny;
	CompVHoughPtr sht, kht;
	CompVMatPtr img, out;
} g;

COMPV_ERROR_CODE toMat(const uint8_t* in, size_t W, size_t H, size_t S, CompVMatPtr* mat)
{
	COMPV_CHECK_CODE_RETURN(CompVImage::newObj8u(mat, COMPV_SUBTYPE_PIXELS_Y, W, H));
	for (size_t j = 0; j < H; ++j) {
		memcpy((*mat)->ptr<uint8_t>(j), in + j * S, W);
	}
	return COMPV_ERROR_CODE_S_OK;
}
void fromMat(const CompVMatPtr& mat, uint8_t* out, size_t So)
{
	for (size_t j = 0; j < mat->rows(); ++j) {
		memcpy(out + j * So, mat->ptr<const uint8_t>(j), mat->cols());
	}
}
double nowMs()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}

extern "C" {

// threads: 1 = single thread, -1 = all cores (CompVBase::init semantics, base/compv_base.cxx:62)
int refshim_init(int threads)
{
	if (g.inited && g.threads == threads) return 0;
	if (g.inited) {
		g.sobel = nullptr; g.canny = nullptr; g.sht = nullptr; g.kht = nullptr; g.img = nullptr; g.out = nullptr;
		CompVCore::deInit();
		CompVBase::deInit();
		g.inited = false;
	}
	CompVDebugMgr::setLevel(COMPV_DEBUG_LEVEL_ERROR);
	if (COMPV_ERROR_CODE_IS_NOK(CompVBase::init(threads))) return -1;
	if (COMPV_ERROR_CODE_IS_NOK(CompVCore::init())) return -2;
	g.inited = true;
	g.threads = threads;
	return 0;
}

int refshim_threads()
{
	CompVThreadDispatcherPtr d = CompVParallel::threadDispatcher();
	return d ? (int)d->threadsCount() : 1;
}

// returns 1 when the AVX2 intrinsics path is the one the reference dispatches to on this host
int refshim_has_avx2()
{
	return CompVCpu::isEnabled(kCpuFlagAVX2) ? 1 : 0;
}

// Sobel detector (id COMPV_SOBEL_ID). out: H rows of W bytes at stride So.
int refshim_sobel(const uint8_t* in, size_t W, size_t H, size_t S, uint8_t* out, size_t So)
{
	CompVMatPtr img, edges;
	if (COMPV_ERROR_CODE_IS_NOK(toMat(in, W, H, S, &img))) return -1;
	CompVEdgeDetePtr dete;
	if (COMPV_ERROR_CODE_IS_NOK(CompVEdgeDete::newObj(&dete, COMPV_SOBEL_ID))) return -2;
	if (COMPV_ERROR_CODE_IS_NOK(dete->process(img, &edges))) return -3;
	fromMat(edges, out, So);
	return 0;
}

// Any of the three gradient detectors. op: 0 = COMPV_SOBEL_ID, 2 = COMPV_SCHARR_ID, 3 = COMPV_PREWITT_ID (the ids of include/compv_hip.h).
int refshim_edge_dete(const uint8_t* in, size_t W, size_t H, size_t S, int op, uint8_t* out, size_t So)
{
	CompVMatPtr img, edges;
	if (COMPV_ERROR_CODE_IS_NOK(toMat(in, W, H, S, &img))) return -1;
	CompVEdgeDetePtr dete;
	const int id = op == 0 ? COMPV_SOBEL_ID : (op == 2 ? COMPV_SCHARR_ID : (op == 3 ? COMPV_PREWITT_ID : -1));
	if (id < 0) return -4;
	if (COMPV_ERROR_CODE_IS_NOK(CompVEdgeDete::newObj(&dete, id))) return -2;
	if (COMPV_ERROR_CODE_IS_NOK(dete->process(img, &edges))) return -3;
	fromMat(edges, out, So);
	return 0;
}

// Canny. thresholdType: 0 = COMPARE_TO_GRADIENT (default), 1 = PERCENT_OF_MEAN
int refshim_canny(const uint8_t* in, size_t W, size_t H, size_t S, float tLow, float tHigh, int ksize, int thresholdType, uint8_t* out, size_t So)
{
	CompVMatPtr img, edges;
	if (COMPV_ERROR_CODE_IS_NOK(toMat(in, W, H, S, &img))) return -1;
	CompVEdgeDetePtr dete;
	if (COMPV_ERROR_CODE_IS_NOK(CompVEdgeDete::newObj(&dete, COMPV_CANNY_ID, tLow, tHigh, (size_t)ksize))) return -2;
	if (thresholdType == 1) {
		if (COMPV_ERROR_CODE_IS_NOK(dete->setInt(COMPV_CANNY_SET_INT_THRESHOLD_TYPE, COMPV_CANNY_THRESHOLD_TYPE_PERCENT_OF_MEAN))) return -4;
	}
	COMPV_ERROR_CODE err = dete->process(img, &edges);
	if (COMPV_ERROR_CODE_IS_NOK(err)) return -3;
	fromMat(edges, out, So);
	return 0;
}

struct RefLine { float rho; float theta; long long strength; };

static int houghRun(int id, const uint8_t* edges, size_t W, size_t H, size_t S, float rho, float thetaDeg, size_t threshold, int maxLines,
	RefLine* lines, size_t cap, size_t* n, double* gs)
{
	CompVMatPtr img;
	if (COMPV_ERROR_CODE_IS_NOK(toMat(edges, W, H, S, &img))) return -1;
	CompVHoughPtr h;
	if (COMPV_ERROR_CODE_IS_NOK(CompVHough::newObj(&h, id, rho, thetaDeg, threshold))) return -2;
	if (maxLines > 0) h->setInt(COMPV_HOUGH_SET_INT_MAXLINES, maxLines);
	CompVHoughLineVector v;
	if (COMPV_ERROR_CODE_IS_NOK(h->process(img, v))) return -3;
	*n = v.size();
	for (size_t i = 0; i < v.size() && i < cap; ++i) {
		lines[i].rho = v[i].rho; lines[i].theta = v[i].theta; lines[i].strength = (long long)v[i].strength;
	}
	if (gs) {
		compv_float64_t val = 0;
		h->getFloat64(COMPV_HOUGHKHT_GET_FLT64_GS, &val);
		*gs = val;
	}
	return 0;
}

int refshim_sht(const uint8_t* edges, size_t W, size_t H, size_t S, float thetaDeg, size_t threshold, int maxLines, RefLine* lines, size_t cap, size_t* n)
{
	return houghRun(COMPV_HOUGHSHT_ID, edges, W, H, S, 1.f, thetaDeg, threshold, maxLines, lines, cap, n, nullptr);
}

int refshim_kht(const uint8_t* edges, size_t W, size_t H, size_t S, float rho, float thetaDeg, size_t threshold, int maxLines, RefLine* lines, size_t cap, size_t* n, double* gs)
{
	return houghRun(COMPV_HOUGHKHT_ID, edges, W, H, S, rho, thetaDeg, threshold, maxLines, lines, cap, n, gs);
}

// Full SHT accumulator through the reference's own SIMD leaves. sinQ/cosQ: the Q16 tables (int32[T], T multiple of 8
// is required by the AVX2 leaf for the part it consumes; the trailing thetas use the scalar expression exactly as
// acc_gather does).  acc: int32[R*accStride] zero-initialised by the caller, R = 2(W+H)+1, barrier = W+H.
int refshim_sht_acc(const uint8_t* edges, size_t W, size_t H, size_t S, const int32_t* sinQ, const int32_t* cosQ, size_t T, int32_t* acc, size_t accStride)
{
	if (T < 32) return -1;
	const size_t Tpad = (T + 15) & ~(size_t)15;
	int32_t* sinA = (int32_t*)CompVMem::mallocAligned((Tpad + 16) * sizeof(int32_t));
	int32_t* cosA = (int32_t*)CompVMem::mallocAligned((Tpad + 16) * sizeof(int32_t));
	int32_t* rts = (int32_t*)CompVMem::mallocAligned((Tpad + 16) * sizeof(int32_t));
	if (!sinA || !cosA || !rts) return -2;
	memset(sinA, 0, (Tpad + 16) * 4); memset(cosA, 0, (Tpad + 16) * 4);
	memcpy(sinA, sinQ, T * 4); memcpy(cosA, cosQ, T * 4);
	const int32_t barrier = (int32_t)(W + H);
	int32_t* pACC = acc + (size_t)barrier * accStride;
	const size_t consumed = T & ~(size_t)7;
	for (size_t j = 0; j < H; ++j) {
		bool haveRow = false;
		for (size_t i = 0; i < W; ++i) {
			if (!edges[j * S + i]) continue;
			if (!haveRow) {
				CompVHoughShtRowTimesSinRho_Intrin_SSE41(sinA, (compv_uscalar_t)j, rts, (compv_uscalar_t)T);
				haveRow = true;
			}
			CompVHoughShtAccGatherRow_8mpd_Intrin_AVX2(cosA, rts, (compv_uscalar_t)i, pACC, (compv_uscalar_t)accStride, (compv_uscalar_t)consumed);
			for (size_t t = consumed; t < T; ++t) {
				const int32_t rho = ((int32_t)i * cosA[t] + rts[t]) >> 16;
				pACC[(ptrdiff_t)t - (ptrdiff_t)rho * (ptrdiff_t)accStride]++;
			}
		}
	}
	CompVMem::free((void**)&sinA); CompVMem::free((void**)&cosA); CompVMem::free((void**)&rts);
	return 0;
}

// Separable correlation exactly as the two hot-path instantiations (unittests/math_convlt.cxx cases 5 and 6).
int refshim_convlt1_8u16s16s(const uint8_t* in, size_t W, size_t H, size_t S, const int16_t* vt, const int16_t* hz, size_t k, int16_t* out)
{
	int16_t* o = out;
	return COMPV_ERROR_CODE_IS_OK((CompVMathConvlt::convlt1<uint8_t, int16_t, int16_t>(in, W, H, S, vt, hz, k, o))) ? 0 : -1;
}
int refshim_convlt1_16s16s16s(const int16_t* in, size_t W, size_t H, size_t S, const int16_t* vt, const int16_t* hz, size_t k, int16_t* out)
{
	int16_t* o = out;
	return COMPV_ERROR_CODE_IS_OK((CompVMathConvlt::convlt1<int16_t, int16_t, int16_t>(in, W, H, S, vt, hz, k, o))) ? 0 : -1;
}

// ---- timing legs for bench.py cpu_baseline (kind == "reference") ----
// Steady-state per-frame time in ms of Canny -> SHT on `frames` frames laid out back to back (frame stride S*H),
// detector objects reused across frames exactly as tests/image/canny.cxx:64-69 loops process().
// stage mask: 1 = canny, 2 = sht on the canny output. Returns total wall ms; writes edge px and line totals.
double refshim_bench_pipeline(const uint8_t* in, size_t W, size_t H, size_t S, size_t frames, float tLow, float tHigh,
	float thetaDeg, size_t threshold, int stages, long long* edgePx, long long* nLines)
{
	CompVEdgeDetePtr canny; CompVHoughPtr sht;
	if (COMPV_ERROR_CODE_IS_NOK(CompVEdgeDete::newObj(&canny, COMPV_CANNY_ID, tLow, tHigh, 3))) return -1;
	if (COMPV_ERROR_CODE_IS_NOK(CompVHough::newObj(&sht, COMPV_HOUGHSHT_ID, 1.f, thetaDeg, threshold))) return -1;
	std::vector<CompVMatPtr> imgs(frames);
	for (size_t f = 0; f < frames; ++f) {
		if (COMPV_ERROR_CODE_IS_NOK(toMat(in + f * S * H, W, H, S, &imgs[f]))) return -1;
	}
	CompVMatPtr edges; CompVHoughLineVector lines;
	long long e = 0, l = 0;
	const double t0 = nowMs();
	for (size_t f = 0; f < frames; ++f) {
		if (COMPV_ERROR_CODE_IS_NOK(canny->process(imgs[f], &edges))) return -1;
		if (stages & 2) {
			if (COMPV_ERROR_CODE_IS_NOK(sht->process(edges, lines))) return -1;
			l += (long long)lines.size();
		}
	}
	const double t1 = nowMs();
	if (edges) {
		for (size_t j = 0; j < H; ++j) { const uint8_t* p = edges->ptr<const uint8_t>(j); for (size_t i = 0; i < W; ++i) e += p[i] ? 1 : 0; }
	}
	if (edgePx) *edgePx = e;
	if (nLines) *nLines = l;
	return t1 - t0;
}

// Per-call wall time of CompVHoughKht::process on `frames` edge maps laid out back to back (frame stride S*H), one Hough object reused
// across the frames as tests/image/houghkht.cxx loops process().  Returns total wall ms.
double refshim_bench_kht(const uint8_t* edges, size_t W, size_t H, size_t S, size_t frames, float rho, float thetaDeg, size_t threshold, long long* nLines)
{
	CompVHoughPtr kht;
	if (COMPV_ERROR_CODE_IS_NOK(CompVHough::newObj(&kht, COMPV_HOUGHKHT_ID, rho, thetaDeg, threshold))) return -1;
	std::vector<CompVMatPtr> maps(frames);
	for (size_t f = 0; f < frames; ++f) {
		if (COMPV_ERROR_CODE_IS_NOK(toMat(edges + f * S * H, W, H, S, &maps[f]))) return -1;
	}
	CompVHoughLineVector lines;
	long long l = 0;
	const double t0 = nowMs();
	for (size_t f = 0; f < frames; ++f) {
		if (COMPV_ERROR_CODE_IS_NOK(kht->process(maps[f], lines))) return -1;
		l += (long long)lines.size();
	}
	const double t1 = nowMs();
	if (nLines) *nLines = l;
	return t1 - t0;
}

// CompVImage::convertGrayscale (samples/hough_lines/main.cxx:102). fmt: index into the table below (the numbering of
// include/compv_hip.h, compvhip_pixfmt). in: H rows of S samples (S*bpp bytes per row). out: H rows of W bytes at stride So.
int refshim_grayscale(const uint8_t* in, int fmt, size_t W, size_t H, size_t S, uint8_t* out, size_t So)
{
	static const COMPV_SUBTYPE kFmt[] = {
		COMPV_SUBTYPE_PIXELS_RGBA32, COMPV_SUBTYPE_PIXELS_ARGB32, COMPV_SUBTYPE_PIXELS_BGRA32, COMPV_SUBTYPE_PIXELS_RGB24, COMPV_SUBTYPE_PIXELS_BGR24,
		COMPV_SUBTYPE_PIXELS_RGB565LE, COMPV_SUBTYPE_PIXELS_RGB565BE, COMPV_SUBTYPE_PIXELS_BGR565LE, COMPV_SUBTYPE_PIXELS_BGR565BE,
		COMPV_SUBTYPE_PIXELS_YUYV422, COMPV_SUBTYPE_PIXELS_UYVY422, COMPV_SUBTYPE_PIXELS_Y };
	if (fmt < 0 || fmt >= (int)(sizeof(kFmt) / sizeof(kFmt[0]))) return -1;
	CompVMatPtr img, gray;
	if (COMPV_ERROR_CODE_IS_NOK(CompVImage::wrap(kFmt[fmt], in, W, H, S, &img))) return -2;
	if (COMPV_ERROR_CODE_IS_NOK(CompVImage::convertGrayscale(img, &gray))) return -3;
	fromMat(gray, out, So);
	return 0;
}

// CompVImage::thresholdOtsu (samples/hough_lines/main.cxx:103)
int refshim_otsu(const uint8_t* in, size_t W, size_t H, size_t S, double* threshold)
{
	CompVMatPtr img;
	if (COMPV_ERROR_CODE_IS_NOK(toMat(in, W, H, S, &img))) return -1;
	double t = 0.0;
	if (COMPV_ERROR_CODE_IS_NOK(CompVImage::thresholdOtsu(img, t))) return -3;
	*threshold = t;
	return 0;
}

// CompVMathGauss::kernelDim1 (float) / kernelDim1FixedPoint and CompVMathConvlt::convlt1FixedPoint
int refshim_gauss_kernel_f32(size_t size, float sigma, float* kernel)
{
	CompVMatPtr k;
	if (COMPV_ERROR_CODE_IS_NOK(CompVMathGauss::kernelDim1<compv_float32_t>(&k, size, sigma))) return -1;
	memcpy(kernel, k->ptr<const compv_float32_t>(), size * sizeof(float));
	return 0;
}
int refshim_gauss_kernel_fixedpoint(size_t size, float sigma, uint16_t* kernel)
{
	CompVMatPtr k;
	if (COMPV_ERROR_CODE_IS_NOK(CompVMathGauss::kernelDim1FixedPoint(&k, size, sigma))) return -1;
	memcpy(kernel, k->ptr<const uint16_t>(), size * sizeof(uint16_t));
	return 0;
}
int refshim_convlt1_fixedpoint(const uint8_t* in, size_t W, size_t H, size_t S, const uint16_t* vt, const uint16_t* hz, size_t k, uint8_t* out)
{
	// aligned copies, as the reference's callers hold CompVMat data
	CompVMatPtr src, dst;
	if (COMPV_ERROR_CODE_IS_NOK(CompVMat::newObjAligned<uint8_t>(&src, H, W, S))) return -1;
	if (COMPV_ERROR_CODE_IS_NOK(CompVMat::newObjAligned<uint8_t>(&dst, H, W, S))) return -1;
	for (size_t j = 0; j < H; ++j) memcpy(src->ptr<uint8_t>(j), in + j * S, W);
	uint8_t* o = dst->ptr<uint8_t>();
	if (COMPV_ERROR_CODE_IS_NOK(CompVMathConvlt::convlt1FixedPoint(src->ptr<const uint8_t>(), W, H, src->stride(), vt, hz, k, o))) return -2;
	for (size_t j = 0; j < H; ++j) memcpy(out + j * S, dst->ptr<const uint8_t>(j), W);
	return 0;
}

// CompVHoughSht / CompVHoughKht::toCartesian on caller-provided polar lines: out[4*i..] = a.x, a.y, b.x, b.y
static int to_cartesian(int id, size_t W, size_t H, const RefLine* lines, size_t n, float* out);
int refshim_sht_to_cartesian(size_t W, size_t H, const RefLine* lines, size_t n, float* out) { return to_cartesian(COMPV_HOUGHSHT_ID, W, H, lines, n, out); }
int refshim_kht_to_cartesian(size_t W, size_t H, const RefLine* lines, size_t n, float* out) { return to_cartesian(COMPV_HOUGHKHT_ID, W, H, lines, n, out); }
static int to_cartesian(int id, size_t W, size_t H, const RefLine* lines, size_t n, float* out)
{
	CompVHoughPtr h;
	if (COMPV_ERROR_CODE_IS_NOK(CompVHough::newObj(&h, id, 1.f, 1.f, 1))) return -2;
	CompVHoughLineVector polar(n);
	for (size_t i = 0; i < n; ++i) { polar[i].rho = lines[i].rho; polar[i].theta = lines[i].theta; polar[i].strength = (size_t)lines[i].strength; }
	CompVLineFloat32Vector cart;
	if (COMPV_ERROR_CODE_IS_NOK(h->toCartesian(W, H, polar, cart))) return -3;
	for (size_t i = 0; i < n; ++i) { out[4 * i] = cart[i].a.x; out[4 * i + 1] = cart[i].a.y; out[4 * i + 2] = cart[i].b.x; out[4 * i + 3] = cart[i].b.y; }
	return 0;
}

} // extern "C"
