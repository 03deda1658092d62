// headless_samples.cxx -- the call sequences of samples/edges_canny/main.cxx:37-72 and samples/hough_lines/main.cxx:52-109
// (minus camera and window, which need devices a headless GPU box does not have), run TWICE against the real CompV
// library: first with the stock CPU factories, then after compv_hip_plugin_register() swapped the factories by id.
// The application code between the two runs is IDENTICAL -- only the registered factory differs -- and the outputs are
// compared: edge maps byte for byte, Hough lines as sets of (rho, theta, strength) (the reference's order among
// equal strengths is unspecified: unstable std::sort, core/features/hough/compv_core_feature_houghsht.cxx:243-249).
//
// usage: headless_samples [W H [frames [cpuThreads]]]      exit code 0 = drop-in parity on every frame
// cpuThreads (default 1) is CompVBase::init()'s thread count for the CPU reference run.  The default is the single-threaded
// path because the reference's multi-threaded gradient is not deterministic: each row band also recomputes |gx|+|gy| for
// its two overlap rows from gx/gy rows that the neighbouring band may not have written yet
// (core/features/edges/compv_core_feature_canny_dete.cxx:190-199), so with many threads and few rows per band the CPU
// result itself occasionally differs from run to run.
#include <compv/base/compv_base.h>
#include <compv/base/compv_features.h>
#include <compv/base/compv_debug.h>
#include <compv/base/image/compv_image.h>
#include <compv/core/compv_core.h>
#include <compv/core/calib/compv_core_calib_camera.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <vector>

using namespace compv;

extern "C" int compv_hip_plugin_register(void);
#include "../include/compv_hip.h" // the pre-processing calls are static CompVImage functions, not factories: called through the C ABI

// SURVEY.md 8(d) synthetic frame
static void synthFrame(CompVMatPtr img, uint32_t seed)
{
	const size_t W = img->cols(), H = img->rows();
	uint32_t s = seed;
	for (size_t j = 0; j < H; ++j) {
		uint8_t* p = img->ptr<uint8_t>(j);
		for (size_t i = 0; i < W; ++i) {
			s = s * 1664525u + 1013904223u;
			uint32_t v = 40u + ((((uint32_t)(i / 64) + (uint32_t)(j / 64)) & 1u) * 150u) + (s >> 28);
			if (((i + 2 * j) % 257) < 3) v = 255u;
			p[i] = (uint8_t)v;
		}
	}
}

struct Result {
	std::vector<uint8_t> sobel, canny, cannyMean;
	std::vector<std::tuple<float, float, size_t> > lines;
	std::vector<float> cart;
	std::vector<std::tuple<float, float, size_t> > khtLines; // in the order returned (the reference's sweep order)
	std::vector<float> khtCart;
	double khtGS;
	double ms;
};

// Detector objects are created once (the samples create them before the camera loop, samples/hough_lines/main.cxx:59-72)
// and reused for every frame; resetObjects() drops them when the factories are swapped.
static CompVEdgeDetePtr ptrCanny, ptrCannyMean, ptrSobel;
static CompVHoughPtr ptrHough, ptrKht;
static void resetObjects() { ptrCanny = nullptr; ptrCannyMean = nullptr; ptrSobel = nullptr; ptrHough = nullptr; ptrKht = nullptr; }

// The application code: same calls, same order, same parameters as the samples.
static COMPV_ERROR_CODE runSamples(size_t W, size_t H, uint32_t seed, Result& r)
{
	CompVMatPtr image, mat, edges, sob;
	CompVHoughLineVector linesPolar;
	CompVLineFloat32Vector linesCartesian;

	COMPV_CHECK_CODE_RETURN(CompVImage::newObj8u(&image, COMPV_SUBTYPE_PIXELS_Y, W, H));
	synthFrame(image, seed);

	if (!ptrSobel) {
		// samples/edges_sobel: CompVEdgeDete::newObj(&dete, COMPV_SOBEL_ID)
		COMPV_CHECK_CODE_RETURN(CompVEdgeDete::newObj(&ptrSobel, COMPV_SOBEL_ID));
		// samples/edges_canny/main.cxx:44-45: newObj(COMPV_CANNY_ID, low, high, kernel)
		COMPV_CHECK_CODE_RETURN(CompVEdgeDete::newObj(&ptrCanny, COMPV_CANNY_ID, 59.f, 119.f, 3));
		// samples/hough_lines/main.cxx:59-72
		COMPV_CHECK_CODE_RETURN(CompVHough::newObj(&ptrHough, COMPV_HOUGHSHT_ID, 1.f, 1.f, 100));
		COMPV_CHECK_CODE_RETURN(ptrHough->setInt(COMPV_HOUGH_SET_INT_MAXLINES, 0));
		COMPV_CHECK_CODE_RETURN(CompVHough::newObj(&ptrKht, COMPV_HOUGHKHT_ID, 1.f, 1.f, 1));
		COMPV_CHECK_CODE_RETURN(ptrKht->setInt(COMPV_HOUGH_SET_INT_MAXLINES, 0));
		COMPV_CHECK_CODE_RETURN(ptrKht->setFloat32(COMPV_HOUGHKHT_SET_FLT32_CLUSTER_MIN_DEVIATION, 2.0f));
		COMPV_CHECK_CODE_RETURN(ptrKht->setInt(COMPV_HOUGHKHT_SET_INT_CLUSTER_MIN_SIZE, 10));
		COMPV_CHECK_CODE_RETURN(ptrKht->setFloat32(COMPV_HOUGHKHT_SET_FLT32_KERNEL_MIN_HEIGTH, 0.002f));
		// hough_lines drives the Canny thresholds through set(): a second detector in PERCENT_OF_MEAN mode
		COMPV_CHECK_CODE_RETURN(CompVEdgeDete::newObj(&ptrCannyMean, COMPV_CANNY_ID));
		COMPV_CHECK_CODE_RETURN(ptrCannyMean->setInt(COMPV_CANNY_SET_INT_THRESHOLD_TYPE, COMPV_CANNY_THRESHOLD_TYPE_PERCENT_OF_MEAN));
		COMPV_CHECK_CODE_RETURN(ptrCannyMean->setFloat32(COMPV_CANNY_SET_FLT32_THRESHOLD_LOW, 0.68f));
		COMPV_CHECK_CODE_RETURN(ptrCannyMean->setFloat32(COMPV_CANNY_SET_FLT32_THRESHOLD_HIGH, 1.36f));
	}
	COMPV_CHECK_CODE_RETURN(image->clone(&mat));

	const auto t0 = std::chrono::steady_clock::now();
	COMPV_CHECK_CODE_RETURN(ptrSobel->process(image, &sob));
	// samples/edges_canny/main.cxx:68-72: process(mat, &mat) IN PLACE
	COMPV_CHECK_CODE_RETURN(ptrCanny->process(mat, &mat));
	// samples/hough_lines/main.cxx:102-109
	COMPV_CHECK_CODE_RETURN(ptrHough->process(mat, linesPolar));
	COMPV_CHECK_CODE_RETURN(ptrHough->toCartesian(mat->cols(), mat->rows(), linesPolar, linesCartesian));

	// samples/hough_lines/main.cxx:59-72 with HOUGH_ID == COMPV_HOUGHKHT_ID: newObj(rho, theta, HOUGHKHT_THRESHOLD) + the three KHT knobs
	{
		CompVHoughLineVector khtPolar;
		CompVLineFloat32Vector khtCartesian;
		COMPV_CHECK_CODE_RETURN(ptrKht->process(mat, khtPolar));
		COMPV_CHECK_CODE_RETURN(ptrKht->toCartesian(mat->cols(), mat->rows(), khtPolar, khtCartesian));
		compv_float64_t gs = 0;
		COMPV_CHECK_CODE_RETURN(ptrKht->getFloat64(COMPV_HOUGHKHT_GET_FLT64_GS, &gs));
		r.khtGS = gs;
		r.khtLines.clear(); r.khtCart.clear();
		for (size_t i = 0; i < khtPolar.size(); ++i) {
			r.khtLines.push_back(std::make_tuple(khtPolar[i].rho, khtPolar[i].theta, khtPolar[i].strength));
			r.khtCart.push_back(khtCartesian[i].a.x); r.khtCart.push_back(khtCartesian[i].a.y); r.khtCart.push_back(khtCartesian[i].b.x); r.khtCart.push_back(khtCartesian[i].b.y);
		}
	}

	COMPV_CHECK_CODE_RETURN(ptrCannyMean->process(image, &edges));
	r.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

	auto grab = [&](const CompVMatPtr& m, std::vector<uint8_t>& out) {
		out.resize(W * H);
		for (size_t j = 0; j < H; ++j) memcpy(&out[j * W], m->ptr<const uint8_t>(j), W);
	};
	grab(sob, r.sobel); grab(mat, r.canny); grab(edges, r.cannyMean);
	r.lines.clear();
	for (size_t i = 0; i < linesPolar.size(); ++i) r.lines.push_back(std::make_tuple(linesPolar[i].rho, linesPolar[i].theta, linesPolar[i].strength));
	std::sort(r.lines.begin(), r.lines.end());
	r.cart.clear();
	{
		// cartesian endpoints follow the polar order: canonicalise through the same sort key
		std::vector<std::tuple<float, float, size_t, float, float, float, float> > c;
		for (size_t i = 0; i < linesPolar.size(); ++i)
			c.push_back(std::make_tuple(linesPolar[i].rho, linesPolar[i].theta, linesPolar[i].strength, linesCartesian[i].a.x, linesCartesian[i].a.y, linesCartesian[i].b.x, linesCartesian[i].b.y));
		std::sort(c.begin(), c.end());
		for (auto& t : c) { r.cart.push_back(std::get<3>(t)); r.cart.push_back(std::get<4>(t)); r.cart.push_back(std::get<5>(t)); r.cart.push_back(std::get<6>(t)); }
	}
	// error behaviour is part of the contract
	CompVEdgeDetePtr bad;
	COMPV_CHECK_CODE_RETURN(CompVEdgeDete::newObj(&bad, COMPV_CANNY_ID, 100.f, 50.f, 3));
	CompVMatPtr tmp;
	const COMPV_ERROR_CODE e = bad->process(image, &tmp);
	COMPV_CHECK_EXP_RETURN(e != COMPV_ERROR_CODE_E_INVALID_STATE, COMPV_ERROR_CODE_E_UNITTEST_FAILED, "tLow >= tHigh must give E_INVALID_STATE");
	CompVHoughPtr badH;
	COMPV_CHECK_EXP_RETURN(COMPV_ERROR_CODE_IS_OK(CompVHough::newObj(&badH, COMPV_HOUGHSHT_ID, 0.5f, 1.f, 100)), COMPV_ERROR_CODE_E_UNITTEST_FAILED, "SHT must reject rho != 1");
	return COMPV_ERROR_CODE_S_OK;
}

// A CONSUMER of the line set: CompVCalibCamera builds its Canny and Hough objects through the same factories
// (core/calib/compv_core_calib_camera.cxx:1255-1279: SHT, theta = 0.5 deg, maxLines = 60 per pattern line, Canny(1.33, 2.66)) and runs
// Canny -> SHT -> toCartesian -> line subdivision / grouping -> intersections on a chessboard view (:127-..).  Same application
// code against whatever factories are registered; every stage output is kept for the comparison.
struct CalibResult {
	int code; size_t rawLines, groupedLines;
	std::vector<uint8_t> edges;
	std::vector<std::tuple<float, float, size_t> > raw;          // canonical order
	std::vector<float> grouped, corners;                          // in the order the calibration produced them
};
static COMPV_ERROR_CODE runCalibration(size_t W, size_t H, CalibResult& r)
{
	CompVMatPtr image;
	COMPV_CHECK_CODE_RETURN(CompVImage::newObj8u(&image, COMPV_SUBTYPE_PIXELS_Y, W, H));
	// chessboard of COMPV_CALIB_PATTERN_ROWS_COUNT x COMPV_CALIB_PATTERN_COLS_COUNT squares, slightly sheared, on a light background
	const size_t rows = COMPV_CALIB_PATTERN_ROWS_COUNT, cols = COMPV_CALIB_PATTERN_COLS_COUNT;
	const double sq = (double)(H * 8 / 10) / (double)rows;
	const double ox = ((double)W - sq * cols) * 0.5, oy = (double)H * 0.1;
	for (size_t j = 0; j < H; ++j) {
		uint8_t* p = image->ptr<uint8_t>(j);
		for (size_t i = 0; i < W; ++i) {
			const double y = ((double)j - oy), x = ((double)i - ox) - 0.04 * y; // shear: the vertical lines lean by 2.3 deg
			uint8_t v = 200;
			if (x >= 0 && y >= 0 && x < sq * cols && y < sq * rows) v = ((((size_t)(x / sq)) + ((size_t)(y / sq))) & 1) ? 230 : 25;
			p[i] = v;
		}
	}
	CompVCalibCameraPtr calib;
	COMPV_CHECK_CODE_RETURN(CompVCalibCamera::newObj(&calib));
	CompVCalibContex ctx;
	COMPV_CHECK_CODE_RETURN(calib->process(image, ctx));
	r.code = (int)ctx.code;
	r.rawLines = ctx.lines_raw.lines_hough.size(); r.groupedLines = ctx.lines_grouped.lines_cartesian.size();
	r.edges.clear();
	if (ctx.edges) { r.edges.resize(W * H); for (size_t j = 0; j < H; ++j) memcpy(&r.edges[j * W], ctx.edges->ptr<const uint8_t>(j), W); }
	r.raw.clear();
	for (const auto& l : ctx.lines_raw.lines_hough) r.raw.push_back(std::make_tuple(l.rho, l.theta, l.strength));
	std::sort(r.raw.begin(), r.raw.end());
	r.grouped.clear();
	for (const auto& l : ctx.lines_grouped.lines_cartesian) { r.grouped.push_back(l.a.x); r.grouped.push_back(l.a.y); r.grouped.push_back(l.b.x); r.grouped.push_back(l.b.y); }
	r.corners.clear();
	for (const auto& q : ctx.plane_curr.intersections) { r.corners.push_back(q.x); r.corners.push_back(q.y); }
	return COMPV_ERROR_CODE_S_OK;
}


// The option contract of the detector classes: the same set() / get() calls -- accepted and refused ones (wrong value size,
// out-of-range value, unknown id) -- against whatever factories are registered; the return codes are compared CPU vs HIP.
static std::vector<int> probeOptions()
{
	std::vector<int> codes;
	CompVEdgeDetePtr canny; CompVHoughPtr sht, kht;
	codes.push_back(CompVEdgeDete::newObj(&canny, COMPV_CANNY_ID, 10.f, 20.f, 3));
	codes.push_back(CompVHough::newObj(&sht, COMPV_HOUGHSHT_ID, 1.f, 1.f, 10));
	codes.push_back(CompVHough::newObj(&kht, COMPV_HOUGHKHT_ID, 0.5f, 1.f, 10));
	CompVHoughPtr none;
	codes.push_back(CompVHough::newObj(&none, COMPV_HOUGHKHT_ID, 1.5f, 1.f, 10));   // rho > 1
	codes.push_back(CompVHough::newObj(&none, COMPV_HOUGHKHT_ID, 0.f, 1.f, 10));    // rho <= 0
	if (!canny || !sht || !kht) return codes;
	const float fpos = 0.5f, fneg = -1.f, fzero = 0.f, fbig = 2.f, fone = 1.f;
	const int ipos = 7, izero = 0, ineg = -3, k3 = 3, k5 = 5, k4 = 4, tMean = COMPV_CANNY_THRESHOLD_TYPE_PERCENT_OF_MEAN, tBad = 12345;
	const bool b = true; const double d = 1.0; const int64_t wide = 1;
	struct Probe { int id; const void* p; size_t n; };
	const Probe cannyProbes[] = {
		{ COMPV_CANNY_SET_INT_THRESHOLD_TYPE, &tMean, sizeof(int) }, { COMPV_CANNY_SET_INT_THRESHOLD_TYPE, &tBad, sizeof(int) }, { COMPV_CANNY_SET_INT_THRESHOLD_TYPE, &wide, sizeof(wide) },
		{ COMPV_CANNY_SET_FLT32_THRESHOLD_LOW, &fpos, sizeof(float) }, { COMPV_CANNY_SET_FLT32_THRESHOLD_LOW, &fzero, sizeof(float) }, { COMPV_CANNY_SET_FLT32_THRESHOLD_LOW, &d, sizeof(d) },
		{ COMPV_CANNY_SET_FLT32_THRESHOLD_HIGH, &fbig, sizeof(float) }, { COMPV_CANNY_SET_FLT32_THRESHOLD_HIGH, &fneg, sizeof(float) },
		{ COMPV_CANNY_SET_INT_KERNEL_SIZE, &k3, sizeof(int) }, { COMPV_CANNY_SET_INT_KERNEL_SIZE, &k5, sizeof(int) }, { COMPV_CANNY_SET_INT_KERNEL_SIZE, &k4, sizeof(int) },
		{ COMPV_CANNY_SET_INT_KERNEL_SIZE, NULL, sizeof(int) }, { COMPV_CANNY_SET_INT_KERNEL_SIZE, &k3, 0 },
	};
	for (const Probe& q : cannyProbes) codes.push_back(canny->set(q.id, q.p, q.n));
	const Probe houghProbes[] = {
		{ COMPV_HOUGH_SET_FLT32_RHO, &fone, sizeof(float) }, { COMPV_HOUGH_SET_FLT32_RHO, &fpos, sizeof(float) }, { COMPV_HOUGH_SET_FLT32_RHO, &fbig, sizeof(float) }, { COMPV_HOUGH_SET_FLT32_RHO, &fzero, sizeof(float) },
		{ COMPV_HOUGH_SET_FLT32_THETA, &fpos, sizeof(float) }, { COMPV_HOUGH_SET_FLT32_THETA, &fneg, sizeof(float) }, { COMPV_HOUGH_SET_FLT32_THETA, &d, sizeof(d) },
		{ COMPV_HOUGH_SET_INT_THRESHOLD, &ipos, sizeof(int) }, { COMPV_HOUGH_SET_INT_THRESHOLD, &izero, sizeof(int) }, { COMPV_HOUGH_SET_INT_THRESHOLD, &wide, sizeof(wide) },
		{ COMPV_HOUGH_SET_INT_MAXLINES, &ipos, sizeof(int) }, { COMPV_HOUGH_SET_INT_MAXLINES, &ineg, sizeof(int) }, { COMPV_HOUGH_SET_INT_MAXLINES, &b, sizeof(b) },
		{ COMPV_HOUGHKHT_SET_FLT32_CLUSTER_MIN_DEVIATION, &fpos, sizeof(float) }, { COMPV_HOUGHKHT_SET_FLT32_CLUSTER_MIN_DEVIATION, &fzero, sizeof(float) },
		{ COMPV_HOUGHKHT_SET_INT_CLUSTER_MIN_SIZE, &ipos, sizeof(int) }, { COMPV_HOUGHKHT_SET_INT_CLUSTER_MIN_SIZE, &izero, sizeof(int) },
		{ COMPV_HOUGHKHT_SET_FLT32_KERNEL_MIN_HEIGTH, &fpos, sizeof(float) }, { COMPV_HOUGHKHT_SET_FLT32_KERNEL_MIN_HEIGTH, &fneg, sizeof(float) },
		{ COMPV_HOUGHKHT_SET_BOOL_OVERRIDE_INPUT_EDGES, &b, sizeof(b) }, { COMPV_HOUGHKHT_SET_BOOL_OVERRIDE_INPUT_EDGES, &ipos, sizeof(int) },
		{ 987654, &ipos, sizeof(int) },
	};
	for (const Probe& q : houghProbes) { codes.push_back(sht->set(q.id, q.p, q.n)); codes.push_back(kht->set(q.id, q.p, q.n)); }
	compv_float64_t gs = -1; float notDouble = 0.f;
	codes.push_back(kht->getFloat64(COMPV_HOUGHKHT_GET_FLT64_GS, &gs)); codes.push_back(gs == 1.0 ? 0 : 1);
	const void* pp = &notDouble;
	codes.push_back(kht->get(COMPV_HOUGHKHT_GET_FLT64_GS, &pp, sizeof(float)));
	codes.push_back(kht->get(424242, &pp, sizeof(compv_float64_t)));
	CompVLineFloat32Vector cart; CompVHoughLineVector polar;
	codes.push_back(sht->toCartesian(0, 10, polar, cart)); codes.push_back(kht->toCartesian(10, 0, polar, cart)); codes.push_back(sht->toCartesian(10, 10, polar, cart));
	return codes;
}

// samples/hough_lines/main.cxx:102-105 on a packed RGB24 camera frame: convertGrayscale -> thresholdOtsu -> Canny thresholds.
// CompVImage::convertGrayscale / thresholdOtsu are static functions (no factory to swap), so the HIP side is the C ABI a
// maintainer would call from inside them: both results must be identical.
static int checkPreproc(size_t W, size_t H, uint32_t seed)
{
	std::vector<uint8_t> rgb(W * H * 3);
	uint32_t s = seed;
	for (size_t j = 0; j < H; ++j)
		for (size_t i = 0; i < W; ++i) {
			s = s * 1664525u + 1013904223u;
			const uint32_t v = 40u + ((((uint32_t)(i / 64) + (uint32_t)(j / 64)) & 1u) * 150u) + (s >> 28);
			uint8_t* p = &rgb[(j * W + i) * 3];
			p[0] = (uint8_t)v; p[1] = (uint8_t)(255u - v); p[2] = (uint8_t)((v * 3u) >> 2);
		}
	CompVMatPtr image, gray;
	double tCpu = -1.0;
	if (COMPV_ERROR_CODE_IS_NOK(CompVImage::wrap(COMPV_SUBTYPE_PIXELS_RGB24, rgb.data(), W, H, W, &image))) return 1;
	if (COMPV_ERROR_CODE_IS_NOK(CompVImage::convertGrayscale(image, &gray))) return 1;
	if (COMPV_ERROR_CODE_IS_NOK(CompVImage::thresholdOtsu(gray, tCpu))) return 1;
	compvhip_ctx* ctx = nullptr;
	if (compvhip_ctx_create(&ctx, -1) != COMPVHIP_OK) return 2;
	std::vector<uint8_t> g(W * H);
	double tHip = -2.0;
	int rc = compvhip_grayscale_u8(ctx, rgb.data(), COMPVHIP_FMT_RGB24, W, H, W, g.data(), W);
	if (rc == COMPVHIP_OK) rc = compvhip_otsu_u8(ctx, g.data(), W, H, W, &tHip);
	compvhip_ctx_destroy(ctx);
	if (rc != COMPVHIP_OK) return 2;
	bool same = true;
	for (size_t j = 0; j < H && same; ++j) same = memcmp(gray->ptr<const uint8_t>(j), &g[j * W], W) == 0;
	printf("pre-processing (%zux%zu RGB24): grayscale %s, Otsu %s [CompV %.0f, HIP %.0f]\n", W, H, same ? "==" : "DIFF", tCpu == tHip ? "==" : "DIFF", tCpu, tHip);
	return (same && tCpu == tHip) ? 0 : 3;
}

int main(int argc, char** argv)
{
	const size_t W = argc > 2 ? (size_t)atoi(argv[1]) : 1280, H = argc > 2 ? (size_t)atoi(argv[2]) : 720;
	const int frames = argc > 3 ? atoi(argv[3]) : 2;
	const int cpuThreads = argc > 4 ? atoi(argv[4]) : 1;
	CompVDebugMgr::setLevel(COMPV_DEBUG_LEVEL_ERROR);
	// CompVInit() of compv_api.h minus GL/camera/drawing (absent on a headless box): base + core
	if (COMPV_ERROR_CODE_IS_NOK(CompVBase::init(cpuThreads)) || COMPV_ERROR_CODE_IS_NOK(CompVCore::init())) { fprintf(stderr, "CompV init failed\n"); return 2; }

	std::vector<Result> cpu(frames), gpu(frames);
	for (int f = 0; f < frames; ++f) {
		if (COMPV_ERROR_CODE_IS_NOK(runSamples(W, H, 12345u + f, cpu[f]))) { fprintf(stderr, "CPU run failed\n"); return 3; }
	}
	resetObjects();
	CalibResult calCpu, calHip;
	if (COMPV_ERROR_CODE_IS_NOK(runCalibration(1280, 720, calCpu))) { fprintf(stderr, "CPU calibration run failed\n"); return 6; }
	const std::vector<int> optCpu = probeOptions();
	if (compv_hip_plugin_register() != 0) { fprintf(stderr, "HIP plugin registration failed (no GPU?)\n"); return 4; }
	const std::vector<int> optHip = probeOptions();
	if (COMPV_ERROR_CODE_IS_NOK(runCalibration(1280, 720, calHip))) { fprintf(stderr, "HIP calibration run failed\n"); return 7; }
	for (int f = 0; f < frames; ++f) {
		if (COMPV_ERROR_CODE_IS_NOK(runSamples(W, H, 12345u + f, gpu[f]))) { fprintf(stderr, "HIP run failed\n"); return 5; }
	}
	int bad = 0;
	for (int f = 0; f < frames; ++f) {
		const bool okS = cpu[f].sobel == gpu[f].sobel, okC = cpu[f].canny == gpu[f].canny, okM = cpu[f].cannyMean == gpu[f].cannyMean;
		const bool okL = cpu[f].lines == gpu[f].lines, okX = cpu[f].cart == gpu[f].cart;
		const bool okK = cpu[f].khtLines == gpu[f].khtLines && cpu[f].khtCart == gpu[f].khtCart && cpu[f].khtGS == gpu[f].khtGS;
		size_t e = 0; for (uint8_t v : cpu[f].canny) e += v ? 1 : 0;
		printf("frame %d (%zux%zu): sobel %s, canny(in-place) %s [%zu edge px], canny(mean mode) %s, hough lines %s [%zu], cartesian %s, KHT lines+order+GS %s [%zu] | CompV CPU %.2f ms, HIP plugin %.2f ms (incl. H2D/D2H)\n",
			f, W, H, okS ? "==" : "DIFF", okC ? "==" : "DIFF", e, okM ? "==" : "DIFF", okL ? "==" : "DIFF", cpu[f].lines.size(), okX ? "==" : "DIFF", okK ? "==" : "DIFF", cpu[f].khtLines.size(), cpu[f].ms, gpu[f].ms);
		bad += !(okS && okC && okM && okL && okX && okK);
	}
	resetObjects();
	{
		size_t diff = 0;
		for (size_t i = 0; i < optCpu.size() && i < optHip.size(); ++i) if (optCpu[i] != optHip[i]) { ++diff; printf("  option probe %zu: CompV %d, HIP %d\n", i, optCpu[i], optHip[i]); }
		diff += optCpu.size() != optHip.size();
		printf("option contract (%zu set/get/newObj/toCartesian probes): %s\n", optCpu.size(), diff ? "DIFF" : "==");
		bad += diff != 0;
	}
	{
		const bool okE = calCpu.edges == calHip.edges, okR = calCpu.raw == calHip.raw, okG = calCpu.grouped == calHip.grouped;
		// The stages up to the corner intersections are deterministic functions of the detectors' outputs and are compared exactly.  The
		// result code is compared up to the homography: that stage is CompV's own RANSAC with rand() sampling (compv_core_calib_camera.cxx:460-466)
		// and may end in OK or NO_ENOUGH_INLIERS on identical corners from one run to the next.
		auto reachedHomography = [](int c) { return c == (int)COMPV_CALIB_CAMERA_RESULT_OK || c == (int)COMPV_CALIB_CAMERA_RESULT_NO_ENOUGH_INLIERS; };
		const bool okC = calCpu.corners == calHip.corners && (calCpu.code == calHip.code || (reachedHomography(calCpu.code) && reachedHomography(calHip.code)));
		printf("calibration client (CompVCalibCamera::process, 1280x720 chessboard): edges %s, raw lines %s [%zu], grouped lines %s [%zu], result code %d/%d + %zu corners %s\n",
			okE ? "==" : "DIFF", okR ? "==" : "DIFF", calCpu.rawLines, okG ? "==" : "DIFF", calCpu.groupedLines, calCpu.code, calHip.code, calCpu.corners.size() / 2, okC ? "==" : "DIFF");
		bad += !(okE && okR && okG && okC);
	}
	bad += checkPreproc(W, H, 4242u) != 0;
	printf(bad ? "DROP-IN PARITY FAILED\n" : "DROP-IN PARITY OK\n");
	return bad ? 1 : 0;
}
