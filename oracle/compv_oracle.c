/* TEST INFRASTRUCTURE ONLY -- see compv_oracle.h.  Plain scalar C restatement of the CompV hot path.
 * Each function cites the reference file:line (relative to /root/reference) whose behaviour it restates.
 * Nothing here is copied from the reference: the code is written from the verified semantic spec
 *  and checked against the compiled reference (oracle/_ref) by tests/.
 */
#include "compv_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* synthetic frame: SURVEY.md 8(d)                                                                  */
/* ------------------------------------------------------------------------------------------------ */
void orc_synth_frame(uint8_t* out, size_t W, size_t H, size_t S, uint32_t seed)
{
	uint32_t s = seed;
	for (size_t j = 0; j < H; ++j) {
		for (size_t i = 0; i < W; ++i) {
			s = s * 1664525u + 1013904223u;
			uint32_t v = 40u + ((((uint32_t)(i / 64) + (uint32_t)(j / 64)) & 1u) * 150u) + (s >> 28);
			if (((i + 2 * j) % 257) < 3) v = 255u;
			out[j * S + i] = (uint8_t)v;
		}
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* separable correlation                                                                            */
/* base/include/compv/base/math/compv_math_convlt.h:98-173 (driver), :176-229 (hz: zero cols [0,r)   */
/* and [W-r,W), correlate the rest), :232-292 (vt: zero rows [0,r) and [H-r,H), correlate the rest); */
/* leaf base/math/intrin/x86/compv_math_convlt_intrin_avx2.cxx:314-430 (int32 sum, packs = saturate). */
/* ------------------------------------------------------------------------------------------------ */
static inline int16_t sat16(int32_t v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

static void hz_pass_8u(const uint8_t* in, size_t W, size_t H, size_t S, const int16_t* kern, size_t k, int16_t* out)
{
	const size_t r = k >> 1;
	for (size_t y = 0; y < H; ++y) {
		for (size_t x = 0; x < W; ++x) {
			if (x < r || x >= W - r) { out[y * S + x] = 0; continue; }
			int32_t sum = 0;
			for (size_t t = 0; t < k; ++t) sum += (int32_t)in[y * S + x - r + t] * (int32_t)kern[t];
			out[y * S + x] = sat16(sum);
		}
	}
}
static void hz_pass_16s(const int16_t* in, size_t W, size_t H, size_t S, const int16_t* kern, size_t k, int16_t* out)
{
	const size_t r = k >> 1;
	for (size_t y = 0; y < H; ++y) {
		for (size_t x = 0; x < W; ++x) {
			if (x < r || x >= W - r) { out[y * S + x] = 0; continue; }
			int32_t sum = 0;
			for (size_t t = 0; t < k; ++t) sum += (int32_t)in[y * S + x - r + t] * (int32_t)kern[t];
			out[y * S + x] = sat16(sum);
		}
	}
}
static void vt_pass_16s(const int16_t* in, size_t W, size_t H, size_t S, const int16_t* kern, size_t k, int16_t* out)
{
	const size_t r = k >> 1;
	for (size_t y = 0; y < H; ++y) {
		for (size_t x = 0; x < W; ++x) {
			if (y < r || y >= H - r) { out[y * S + x] = 0; continue; }
			int32_t sum = 0;
			for (size_t t = 0; t < k; ++t) sum += (int32_t)in[(y - r + t) * S + x] * (int32_t)kern[t];
			out[y * S + x] = sat16(sum);
		}
	}
}

int orc_convlt1_8u16s16s(const uint8_t* in, size_t W, size_t H, size_t S, const int16_t* vt, const int16_t* hz, size_t k, int16_t* out)
{
	/* argument check: compv_math_convlt.h:100 */
	if (!in || W < k || H < k || S < W || !vt || !hz || !(k & 1) || !out) return ORC_E_INVALID_PARAMETER;
	int16_t* tmp = (int16_t*)calloc(S * H, sizeof(int16_t));
	if (!tmp) return ORC_E_OUT_OF_MEMORY;
	hz_pass_8u(in, W, H, S, hz, k, tmp);
	vt_pass_16s(tmp, W, H, S, vt, k, out);
	free(tmp);
	return ORC_OK;
}
int orc_convlt1_16s16s16s(const int16_t* in, size_t W, size_t H, size_t S, const int16_t* vt, const int16_t* hz, size_t k, int16_t* out)
{
	if (!in || W < k || H < k || S < W || !vt || !hz || !(k & 1) || !out) return ORC_E_INVALID_PARAMETER;
	int16_t* tmp = (int16_t*)calloc(S * H, sizeof(int16_t));
	if (!tmp) return ORC_E_OUT_OF_MEMORY;
	hz_pass_16s(in, W, H, S, hz, k, tmp);
	vt_pass_16s(tmp, W, H, S, vt, k, out);
	free(tmp);
	return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* gradient: kernels base/include/compv/base/compv_features.h:124-133; call pattern                  */
/* core/features/edges/compv_core_feature_canny_dete.cxx:237-241 (Gx: vt=Gx_vt, hz=Gx_hz; Gy swapped)*/
/* L1 magnitude base/math/intrin/x86/compv_math_utils_intrin_avx2.cxx:21-48 (adds_epu16 of abs).     */
/* ------------------------------------------------------------------------------------------------ */
static const int16_t kSobel3_vt[3] = { 1, 2, 1 }, kSobel3_hz[3] = { -1, 0, 1 };
static const int16_t kSobel5_vt[5] = { 1, 4, 6, 4, 1 }, kSobel5_hz[5] = { 1, 2, 0, -2, -1 };
static const int16_t kScharr_vt[3] = { 3, 10, 3 }, kScharr_hz[3] = { -1, 0, 1 };
static const int16_t kPrewitt_vt[3] = { 1, 1, 1 }, kPrewitt_hz[3] = { -1, 0, 1 };

static int op_kernels(int op, const int16_t** vt, const int16_t** hz, size_t* k)
{
	switch (op) {
	case ORC_OP_SOBEL3: *vt = kSobel3_vt; *hz = kSobel3_hz; *k = 3; return 0;
	case ORC_OP_SOBEL5: *vt = kSobel5_vt; *hz = kSobel5_hz; *k = 5; return 0;
	case ORC_OP_SCHARR: *vt = kScharr_vt; *hz = kScharr_hz; *k = 3; return 0;
	case ORC_OP_PREWITT: *vt = kPrewitt_vt; *hz = kPrewitt_hz; *k = 3; return 0;
	default: return -1;
	}
}

int orc_gradient(const uint8_t* in, size_t W, size_t H, size_t S, int op, int16_t* gx, int16_t* gy, uint16_t* g)
{
	const int16_t *vt, *hz; size_t k;
	if (op_kernels(op, &vt, &hz, &k)) return ORC_E_INVALID_PARAMETER;
	int err;
	if ((err = orc_convlt1_8u16s16s(in, W, H, S, vt, hz, k, gx))) return err;
	if ((err = orc_convlt1_8u16s16s(in, W, H, S, hz, vt, k, gy))) return err;
	for (size_t y = 0; y < H; ++y) {
		for (size_t x = 0; x < W; ++x) {
			const int32_t ax = abs((int32_t)gx[y * S + x]), ay = abs((int32_t)gy[y * S + x]);
			const int32_t s = ax + ay;
			g[y * S + x] = (uint16_t)(s > 65535 ? 65535 : s);
		}
	}
	return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* Sobel / Scharr / Prewitt detector                                                                 */
/* core/features/edges/compv_core_feature_edge_dete.cxx:55-206 (single-thread branch :188-203):       */
/*   gmax = max(g) through CompVMathUtilsMax_16u_Intrin_SSE41, whose final horizontal reduce only    */
/*   folds lanes {0,1,2,4} of the 8 column-lane maxima (quirk Q1,                                    */
/*   base/math/intrin/x86/compv_math_utils_intrin_sse41.cxx:55-63); scale = 255.f/float(gmax);        */
/*   out = sat_u8(trunc(float(g)*scale)) (base/math/intrin/x86/compv_math_utils_intrin_sse2.cxx:165). */
/*   gmax == 0 (single-thread branch overwrites the initial 1) -> scale = inf -> every pixel 0.       */
/* ------------------------------------------------------------------------------------------------ */
int orc_edge_dete(const uint8_t* in, size_t W, size_t H, size_t S, int op, uint8_t* out, size_t So, uint16_t* gmax_out)
{
	if (!in || !out || So < W) return ORC_E_INVALID_PARAMETER;
	int16_t* gx = (int16_t*)calloc(S * H, 2);
	int16_t* gy = (int16_t*)calloc(S * H, 2);
	uint16_t* g = (uint16_t*)calloc(S * H, 2);
	if (!gx || !gy || !g) { free(gx); free(gy); free(g); return ORC_E_OUT_OF_MEMORY; }
	int err = orc_gradient(in, W, H, S, op, gx, gy, g);
	if (!err) {
		uint16_t gmax = 0;
		for (size_t y = 0; y < H; ++y) {
			for (size_t x = 0; x < W; ++x) {
				const size_t lane = x & 7;
				if ((lane == 0 || lane == 1 || lane == 2 || lane == 4) && g[y * S + x] > gmax) gmax = g[y * S + x];
			}
		}
		if (gmax_out) *gmax_out = gmax;
		if (gmax == 0) {
			for (size_t y = 0; y < H; ++y) memset(out + y * So, 0, W);
		}
		else {
			const volatile float scale = 255.f / (float)gmax;
			for (size_t y = 0; y < H; ++y) {
				for (size_t x = 0; x < W; ++x) {
					const volatile float p = (float)g[y * S + x] * scale; /* one correctly-rounded f32 multiply */
					const int32_t v = (int32_t)p; /* cvttps: truncation */
					out[y * So + x] = (uint8_t)(v > 255 ? 255 : (v < 0 ? 0 : v));
				}
			}
		}
	}
	free(gx); free(gy); free(g);
	return err;
}

/* ------------------------------------------------------------------------------------------------ */
/* Canny                                                                                             */
/* ------------------------------------------------------------------------------------------------ */
/* thresholds: core/features/edges/compv_core_feature_canny_dete.cxx:251-266 */
int orc_canny_thresholds(float fLow, float fHigh, int type, uint32_t sum, size_t W, size_t H, uint16_t* tLow, uint16_t* tHigh)
{
	if (fLow >= fHigh) return ORC_E_INVALID_STATE; /* :126 */
	uint16_t lo, hi;
	if (type == ORC_THRESHOLD_PERCENT_OF_MEAN) {
		uint8_t mean = (uint8_t)(sum / (uint32_t)(W * H));
		mean = (uint8_t)(mean < 1 ? 1 : mean); /* CLIP3(1,255,mean) on a uint8 */
		lo = (uint16_t)((float)mean * fLow);
		hi = (uint16_t)((float)mean * fHigh);
	}
	else if (type == ORC_THRESHOLD_COMPARE_TO_GRADIENT) {
		const float l = fLow < 1.f ? 1.f : (fLow > 65535.f ? 65535.f : fLow);
		const float h = fHigh < 1.f ? 1.f : (fHigh > 65535.f ? 65535.f : fHigh);
		lo = (uint16_t)l;
		hi = (uint16_t)h;
	}
	else return ORC_E_INVALID_PARAMETER;
	/* tLow = max(1,tLow); tHigh = max(tLow+2,tHigh) -- int arithmetic, then stored to uint16 */
	lo = (uint16_t)(lo < 1 ? 1 : lo);
	{
		const int t = (int)lo + 2;
		hi = (uint16_t)(t > (int)hi ? t : (int)hi);
	}
	*tLow = lo; *tHigh = hi;
	return ORC_OK;
}

/* Column coverage of nms_gather and of the hysteresis seed scan (quirk Q3).
 * nms_gather: ...canny_dete.cxx:334-412 -- SIMD row function covers col = 1, 1+mpw, ... while col < (W-1)-(mpw-1)
 * (AVX2 16mpw when W-1 >= 16: intrin_avx2.cxx:166; SSSE3 8mpw when W-1 >= 8: intrin_ssse3.cxx:32), then the scalar
 * row function restarts at colStart = (W-1) & -(mpw-1)  (:396) -- *not* a multiple-of-mpw mask, it clears bits 1..3
 * (mpw=16) or 1..2 (mpw=8) -- and runs to W-1.  hysteresis(): same formula (:497-514, SSE2 16mpw/8mpw). */
void orc_canny_coverage(size_t W, size_t* simdEnd, size_t* cStart)
{
	const size_t maxCols = W - 1;
	size_t mpw = 1;
	if (maxCols >= 16) mpw = 16; else if (maxCols >= 8) mpw = 8;
	if (mpw == 1) { *simdEnd = 1; *cStart = 1; return; }
	size_t col = 1;
	while (col + (mpw - 1) < maxCols) col += mpw; /* col < maxCols - (mpw-1) */
	*simdEnd = col;
	*cStart = maxCols & (size_t)(-(ptrdiff_t)(mpw - 1));
}

int orc_canny(const uint8_t* in, size_t W, size_t H, size_t S, float fLow, float fHigh, int ksize, int type,
              uint8_t* out, size_t So, uint16_t* gnms)
{
	if (!in || !out || So < W || (ksize != 3 && ksize != 5)) return ORC_E_INVALID_PARAMETER;
	if (fLow >= fHigh) return ORC_E_INVALID_STATE;
	const size_t n = S * H;
	int16_t* gx = (int16_t*)calloc(n, 2);
	int16_t* gy = (int16_t*)calloc(n, 2);
	uint16_t* g = (uint16_t*)calloc(n, 2);
	uint8_t* nms = (uint8_t*)calloc(n, 1);
	uint32_t* stack = (uint32_t*)malloc((W * H + 8) * sizeof(uint32_t));
	int err = ORC_OK;
	if (!gx || !gy || !g || !nms || !stack) { err = ORC_E_OUT_OF_MEMORY; goto done; }
	if ((err = orc_gradient(in, W, H, S, ksize == 3 ? ORC_OP_SOBEL3 : ORC_OP_SOBEL5, gx, gy, g))) goto done;

	uint32_t sum = 0;
	if (type == ORC_THRESHOLD_PERCENT_OF_MEAN) { /* CompVMathUtils::sum<u8,u32>, :243 */
		for (size_t y = 0; y < H; ++y) for (size_t x = 0; x < W; ++x) sum += in[y * S + x];
	}
	uint16_t tLow, tHigh;
	if ((err = orc_canny_thresholds(fLow, fHigh, type, sum, W, H, &tLow, &tHigh))) goto done;
	/* The SIMD leaves compare g against the thresholds as *signed* int16 (intrin_avx2.cxx:144-147, intrin_sse2.cxx:105):
	 * thresholds above 32767 are an artefact regime this restatement (and the HIP path) rejects. */
	if (tHigh > 32767) { err = ORC_E_INVALID_PARAMETER; goto done; }

	size_t simdEnd, cStart;
	orc_canny_coverage(W, &simdEnd, &cStart);
#define COVERED(x) (((x) >= 1 && (x) < simdEnd) || ((x) >= cStart && (x) < W - 1))

	/* NMS gather on the unsuppressed g: ...canny_dete.cxx:566-598 (scalar statement of the rule),
	 * constants core/include/compv/core/features/edges/compv_core_feature_canny_dete.h:58-61.
	 * int32 products wrap exactly like _mm256_mullo_epi32 (only reachable with the 5x5 kernel). */
	for (size_t y = 1; y + 1 < H; ++y) {
		for (size_t x = 1; x + 1 < W; ++x) {
			if (!COVERED(x)) continue;
			const size_t i = y * S + x;
			const uint16_t gc = g[i];
			if (gc <= tLow) continue;
			const int32_t gxi = gx[i], gyi = gy[i];
			const int32_t ay = (int32_t)((uint32_t)abs(gyi) << 16);
			const int32_t ax = abs(gxi);
			const int32_t t1 = (int32_t)(27145u * (uint32_t)ax);
			const int32_t t2 = (int32_t)(158217u * (uint32_t)ax);
			uint16_t n1, n2;
			if (ay < t1) { n1 = g[i - 1]; n2 = g[i + 1]; }
			else if (ay < t2) {
				if ((gxi ^ gyi) < 0) { n1 = g[i - 1 + S]; n2 = g[i + 1 - S]; }
				else { n1 = g[i - 1 - S]; n2 = g[i + 1 + S]; }
			}
			else { n1 = g[i - S]; n2 = g[i + S]; }
			if (n1 > gc || n2 > gc) nms[i] = 0xff;
		}
	}
	/* NMS apply: :414-460 */
	for (size_t i = 0; i < n; ++i) if (nms[i]) g[i] = 0;
	if (gnms) memcpy(gnms, g, n * 2);

	/* hysteresis: :462-528 + :600-680 -- closure of {g>tLow} under 8-connectivity from seeds {g>tHigh} found in the
	 * covered columns of rows 1..H-2; only interior pixels expand (:628). */
	for (size_t y = 0; y < H; ++y) memset(out + y * So, 0, W);
	for (size_t y = 1; y + 1 < H; ++y) {
		for (size_t x = 1; x + 1 < W; ++x) {
			if (!COVERED(x)) continue;
			if (g[y * S + x] > tHigh && !out[y * So + x]) {
				size_t sp = 0;
				out[y * So + x] = 0xff;
				stack[sp++] = (uint32_t)((y << 16) | x);
				while (sp) {
					const uint32_t e = stack[--sp];
					const size_t c = e & 0xffff, r = e >> 16;
					if (!(r && c && r < H - 1 && c < W - 1)) continue;
					for (int dy = -1; dy <= 1; ++dy) {
						for (int dx = -1; dx <= 1; ++dx) {
							if (!dx && !dy) continue;
							const size_t rr = r + dy, cc = c + dx;
							if (g[rr * S + cc] > tLow && !out[rr * So + cc]) {
								out[rr * So + cc] = 0xff;
								stack[sp++] = (uint32_t)((rr << 16) | cc);
							}
						}
					}
				}
			}
		}
	}
#undef COVERED
done:
	free(gx); free(gy); free(g); free(nms); free(stack);
	return err;
}

/* ------------------------------------------------------------------------------------------------ */
/* Hough SHT                                                                                         */
/* ------------------------------------------------------------------------------------------------ */
static const float kPiF = 3.1415926535897932384626433f;  /* kfMathTrigPi, base/math/compv_math.cxx:27 */
#define kPiOver180F (kPiF / 180.f)                         /* kfMathTrigPiOver180, base/math/compv_math.cxx:30 */

/* core/features/hough/compv_core_feature_houghsht.cxx:42-52 (m_fTheta = theta*kfMathTrigPiOver180),
 * :318-348 initCoords: R = round((2(W+H)+1)/rho), T = round(pi/theta) with "+0.5 then truncate". */
int orc_sht_dims(size_t W, size_t H, float thetaDeg, size_t* R, size_t* T, float* thetaRad_out)
{
	if (!W || !H || !(thetaDeg > 0.f)) return ORC_E_INVALID_PARAMETER;
	const float fTheta = thetaDeg * kPiOver180F;
	const float fRho = 1.f;
	*R = (size_t)((double)((float)(((W + H) << 1) + 1) / fRho) + 0.5);
	*T = (size_t)((double)(kPiF / fTheta) + 0.5);
	if (thetaRad_out) *thetaRad_out = fTheta;
	return ORC_OK;
}

/* tables: :335-339 -- float32 running angle, std::sin/std::cos on float (sinf/cosf), (v*rho)*65535.f truncated. */
int orc_sht_tables(float thetaDeg, size_t T, int32_t* sinQ, int32_t* cosQ)
{
	const float fTheta = thetaDeg * kPiOver180F;
	const float fRho = 1.f;
	float tt = 0.f;
	for (size_t t = 0; t < T; ++t, tt += fTheta) {
		sinQ[t] = (int32_t)((sinf(tt) * fRho) * 65535.f);
		cosQ[t] = (int32_t)((cosf(tt) * fRho) * 65535.f);
	}
	return ORC_OK;
}

/* votes: :350-481 + leaf :607-627 -- rho = (col*cosQ[t] + row*sinQ[t]) >> 16 (int32, arithmetic shift),
 * acc[(barrier - rho)*stride + t]++ for every non-zero edge pixel, raster order. */
int orc_sht_acc(const uint8_t* edges, size_t W, size_t H, size_t S, const int32_t* sinQ, const int32_t* cosQ, size_t T,
                int32_t* acc, size_t accStride)
{
	const int32_t barrier = (int32_t)(W + H);
	for (size_t y = 0; y < H; ++y) {
		for (size_t x = 0; x < W; ++x) {
			if (!edges[y * S + x]) continue;
			for (size_t t = 0; t < T; ++t) {
				const int32_t rho = ((int32_t)x * cosQ[t] + (int32_t)y * sinQ[t]) >> 16;
				acc[(size_t)(barrier - rho) * accStride + t]++;
			}
		}
	}
	return ORC_OK;
}

static int line_cmp(const void* a, const void* b)
{
	const orc_line *x = (const orc_line*)a, *y = (const orc_line*)b;
	if (x->strength != y->strength) return x->strength > y->strength ? -1 : 1;
	if (x->row != y->row) return x->row < y->row ? -1 : 1;
	return x->col < y->col ? -1 : (x->col > y->col ? 1 : 0);
}

/* NMS + lines: nms_gather :483-533 with the SSE2 row leaf (intrin_sse2.cxx:16-49): theta columns [1, (T-1)&~3]
 * of rho rows 1..R-2 are suppressed when acc>thr and any 8-neighbour is strictly greater; the scalar remainder call
 * never iterates (quirk Q2), so columns 0 and > (T-1)&~3 are thresholded without NMS.  nms_apply :535-564,:652-668:
 * every non-suppressed acc>thr over ALL rows/cols -> (rho = barrier-row, theta = col*fTheta (f32), strength).
 * The reference then std::sort()s by strength only (unstable, :243-249); ties are ordered here canonically by
 * (row asc, col asc) so that results are comparable. */
int orc_sht_lines(const int32_t* acc, size_t R, size_t T, size_t accStride, int32_t threshold, int32_t barrier, float thetaRad,
                  int maxLines, orc_line* lines, size_t cap, size_t* n)
{
	if (T < 5) return ORC_E_INVALID_PARAMETER; /* the <5-column scalar-only dispatch is not restated */
	const size_t nmsLast = (T - 1) & ~(size_t)3; /* inclusive */
	size_t cnt = 0;
	for (size_t r = 0; r < R; ++r) {
		for (size_t c = 0; c < T; ++c) {
			const int32_t a = acc[r * accStride + c];
			if (a <= threshold) continue;
			int suppressed = 0;
			if (r >= 1 && r + 1 < R && c >= 1 && c <= nmsLast) {
				const int32_t* p = &acc[r * accStride + c];
				const ptrdiff_t s = (ptrdiff_t)accStride;
				suppressed = p[-1] > a || p[1] > a || p[-s - 1] > a || p[-s] > a || p[-s + 1] > a || p[s - 1] > a || p[s] > a || p[s + 1] > a;
			}
			if (suppressed) continue;
			if (cnt < cap) {
				lines[cnt].rho = (float)(barrier - (int32_t)r);
				lines[cnt].theta = (float)c * thetaRad;
				lines[cnt].strength = a;
				lines[cnt].row = (int32_t)r;
				lines[cnt].col = (int32_t)c;
			}
			++cnt;
		}
	}
	const size_t have = cnt < cap ? cnt : cap;
	qsort(lines, have, sizeof(orc_line), line_cmp);
	if (maxLines > 0 && cnt > (size_t)maxLines) cnt = (size_t)maxLines;
	*n = cnt;
	return ORC_OK;
}

int orc_sht(const uint8_t* edges, size_t W, size_t H, size_t S, float thetaDeg, int32_t threshold, int maxLines,
            orc_line* lines, size_t cap, size_t* n)
{
	size_t R, T; float thetaRad;
	int err = orc_sht_dims(W, H, thetaDeg, &R, &T, &thetaRad);
	if (err) return err;
	const size_t accStride = (T + 15) & ~(size_t)15;
	int32_t* sinQ = (int32_t*)calloc(T, 4);
	int32_t* cosQ = (int32_t*)calloc(T, 4);
	int32_t* acc = (int32_t*)calloc(R * accStride, 4);
	if (!sinQ || !cosQ || !acc) { free(sinQ); free(cosQ); free(acc); return ORC_E_OUT_OF_MEMORY; }
	orc_sht_tables(thetaDeg, T, sinQ, cosQ);
	orc_sht_acc(edges, W, H, S, sinQ, cosQ, T, acc, accStride);
	err = orc_sht_lines(acc, R, T, accStride, threshold, (int32_t)(W + H), thetaRad, maxLines, lines, cap, n);
	free(sinQ); free(cosQ); free(acc);
	return err;
}

/* ================================================================================================================
 * Caller-side pre-processing.  Restates:
 *   CompVImageConvToGrayscale::process            base/image/compv_image_conv_to_grayscale.cxx:35-90
 *   rgb24family_to_y_C / rgb32family_to_y_C       base/image/compv_image_conv_rgbfamily.cxx:93-117, 243-268
 *   rgb565family_to_y (macro)                     base/image/compv_image_conv_rgbfamily.cxx:403-426
 *   coefficient tables (RY,GY,BY = 33,65,13)      base/image/compv_image_conv_common.cxx:29-135
 *   yuyv422_to_y_C / uyvy422_to_y_c               base/image/compv_image_conv_to_grayscale.cxx:233-282
 *   CompVImageThreshold::otsu                     base/image/compv_image_threshold.cxx:52-114
 * ================================================================================================================ */
int orc_fmt_bytes(int fmt)
{
	switch (fmt) {
	case ORC_FMT_RGBA32: case ORC_FMT_ARGB32: case ORC_FMT_BGRA32: return 4;
	case ORC_FMT_RGB24: case ORC_FMT_BGR24: return 3;
	case ORC_FMT_RGB565LE: case ORC_FMT_RGB565BE: case ORC_FMT_BGR565LE: case ORC_FMT_BGR565BE: return 2;
	case ORC_FMT_YUYV422: case ORC_FMT_UYVY422: return 2;
	case ORC_FMT_Y: return 1;
	default: return 0;
	}
}

static uint8_t orc_luma(int c0, int c1, int c2, int a, int b, int c)
{
	/* Y = ((c0*a + c1*b + c2*c) >> 7) + 16, clampPixel8 (never clamps: max 237) */
	int v = ((c0 * a + c1 * b + c2 * c) >> 7) + 16;
	return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

int orc_grayscale(const uint8_t* in, int fmt, size_t W, size_t H, size_t S, uint8_t* out, size_t So)
{
	const int bpp = orc_fmt_bytes(fmt);
	if (!bpp || !in || !out || S < W || So < W) return -1;
	const int RY = 33, GY = 65, BY = 13;
	for (size_t j = 0; j < H; ++j) {
		const uint8_t* p = in + j * S * (size_t)bpp;
		uint8_t* o = out + j * So;
		for (size_t i = 0; i < W; ++i, p += bpp) {
			switch (fmt) {
			case ORC_FMT_RGBA32: o[i] = orc_luma(RY, GY, BY, p[0], p[1], p[2]); break;
			case ORC_FMT_ARGB32: o[i] = orc_luma(RY, GY, BY, p[1], p[2], p[3]); break;
			case ORC_FMT_BGRA32: o[i] = orc_luma(BY, GY, RY, p[0], p[1], p[2]); break;
			case ORC_FMT_RGB24: o[i] = orc_luma(RY, GY, BY, p[0], p[1], p[2]); break;
			case ORC_FMT_BGR24: o[i] = orc_luma(BY, GY, RY, p[0], p[1], p[2]); break;
			case ORC_FMT_RGB565LE: case ORC_FMT_RGB565BE: case ORC_FMT_BGR565LE: case ORC_FMT_BGR565BE: {
				const int be = (fmt == ORC_FMT_RGB565BE || fmt == ORC_FMT_BGR565BE);
				const int bgr = (fmt == ORC_FMT_BGR565LE || fmt == ORC_FMT_BGR565BE);
				/* the sample is read as a native (little-endian) u16; big-endian input is byte-swapped first */
				uint16_t k = (uint16_t)(p[0] | (p[1] << 8));
				if (be) k = (uint16_t)((k << 8) | (k >> 8));
				uint16_t r = (uint16_t)((k & 0xF800) >> 8); r |= (uint16_t)(r >> 5);
				uint16_t g = (uint16_t)((k & 0x07E0) >> 3); g |= (uint16_t)(g >> 6);
				uint16_t b = (uint16_t)((k & 0x001F) << 3); b |= (uint16_t)(b >> 5);
				o[i] = bgr ? orc_luma(BY, GY, RY, r, g, b) : orc_luma(RY, GY, BY, r, g, b);
				break;
			}
			case ORC_FMT_YUYV422: o[i] = p[0]; break;
			case ORC_FMT_UYVY422: o[i] = p[1]; break;
			default: o[i] = p[0]; break;
			}
		}
	}
	return 0;
}

void orc_hist256(const uint8_t* in, size_t W, size_t H, size_t S, uint32_t* hist)
{
	memset(hist, 0, 256 * sizeof(uint32_t));
	for (size_t j = 0; j < H; ++j)
		for (size_t i = 0; i < W; ++i) hist[in[j * S + i]]++;
}

int orc_otsu_from_hist(const uint32_t* hist, size_t Npx)
{
	/* compv_image_threshold.cxx:64-105: u32 sums (wrap like the reference), f32 arithmetic in source order */
	uint32_t sumA256[256], sum32 = 0;
	for (int i = 0; i < 256; ++i) { sumA256[i] = (uint32_t)i * hist[i]; sum32 += sumA256[i]; }
	const float sumf = (float)sum32;
	const int N = (int)Npx;
	float sumB = 0.f, varMax = 0.f;
	int q1 = 0, q2 = 0, thr = 0;
	for (int i = 0; i < 256; ++i) {
		q1 += (int)hist[i];
		if (q1) {
			q2 = N - q1;
			if (!q2) break;
			const float q1f = (float)q1, q2f = (float)q2;
			sumB += (float)sumA256[i];
			const float mf = (sumB / q1f) - ((sumf - sumB) / q2f);
			const float varB = q1f * q2f * mf * mf;
			if (varB > varMax) { varMax = varB; thr = i; }
		}
	}
	return thr;
}

int orc_otsu(const uint8_t* in, size_t W, size_t H, size_t S)
{
	uint32_t hist[256];
	orc_hist256(in, W, H, S, hist);
	return orc_otsu_from_hist(hist, W * H);
}

void orc_otsu_canny_thresholds(int t, float fLowFactor, float fHighFactor, int* tLow, int* tHigh)
{
	/* samples/hough_lines/main.cxx:104-105: setFloat32(LOW, static_cast<float>(threshold * 0.5)), HIGH = static_cast<float>(threshold) */
	const float fLow = (float)((double)t * (double)fLowFactor);
	const float fHigh = (float)((double)t * (double)fHighFactor);
	uint16_t lo = 1, hi = 3;
	/* t == 0 (or factors that give LOW >= HIGH): the reference's set() rejects a threshold <= 0 (canny_dete.cxx:86-99) and
	 * process() rejects tLow >= tHigh (:126) -- the sample then skips the frame.  A batched device path cannot skip: it uses the
	 * smallest legal pair (1,3) and reports the Otsu value so the caller can tell. */
	if (fLow > 0.f && fHigh > 0.f && fLow < fHigh) orc_canny_thresholds(fLow, fHigh, 0, 0, 1, 1, &lo, &hi);
	*tLow = lo; *tHigh = hi;
}

/* ================================================================================================================
 * Optional Gaussian pre-blur: CompVMathGauss::kernelDim1FixedPoint + CompVMathConvlt::convlt1FixedPoint
 * ================================================================================================================ */
int orc_gauss_kernel_f32(size_t size, float sigma, float* kernel)
{
	/* compv_math_gauss.h:24-55, T = float: note which sub-expressions are float and which are double */
	if (!kernel || !(size & 1)) return ORC_E_INVALID_PARAMETER;
	const size_t half = size >> 1;
	const float sigma2_times2 = (float)(2 * (sigma * sigma));
	const float a = (float)(1 / sqrt(3.14159265358979323846 * sigma2_times2)); /* COMPV_MATH_PI is a double literal */
	float sum = a;
	kernel[half] = a;
	for (size_t x = 1; x <= half; ++x) {
		const float k = (float)(a * exp(-(double)((x * x) / sigma2_times2)));
		kernel[x + half] = k;
		kernel[half - x] = k;
		sum += (k + k);
	}
	sum = 1 / sum;
	for (size_t x = 0; x < size; ++x) kernel[x] *= sum;
	return ORC_OK;
}

int orc_gauss_kernel_fixedpoint(size_t size, float sigma, uint16_t* kernel)
{
	if (!kernel || !(size & 1) || size > 255) return ORC_E_INVALID_PARAMETER;
	float f[255];
	int err = orc_gauss_kernel_f32(size, sigma, f);
	if (err) return err;
	for (size_t i = 0; i < size; ++i) kernel[i] = (uint16_t)(f[i] * 0xffff); /* compv_math_convlt.h:88 */
	return ORC_OK;
}

static void fxp_pass(const uint8_t* in, size_t W, size_t H, size_t S, const uint16_t* kern, size_t k, uint8_t* out, int vertical)
{
	const size_t r = k >> 1;
	for (size_t y = 0; y < H; ++y) {
		for (size_t x = 0; x < W; ++x) {
			const int border = vertical ? (y < r || y >= H - r) : (x < r || x >= W - r);
			if (border) { out[y * S + x] = 0; continue; }
			unsigned int sum = 0;
			for (size_t t = 0; t < k; ++t) {
				const unsigned int v = vertical ? in[(y - r + t) * S + x] : in[y * S + x - r + t];
				sum += (v * (unsigned int)kern[t]) >> 16; /* compv_math_convlt.h:393-396: per-tap mulhi, then saturate */
			}
			out[y * S + x] = (uint8_t)(sum > 255 ? 255 : sum);
		}
	}
}

int orc_convlt1_fixedpoint(const uint8_t* in, size_t W, size_t H, size_t S, const uint16_t* vt, const uint16_t* hz, size_t k, uint8_t* out)
{
	if (!in || W < k || H < k || S < W || !vt || !hz || !(k & 1) || !out) return ORC_E_INVALID_PARAMETER; /* compv_math_convlt.h:100 */
	uint8_t* tmp = (uint8_t*)calloc(S * H, 1);
	if (!tmp) return ORC_E_OUT_OF_MEMORY;
	fxp_pass(in, W, H, S, hz, k, tmp, 0);
	fxp_pass(tmp, W, H, S, vt, k, out, 1);
	free(tmp);
	return ORC_OK;
}

/* CompVHoughSht::toCartesian, core/features/hough/compv_core_feature_houghsht.cxx:566-589 (the "#if 1" branch) */
void orc_sht_to_cartesian(size_t W, size_t H, const orc_line* lines, size_t n, float* out)
{
	const float widthF = (float)W, heightF = (float)H;
	const float r = sqrtf((widthF * widthF) + (heightF * heightF));
	for (size_t i = 0; i < n; ++i) {
		const float theta = lines[i].theta, rho = lines[i].rho;
		float* o = out + 4 * i;
		if (theta == 0.f) { o[0] = rho; o[1] = r; o[2] = rho; o[3] = -r; }
		else {
			const float a = cosf(theta), b = (1.f / sinf(theta));
			o[0] = 0.f; o[1] = (rho * b);
			o[2] = widthF; o[3] = ((rho - (widthF * a)) * b);
		}
	}
}

/* CompVHoughKht::toCartesian, core/features/hough/compv_core_feature_houghkht.cxx:1249-1280: rho is measured from the image centre */
void orc_kht_to_cartesian(size_t W, size_t H, const orc_line* lines, size_t n, float* out)
{
	const float widthF = (float)W, heightF = (float)H;
	const float r = sqrtf((widthF * widthF) + (heightF * heightF));
	const float half_widthF = widthF * 0.5f, half_heightF = heightF * 0.5f;
	for (size_t i = 0; i < n; ++i) {
		const float rho = lines[i].rho, theta = lines[i].theta;
		float* o = out + 4 * i;
		if (theta == 0.f) { o[0] = o[2] = (rho + half_widthF); o[1] = r; o[3] = -r; }
		else {
			const float a = (cosf(theta) * half_widthF), b = (1.f / sinf(theta));
			o[0] = 0.f; o[1] = ((rho + a) * b) + half_heightF;
			o[2] = widthF; o[3] = ((rho - a) * b) + half_heightF;
		}
	}
}
