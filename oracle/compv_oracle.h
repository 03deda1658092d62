/* TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C) of the CompV Sobel -> Canny -> Hough(SHT) hot path.
 *
 * This is the parity ORACLE for the HIP path in compv_amd/.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library never links, loads or calls it.
 *
 * Parity is PINNED: (1) orc_convlt1_* reproduces the reference's own synthetic known-answer MD5s
 * (unittests/math_convlt.cxx:24-25, cases 5 and 6); (2) every function below is checked bit-for-bit
 * against the real CompV library compiled from /root/reference (oracle/build_ref.sh -> oracle/_ref)
 * by tests/test_oracle_vs_ref.py, and against fixtures generated from it (tests/golden/).
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef COMPV_ORACLE_H
#define COMPV_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_OP_SOBEL3 = 0, ORC_OP_SOBEL5 = 1, ORC_OP_SCHARR = 2, ORC_OP_PREWITT = 3 };
enum { ORC_THRESHOLD_COMPARE_TO_GRADIENT = 0, ORC_THRESHOLD_PERCENT_OF_MEAN = 1 };
enum { ORC_OK = 0, ORC_E_INVALID_PARAMETER = -1, ORC_E_INVALID_STATE = -2, ORC_E_OUT_OF_MEMORY = -3 };

typedef struct orc_line { float rho; float theta; int64_t strength; int32_t row; int32_t col; } orc_line;

/* SURVEY.md 8(d) deterministic synthetic frame (checkerboard + 4-bit LCG noise + slanted bright lines). */
void orc_synth_frame(uint8_t* out, size_t W, size_t H, size_t S, uint32_t seed);

/* Separable correlation, zero OUTPUT border, int32 accumulate, saturate to int16 after each pass. */
int orc_convlt1_8u16s16s(const uint8_t* in, size_t W, size_t H, size_t S, const int16_t* vt, const int16_t* hz, size_t k, int16_t* out);
int orc_convlt1_16s16s16s(const int16_t* in, size_t W, size_t H, size_t S, const int16_t* vt, const int16_t* hz, size_t k, int16_t* out);

/* gx, gy (int16) and g = |gx|+|gy| (uint16), all S*H elements. */
int orc_gradient(const uint8_t* in, size_t W, size_t H, size_t S, int op, int16_t* gx, int16_t* gy, uint16_t* g);

/* Sobel/Scharr/Prewitt edge detector (normalised gradient magnitude), quirk Q1 included. */
int orc_edge_dete(const uint8_t* in, size_t W, size_t H, size_t S, int op, uint8_t* out, size_t So, uint16_t* gmax_out);

/* Canny thresholds exactly as process() derives them. sum = sum of input pixels (only read in mean mode). */
int orc_canny_thresholds(float fLow, float fHigh, int type, uint32_t sum, size_t W, size_t H, uint16_t* tLow, uint16_t* tHigh);

/* Column coverage of the SIMD + scalar-remainder dispatch (quirk Q3): covered x are [1,simdEnd) U [cStart,W-1). */
void orc_canny_coverage(size_t W, size_t* simdEnd, size_t* cStart);

/* Full Canny. gnms (optional, S*H u16) receives the gradient after non-maximum suppression. */
int orc_canny(const uint8_t* in, size_t W, size_t H, size_t S, float fLow, float fHigh, int ksize, int type,
              uint8_t* out, size_t So, uint16_t* gnms);

/* SHT geometry and Q16 tables. thetaRad_out = (float)thetaDeg * kfMathTrigPiOver180 (f32). */
int orc_sht_dims(size_t W, size_t H, float thetaDeg, size_t* R, size_t* T, float* thetaRad_out);
int orc_sht_tables(float thetaDeg, size_t T, int32_t* sinQ, int32_t* cosQ);
int orc_sht_acc(const uint8_t* edges, size_t W, size_t H, size_t S, const int32_t* sinQ, const int32_t* cosQ, size_t T,
                int32_t* acc, size_t accStride);
/* NMS (quirk Q2) + threshold + canonical sort (strength desc, row asc, col asc); maxLines <= 0 = unlimited. */
int orc_sht_lines(const int32_t* acc, size_t R, size_t T, size_t accStride, int32_t threshold, int32_t barrier, float thetaRad,
                  int maxLines, orc_line* lines, size_t cap, size_t* n);
int orc_sht(const uint8_t* edges, size_t W, size_t H, size_t S, float thetaDeg, int32_t threshold, int maxLines,
            orc_line* lines, size_t cap, size_t* n);
/* all n lines of a frame (canonical order in) -> the order the reference's unstable std::sort leaves them in (kht_sort.cpp) */
void orc_sht_reference_order(orc_line* lines, size_t n);

/* ---- caller-side pre-processing of the samples (samples/hough_lines/main.cxx:102-105) ---- */
/* Pixel formats, numbered as compvhip_pixfmt in include/compv_hip.h. */
enum { ORC_FMT_RGBA32 = 0, ORC_FMT_ARGB32, ORC_FMT_BGRA32, ORC_FMT_RGB24, ORC_FMT_BGR24, ORC_FMT_RGB565LE, ORC_FMT_RGB565BE,
       ORC_FMT_BGR565LE, ORC_FMT_BGR565BE, ORC_FMT_YUYV422, ORC_FMT_UYVY422, ORC_FMT_Y, ORC_FMT_COUNT };
/* bytes per sample (pixel) of a packed format, 0 if invalid */
int orc_fmt_bytes(int fmt);
/* CompVImage::convertGrayscale: in = H rows of S samples (S * orc_fmt_bytes bytes per row), out = H rows at stride So. */
int orc_grayscale(const uint8_t* in, int fmt, size_t W, size_t H, size_t S, uint8_t* out, size_t So);
/* CompVMathHistogram::build on a u8 plane (cols only). */
void orc_hist256(const uint8_t* in, size_t W, size_t H, size_t S, uint32_t* hist);
/* CompVImageThreshold::otsu from a histogram / from an image; N = W*H. Returns the threshold 0..255. */
int orc_otsu_from_hist(const uint32_t* hist, size_t N);
int orc_otsu(const uint8_t* in, size_t W, size_t H, size_t S);
/* thresholds the sample derives from the Otsu value t for Canny: LOW = (float)(t*0.5), HIGH = (float)t, then the
 * COMPARE_TO_GRADIENT rule of orc_canny_thresholds. */
void orc_otsu_canny_thresholds(int t, float fLowFactor, float fHighFactor, int* tLow, int* tHigh);

/* ---- optional Gaussian pre-blur ---- */
/* CompVMathGauss::kernelDim1<float> followed by CompVMathConvlt::fixedPointKernel (base/include/compv/base/math/compv_math_gauss.h:24-55,
 * base/include/compv/base/math/compv_math_convlt.h:77-92): size odd, kernel receives `size` Q16 weights. */
int orc_gauss_kernel_f32(size_t size, float sigma, float* kernel);
int orc_gauss_kernel_fixedpoint(size_t size, float sigma, uint16_t* kernel);
/* CompVMathConvlt::convlt1FixedPoint (compv_math_convlt.h:31-33,98-173,386-405): hz pass then vt pass, each
 * out = clip255(sum_k ((in[k] * kern[k]) >> 16)), zero OUTPUT border of kernSize/2, u8 intermediate. out has stride S. */
int orc_convlt1_fixedpoint(const uint8_t* in, size_t W, size_t H, size_t S, const uint16_t* vt, const uint16_t* hz, size_t k, uint8_t* out);

/* CompVHoughSht::toCartesian (core/features/hough/compv_core_feature_houghsht.cxx:566-589): out[4*i..] = a.x, a.y, b.x, b.y */
void orc_sht_to_cartesian(size_t W, size_t H, const orc_line* lines, size_t n, float* out);
void orc_kht_to_cartesian(size_t W, size_t H, const orc_line* lines, size_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif
