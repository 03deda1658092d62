// kht_host.cpp -- host stages of the kernel-based Hough transform (KHT).
//
// Replaces, behind compvhip_houghkht_u8 (include/compv_hip.h), CompVHoughKht::process
// (core/features/hough/compv_core_feature_houghkht.cxx:208-447).  KHT is a sequential, latency-bound algorithm: edge
// linking follows chains pixel by pixel in raster order and destroys the pixels it visits (Appendix A, :544-760) and the final
// sweep over the sorted vote cells (:1207-1247) is order dependent by definition.  Those stages stay on the host, in float64 with the reference's operation order so that the results are
// bit-identical; the data-parallel stages -- the cluster subdivision of every string, the per-cluster statistics of Algorithm 2, Algorithm-4 Gaussian voting into the
// (rho,theta) count map and the 3x3 smoothing + thresholding of that map -- run on the GPU (kht_kernels.hip).
//
// Compiled with -ffp-contract=off: every double operation below must round exactly once, like the SSE2 reference build.
#include "kht.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>

namespace compvhip {

namespace {
const float kPiF = 3.1415926535897932384626433f; // kfMathTrigPi (base/math/compv_math.cxx:27)
inline float piOver180() { return kPiF / 180.f; } // kfMathTrigPiOver180 (:30)
const double kPi = 3.14159265358979323846;      // M_PI
const double kTwoPi = 2.0 * kPi;
const double kRadToDeg = 180.0 / kPi;
} // namespace

// initCoords, houghkht.cxx:501-541
bool khtAxes(size_t W, size_t H, float rho, float thetaDeg, KhtAxes& ax)
{
	if (!W || !H || !(rho > 0.f) || rho > 1.f || !(thetaDeg > 0.f)) return false;
	ax.dRho = static_cast<double>(rho * 1.f);
	ax.dThetaRad = static_cast<double>(thetaDeg * piOver180());
	ax.dThetaDeg = (ax.dThetaRad * 180.0) / kPi; // COMPV_MATH_RADIAN_TO_DEGREE
	ax.r = std::sqrt(static_cast<double>((W * W) + (H * H)));
	ax.rhoN = static_cast<size_t>((ax.r + 1.0) / ax.dRho);
	ax.T = static_cast<size_t>(180.0 / ax.dThetaDeg);
	ax.W = W; ax.H = H;
	return ax.rhoN >= 2 && ax.T >= 2;
}

void khtFillAxes(const KhtAxes& ax, std::vector<double>& rho, std::vector<double>& theta)
{
	rho.assign(ax.rhoN, 0.0); theta.assign(ax.T, 0.0); // index 0 is never written by the reference (:519,526)
	double v = -(ax.r * 0.5);
	for (size_t i = 1; i < ax.rhoN; ++i, v += ax.dRho) rho[i] = v;
	v = 0.0;
	for (size_t i = 1; i < ax.T; ++i, v += ax.dThetaDeg) theta[i] = v;
}

// ---- Appendix A: linking (Algorithms 5 and 6) on a bit plane ------------------------------------------------------------------------------
// The reference walks a byte map: per step up to eight neighbour tests in the fixed priority TL,T,TR, L,R, BL,B,BR (:666-703), and the raster
// scan for the next seed touches every byte.  Here the edge map is ONE BIT per pixel in rows of 64-bit words with a zero border (a pad word left
// and right of every row, zero rows above and below): a step reads the 3-bit windows of the rows y-1, y, y+1, packs them into a 9-bit code whose
// bit order IS the priority order (TL T TR | L C R | BL B BR, the centre masked out) and looks the next pixel up by that code; pixels outside the
// image read as zero, which is what the reference's bounds tests amount to.  Loads and the erasing store are aligned 64-bit accesses of the same
// words (a narrower store followed by a wider load of the same bytes defeats store forwarding: +40 % on the walk).  The seed scan walks the words.
// The device-resident batch path downloads the plane already packed (1/8 of the bytes).  Same strings, same point order as the byte walk (tests:
// fixtures from the compiled reference).  tools/kht_lab/link_bench times this file alone (4K benchmark map, EPYC 9575F host of the GPU box: 2.6 ms,
// 7.8 ns per linked pixel -- the dependency chain of a step: three loads -> code -> table -> next address).
void KhtBitPlane::reset(size_t W_, size_t H_)
{
	W = W_; H = H_;
	pitch = ((W + 63) / 64 + 2) * 8;     // bytes: one pad word left, >= one right
	buf.assign((H + 2) * pitch, 0);
}

namespace {
// 32 pixels -> 32 bits (any non-zero byte is an edge, :556)
__attribute__((target("avx2"))) void packRowAvx2(const uint8_t* src, size_t W, uint8_t* dst)
{
	size_t x = 0;
	const __m256i zero = _mm256_setzero_si256();
	for (; x + 32 <= W; x += 32) {
		const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + x));
		const uint32_t m = ~static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, zero)));
		std::memcpy(dst + (x >> 3), &m, 4);
	}
	for (; x < W; ++x) if (src[x]) dst[x >> 3] = static_cast<uint8_t>(dst[x >> 3] | (1u << (x & 7)));
}
void packRowSwar(const uint8_t* src, size_t W, uint8_t* dst)
{
	size_t x = 0;
	for (; x + 8 <= W; x += 8) {
		uint64_t q; std::memcpy(&q, src + x, 8);
		if (!q) continue;   // (most of an edge map)
		// bit 7 of every non-zero byte, gathered into one byte
		const uint64_t nz = (((q & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | q) & 0x8080808080808080ull;
		dst[x >> 3] = static_cast<uint8_t>((nz * 0x0002040810204081ull) >> 56);
	}
	for (; x < W; ++x) if (src[x]) dst[x >> 3] = static_cast<uint8_t>(dst[x >> 3] | (1u << (x & 7)));
}
} // namespace

// bytes -> bits
void khtPackBytes(const uint8_t* e, size_t W, size_t H, size_t S, KhtBitPlane& plane)
{
	plane.reset(W, H);
	const bool avx2 = __builtin_cpu_supports("avx2");
	for (size_t y = 0; y < H; ++y) {
		uint8_t* dst = plane.row(static_cast<int>(y)) + 8;   // behind the pad word
		if (avx2) packRowAvx2(e + y * S, W, dst);
		else packRowSwar(e + y * S, W, dst);
	}
}

// rows of 32-bit mask words (bit i of word k = pixel 32 k + i: bytes_to_bits_kernel's layout) -> plane; bits past W are ignored
void khtPlaneFromWords(const uint32_t* words, size_t wordsPerRow, size_t W, size_t H, KhtBitPlane& plane)
{
	plane.reset(W, H);
	const size_t nbytes = (W + 7) / 8;
	for (size_t y = 0; y < H; ++y) {
		uint8_t* dst = plane.row(static_cast<int>(y)) + 8;
		std::memcpy(dst, words + y * wordsPerRow, nbytes);
		if (W & 7) dst[nbytes - 1] = static_cast<uint8_t>(dst[nbytes - 1] & ((1u << (W & 7)) - 1u));
	}
}

namespace {
// Horizontal run of the walk, many pixels per call (out of line: inlined, its registers cost the single-step loop a sixth of its speed).  The walk stands on
// pixel x (bit bx of *mw) and its next step goes right (d == 5: nothing above, nothing to the left) or left (d == 3: nothing above): it goes on to x +- 2, ...
// for as long as the next pixel is set and the pixel it is LEFT FROM has nothing in the row above (its other neighbour in the row is the pixel just erased;
// L and R outrank the row below).  Within the current word.  Erases and appends the n >= 1 pixels of the run; returns n.
__attribute__((noinline)) int khtRun(uint64_t* mw, ptrdiff_t P, int bx, bool right, int x, int y, KhtPoint* out)
{
	const uint64_t* const tw = mw - P;
	const uint64_t T0 = tw[0];
	// blocked pixels: the row above has a pixel at p - 1, p or p + 1 (neighbour words included); bit bx is clear
	const uint64_t blk = T0 | (T0 << 1) | (T0 >> 1) | (tw[-1] >> 63) | (tw[1] << 63);
	int n;
	if (right) {
		const int cntSet = __builtin_ctzll(~(mw[0] >> (bx + 1)));              // set pixels right of x, up to the end of the word
		const uint64_t br = blk >> bx;                                          // bit j = pixel x + j blocked
		const int firstBlocked = br ? __builtin_ctzll(br) : 64;
		n = cntSet < firstBlocked ? cntSet : firstBlocked;                     // >= 1: this is the step the caller's code decided
		mw[0] &= ~((n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << (bx + 1));
		for (int j = 1; j <= n; ++j) { out->x = x + j; out->y = y; ++out; }
	}
	else {
		const int cntSet = __builtin_clzll(~(mw[0] << (64 - bx)));             // set pixels left of x, down to the start of the word
		const uint64_t bl = blk << (63 - bx);                                   // bit 63 - j = pixel x - j blocked
		const int firstBlocked = bl ? __builtin_clzll(bl) : 64;
		n = cntSet < firstBlocked ? cntSet : firstBlocked;
		mw[0] &= ~((n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << (bx - n));
		for (int j = 1; j <= n; ++j) { out->x = x - j; out->y = y; ++out; }
	}
	return n;
}
} // namespace

// set pixels of the plane = the most points khtLink can produce (every point it emits erases one)
size_t khtPlaneCount(const KhtBitPlane& plane)
{
	const uint64_t* w = reinterpret_cast<const uint64_t*>(plane.buf.data());
	const size_t n = plane.buf.size() / 8;
	size_t c = 0;
	for (size_t i = 0; i < n; ++i) c += static_cast<size_t>(__builtin_popcountll(w[i]));
	return c;
}

// pts: room for khtPlaneCount(plane) points (the caller's buffer: pinned memory on the device path, so that the strings upload without staging)
size_t khtLink(KhtBitPlane& plane, size_t minSize, KhtPoint* pts, std::vector<KhtRange>& strings)
{
	strings.clear();
	const int W = static_cast<int>(plane.W), H = static_cast<int>(plane.H);
	const ptrdiff_t P = static_cast<ptrdiff_t>(plane.pitch / 8);              // words per row
	const int64_t PB = static_cast<int64_t>(P) * 64;                           // bits per row
	uint64_t* const base = reinterpret_cast<uint64_t*>(plane.row(0));          // pixel (x, y) = bit pos = y * PB + 64 + x of the plane behind base (row -1 and the pad words are zero)
	// The position of the walk is ONE number, the bit index: word = pos >> 6, bit = pos & 63, a step adds dy * PB + dx -- no multiply, no (x, y) -> address
	// arithmetic in the dependency chain of a step (window loads -> code -> table -> next position); x and y ride along for the output only.
	// The 9-bit code is TL T TR | L C R | BL B BR with the centre (the pixel the walk stands on) masked out: its lowest set bit is the reference's
	// priority order (TL, T, TR, L, R, BL, B, BR), d = 3 (dy + 1) + (dx + 1), and the whole step comes out of ONE table lookup by the code
	// (512 x 8 bytes): bit-index delta, dx, dy, d.
	struct Step { int32_t delta; int8_t dx, dy, d, pad; };
	Step steps[512];
	for (int code = 1; code < 512; ++code) {
		const int d = __builtin_ctz(static_cast<unsigned>(code));
		Step s; s.dx = static_cast<int8_t>(d % 3 - 1); s.dy = static_cast<int8_t>(d / 3 - 1); s.d = static_cast<int8_t>(d); s.pad = 0;
		s.delta = static_cast<int32_t>(s.dy * PB + s.dx);
		steps[code] = s;
	}
	steps[0] = Step{ 0, 0, 0, 4, 0 };
	// The horizontal-run shortcut (khtRun) pays on line art -- up to 63 pixels per call: a 4K map of stripes links in 0.7 instead of 5.6 ms -- and costs on
	// noisy maps, whose runs are one or two pixels long (72 000 calls for 120 000 pixels on the benchmark frame).  It is off until kRunProbe horizontal single
	// steps in a row were seen and switches itself off again while the runs it finds stay short.
	constexpr int kRunScore = 4, kRunProbe = 32;
	int runScore = 0, hz = 0;
	KhtPoint* out = pts;
	const uint64_t* const rowT = base - P; const uint64_t* const rowB = base + P;   // the rows above / below, same word index
	// walks from (x, y), appending every pixel it reaches and erasing it (and the start pixel); returns when no neighbour is left
	auto walk = [&](int64_t pos, int x, int y) {
		for (;;) {
			int d, bx;
			// single steps: nothing in this loop but the step itself (the run shortcut is left through a break -- with its call inside the loop the register
			// allocator spills in the loop, 15 % of the walk)
			for (;;) {
				const int64_t q = pos - 1;                               // bit index of the left neighbour
				const int64_t wi = q >> 6;
				const int sh = static_cast<int>(q & 63);
				// the window may straddle two words (x & 63 is 63 or 0: rare in a photograph, 42 % of the steps on the benchmark frames, whose checkerboard edges sit on
				// multiples of 64): always funnel the next word in -- for sh <= 61 its bits land above the three that are kept -- rather than branch on it
				const int up = 63 - sh;
				const uint64_t t = (rowT[wi] >> sh) | ((rowT[wi + 1] << 1) << up);
				const uint64_t m = (base[wi] >> sh) | ((base[wi + 1] << 1) << up);
				const uint64_t b = (rowB[wi] >> sh) | ((rowB[wi + 1] << 1) << up);
				// the pixel the walk stands on is masked out of the code here and erased in memory BEHIND the loads of its window (a store in front of them, to the
				// very word the centre row is read from, puts a read-modify-write and a store-to-load forward into every step's dependency chain)
				const uint32_t code = (static_cast<uint32_t>(t & 7u) | (static_cast<uint32_t>(m & 7u) << 3) | (static_cast<uint32_t>(b & 7u) << 6)) & ~0x10u;
				base[pos >> 6] &= ~(1ull << (pos & 63));
				if (!code) return;
				const Step st = steps[code];
				d = st.d;
				bx = static_cast<int>(pos & 63);                         // = x & 63
				if (__builtin_expect(runScore > 0, 0) && ((d == 5 && bx != 63) || (d == 3 && bx != 0))) break;
				pos += st.delta;
				x += st.dx;
				y += st.dy;
				out->x = x; out->y = y; ++out;
				hz = (d == 5 || d == 3) ? hz + 1 : 0;                    // (a select, not a branch)
				if (__builtin_expect(hz >= kRunProbe, 0)) { runScore = kRunScore; hz = 0; }
			}
			const int n = khtRun(base + (pos >> 6), P, bx, d == 5, x, y, out);
			const int sn = (d == 5) ? n : -n;
			x += sn; pos += sn; out += n;
			runScore = (n >= 8) ? (runScore < 62 ? runScore + 2 : 64) : runScore - 1;
		}
	};
	// raster scan of rows 1..H-2, columns 1..W-2 (:552-556), 64 pixels at a time; a walk may erase pixels of the word being scanned: reload
	const int lastWord = (W - 2) >> 6;
	for (int yr = 1; yr < H - 1; ++yr) {
		uint64_t* const row = base + static_cast<ptrdiff_t>(yr) * P + 1;
		for (int k = 0; k <= lastWord; ++k) {
			uint64_t valid = ~0ull;
			if (k == 0) valid &= ~1ull;                                             // column 0 is no seed
			if (k == lastWord && ((W - 2) & 63) != 63) valid &= (~0ull >> (63 - ((W - 2) & 63)));   // nor are columns >= W - 1
			for (;;) {
				const uint64_t v = row[k] & valid;
				if (!v) break;
				const int xr = 64 * k + __builtin_ctzll(v);
				const int64_t pr = static_cast<int64_t>(yr) * PB + 64 + xr;
				KhtPoint* const begin = out;
				row[k] &= ~(1ull << (xr & 63));
				out->x = xr; out->y = yr; ++out;
				walk(pr, xr, yr);                                    // forward: append and erase (:724-728)
				KhtPoint* const rev = out;
				walk(pr, xr, yr);                                    // backward from the reference pixel (:733-746)
				if (static_cast<size_t>(out - begin) >= minSize) {
					std::reverse(begin, rev);                        // the string then runs end to end (:751-755)
					KhtRange r; r.begin = static_cast<size_t>(begin - pts); r.end = static_cast<size_t>(out - pts); strings.push_back(r);
				}
				else out = begin;
			}
		}
	}
	return static_cast<size_t>(out - pts);
}

namespace {
// CompVMathEigen<double>::find2x2 (base/math/compv_math_eigen.cxx:285-342), sort = norm = true
void find2x2(const double (&A)[4], double (&D)[4], double (&Q)[4])
{
	bool norm = true;
	const double trace = A[0] + A[3];
	const double traceDiv2 = trace / 2.0;
	const double det = (A[0] * A[3]) - (A[1] * A[2]);
	const double sq = std::sqrt(((trace * trace) / 4.0) - det);
	D[1] = D[2] = 0.0;
	D[0] = traceDiv2 + sq;
	D[3] = traceDiv2 - sq;
	if (A[2] != 0) { Q[0] = D[0] - A[3]; Q[2] = A[2]; Q[1] = D[3] - A[3]; Q[3] = A[2]; }
	else if (A[1] != 0) { Q[0] = A[1]; Q[2] = D[0] - A[0]; Q[1] = A[1]; Q[3] = D[3] - A[0]; }
	else {
		norm = false;
		if (A[3] != 0.0) { Q[0] = 0.0; Q[2] = 1.0; Q[1] = 1.0; Q[3] = 0.0; }
		else { Q[0] = 1.0; Q[2] = 0.0; Q[1] = 0.0; Q[3] = 1.0; }
	}
	if (norm) {
		const double m02 = 1.0 / std::sqrt(Q[0] * Q[0] + Q[2] * Q[2]);
		const double m13 = 1.0 / std::sqrt(Q[1] * Q[1] + Q[3] * Q[3]);
		Q[0] *= m02; Q[2] *= m02; Q[1] *= m13; Q[3] *= m13;
	}
	if (D[0] < D[3]) {
		double a = Q[0], b = Q[2];
		Q[0] = Q[1]; Q[2] = Q[3]; Q[1] = a; Q[3] = b;
		a = D[0]; D[0] = D[3]; D[3] = a;
	}
}

// (1 + x/1024)^1024 (:77-88)
inline double expFastSmall(double x)
{
	x = 1.0 + (x * (1.0 / 1024.0));
	x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
	return x;
}

// __gauss_Eq15 (:834-846)
double gaussEq15(double rho, double theta, const KhtKernel& k)
{
	const double s = std::sqrt(k.sigmaRhoSquare) * std::sqrt(k.sigmaThetaSquare);
	const double sScale = 1.0 / s;
	const double r = k.sigmaRhoTimesTheta * sScale;
	const double omr = 1.0 - (r * r);
	const double x = 1.0 / (kTwoPi * s * std::sqrt(omr));
	const double y = 1.0 / (2.0 * omr);
	const double z = ((rho * rho) / k.sigmaRhoSquare) - (((r * 2.0) * rho * theta) * sScale) + ((theta * theta) / k.sigmaThetaSquare);
	return x * expFastSmall(-z * y);
}
} // namespace

// Algorithm 2 (:885-1026): the per-cluster statistics run on the GPU (kht_kernels.hip, kht_stats_kernel).  What is left here is the
// one libm call of the stage -- theta = acos(vx) in degrees (:949; the device acos is not glibc's) -- and hmax.
void khtFinishKernels(std::vector<KhtKernel>& kernels, double& hmax)
{
	hmax = 0.0;
	for (KhtKernel& K : kernels) {
		K.theta = std::acos(K.theta) * kRadToDeg;
		if (K.h > hmax) hmax = K.h;
	}
}

// discard short kernels (:1029-1041), Gmin (:1044-1062), GS (:377)
double khtPruneAndScale(std::vector<KhtKernel>& kernels, double hmax, double minHeight)
{
	const double hScale = 1.0 / hmax;
	kernels.erase(std::remove_if(kernels.begin(), kernels.end(), [&](const KhtKernel& k) { return (k.h * hScale) < minHeight; }), kernels.end());
	double Gmin = DBL_MAX;
	for (const KhtKernel& k : kernels) {
		const double M[4] = { k.sigmaRhoSquare, k.sigmaRhoTimesTheta, k.m2, k.sigmaThetaSquare };
		double D[4], Q[4];
		find2x2(M, D, Q);
		const double w = std::sqrt(D[3]);
		const double g = gaussEq15(Q[1] * w, Q[3] * w, k);
		if (g < Gmin) Gmin = g;
	}
	return (Gmin == 0.0) ? 1.0 : std::max(1.0 / Gmin, 1.0);
}

// Per-kernel constants of vote_Algorithm4 (:1091-1103) -- every division and square root of the voting stage is done here,
// on the host, so that the GPU loop only multiplies, adds and truncates.
void khtVoteParams(const KhtAxes& ax, const std::vector<KhtKernel>& kernels, std::vector<KhtVoteParams>& params)
{
	params.resize(kernels.size());
	const double rhoScale = 1.0 / ax.dRho, thetaScale = 1.0 / ax.dThetaDeg;
	const double rhoMaxNeg = -(ax.r * 0.5);
	for (size_t i = 0; i < kernels.size(); ++i) {
		const KhtKernel& k = kernels[i];
		KhtVoteParams& p = params[i];
		p.srsScale = 1.0 / k.sigmaRhoSquare;
		p.stsScale = 1.0 / k.sigmaThetaSquare;
		const double s = std::sqrt(k.sigmaRhoSquare) * std::sqrt(k.sigmaThetaSquare);
		p.sScale = 1.0 / s;
		const double r = k.sigmaRhoTimesTheta * p.sScale;
		const double omr = 1.0 - (r * r);
		p.r2 = r * 2.0;
		p.x = 1.0 / (kTwoPi * s * std::sqrt(omr));
		p.y = 1.0 / (2.0 * omr);
		p.rhoIndex = static_cast<unsigned>(static_cast<size_t>(std::fabs((k.rho - rhoMaxNeg) * rhoScale) + 0.5) + 1); // :1077
		p.thetaIndex = static_cast<unsigned>(static_cast<size_t>(std::fabs(k.theta * thetaScale) + 0.5) + 1);         // :1078
	}
}

// Section 3.4 (:1195-1247): std::sort on the count alone -- unstable but deterministic for one libstdc++ and one input
// order, which is why the cells are first put into the reference's emission order (theta-major; within a theta row the SIMD
// scan, then the scalar remainder) -- then the sweep with the visited map.
void khtPeaks(const KhtAxes& ax, std::vector<KhtCell>& cells, int maxLines, std::vector<KhtLine>& lines, KhtPeaksWork& wk)
{
	lines.clear();
	// (1) the reference's emission order: `order` is a unique key below 2^32 -- LSD radix sort, 11 bits per pass, only the passes the largest key needs
	{
		uint32_t maxKey = 0;
		for (const KhtCell& c : cells) maxKey = std::max(maxKey, c.order);
		wk.tmp.resize(cells.size());
		KhtCell* src = cells.data(); KhtCell* dst = wk.tmp.data();
		for (int shift = 0; shift < 32 && (maxKey >> shift) != 0; shift += 11) {
			uint32_t hist[2048] = { 0 };
			for (size_t i = 0; i < cells.size(); ++i) ++hist[(src[i].order >> shift) & 2047u];
			uint32_t sum = 0;
			for (uint32_t& h : hist) { const uint32_t c = h; h = sum; sum += c; }
			for (size_t i = 0; i < cells.size(); ++i) dst[hist[(src[i].order >> shift) & 2047u]++] = src[i];
			std::swap(src, dst);
		}
		if (src != cells.data()) std::memcpy(cells.data(), src, cells.size() * sizeof(KhtCell));
	}
	// (2) the reference's std::sort on the count alone (:1195-1205): unstable, but a function of the sequence of counts only -- the permutation is
	// found on 8-byte (count, position) records (same comparisons, same moves as on the reference's 24-byte cells) and applied afterwards
	std::vector<KhtPeaksWork::Rec>& recs = wk.recs;
	recs.resize(cells.size());
	for (size_t i = 0; i < cells.size(); ++i) { recs[i].count = cells[i].count; recs[i].pos = static_cast<uint32_t>(i); }
	// the indices of the cells out of their keys (kht_peaks_kernel: order = theta * 2 vs + rho, or ... + vs + rho): the keys are sorted, theta only ever grows -- no division
	std::vector<KhtPeaksWork::Idx>& idx = wk.idx;
	idx.resize(cells.size());
	{
		const uint64_t vsKey = ax.rhoN + 2, row = 2 * vsKey;
		uint64_t ti = 0, base = 0;   // base = ti * row
		for (size_t i = 0; i < cells.size(); ++i) {
			const uint64_t o = cells[i].order;
			while (o >= base + row) { ++ti; base += row; }
			const uint64_t rem = o - base;
			idx[i].theta = static_cast<uint32_t>(ti);
			idx[i].rho = static_cast<uint32_t>(rem >= vsKey ? rem - vsKey : rem);
		}
	}
	std::sort(recs.begin(), recs.end(), [](const KhtPeaksWork::Rec& a, const KhtPeaksWork::Rec& b) { return a.count > b.count; });
	// the axes and the visited map live in the caller's workspace: a worker thread of the batch entry point would otherwise map and unmap a megabyte per
	// frame (32 threads doing that at once spent more time in the kernel's address-space lock than in the sweep)
	if (wk.axW != ax.W || wk.axH != ax.H || wk.axRho != ax.dRho || wk.axTheta != ax.dThetaDeg || wk.rho.size() != ax.rhoN) {
		khtFillAxes(ax, wk.rho, wk.theta);
		wk.axW = ax.W; wk.axH = ax.H; wk.axRho = ax.dRho; wk.axTheta = ax.dThetaDeg;
	}
	const std::vector<double>& rho = wk.rho; const std::vector<double>& theta = wk.theta;
	const size_t vs = ax.rhoN + 2;
	if (wk.visited.size() != (ax.T + 2) * vs) wk.visited.assign((ax.T + 2) * vs, 0);
	std::vector<uint8_t>& visited = wk.visited;   // all zero on entry; the cells marked below are cleared again on the way out
	// the map is shared by the frames a worker handles one after the other: whatever ends the sweep (push_back may throw) must leave it all zero
	struct VisitedGuard {
		std::vector<uint8_t>& v; const std::vector<KhtPeaksWork::Idx>& idx; size_t vs;
		~VisitedGuard() { for (const KhtPeaksWork::Idx& c : idx) v[static_cast<size_t>(c.theta) * vs + c.rho] = 0; }
	} guard{ visited, idx, vs };
	for (const KhtPeaksWork::Rec& rec : recs) {
		const KhtPeaksWork::Idx c = idx[rec.pos];
		uint8_t* p = visited.data() + static_cast<size_t>(c.theta) * vs + c.rho;
		const uint8_t *t = p - vs, *b = p + vs;
		const bool seen = t[-1] || t[0] || t[1] || p[-1] || p[1] || b[-1] || b[0] || b[1];
		if (!seen) {
			KhtLine l;
			l.rho = static_cast<float>(rho[c.rho]);
			l.theta = static_cast<float>((theta[c.theta] * kPi) / 180.0); // COMPV_MATH_DEGREE_TO_RADIAN
			l.strength = rec.count;
			l.rhoIndex = static_cast<int32_t>(c.rho); l.thetaIndex = static_cast<int32_t>(c.theta);
			lines.push_back(l);
		}
		*p = 0xff;
	}
	if (maxLines > 0 && lines.size() > static_cast<size_t>(maxLines)) lines.resize(static_cast<size_t>(maxLines));
}

} // namespace compvhip
