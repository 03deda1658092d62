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

// ---- Appendix A: linking (Algorithms 5 and 6) --------------------------------------------------------------------------
namespace {
// Algorithm 6: the next set 8-neighbour in the fixed priority TL,T,TR, L,R, BL,B,BR (:666-703)
inline bool nextPixel(const uint8_t* e, size_t S, int W, int H, int& x, int& y)
{
	const int xs = x, ys = y;
	const bool left = xs > 0, right = (xs + 1) < W;
	const uint8_t* c = e + static_cast<size_t>(ys) * S + xs;
	if (ys > 0) {
		const uint8_t* t = c - S;
		if (left && t[-1]) { x = xs - 1; y = ys - 1; return true; }
		else if (*t) { y = ys - 1; return true; }
		else if (right && t[1]) { x = xs + 1; y = ys - 1; return true; }
	}
	if (left && c[-1]) { x = xs - 1; return true; }
	else if (right && c[1]) { x = xs + 1; return true; }
	else if ((ys + 1) < H) {
		const uint8_t* b = c + S;
		if (left && b[-1]) { x = xs - 1; y = ys + 1; return true; }
		else if (*b) { y = ys + 1; return true; }
		else if (right && b[1]) { x = xs + 1; y = ys + 1; return true; }
	}
	return false;
}
} // namespace

void khtLink(uint8_t* e, size_t W, size_t H, size_t S, size_t minSize, std::vector<KhtPos>& poss, std::vector<KhtRange>& strings)
{
	poss.clear(); strings.clear();
	const int Wi = static_cast<int>(W), Hi = static_cast<int>(H);
	const double hw = static_cast<double>(W) * 0.5, hh = static_cast<double>(H) * 0.5;
	auto push = [&](int y, int x) { KhtPos p; p.y = y; p.x = x; p.cx = x - hw; p.cy = y - hh; poss.push_back(p); };
	// raster scan of rows 1..H-2, columns 1..W-2 (:552-556); wide zero runs are skipped 8 bytes at a time
	for (int yr = 1; yr < Hi - 1; ++yr) {
		const uint8_t* row = e + static_cast<size_t>(yr) * S;
		for (int xr = 1; xr < Wi - 1; ++xr) {
			if (!row[xr]) {
				if (((reinterpret_cast<uintptr_t>(row + xr) & 7) == 0) && xr + 8 < Wi - 1) {
					uint64_t q; std::memcpy(&q, row + xr, 8);
					if (!q) { xr += 7; }
				}
				continue;
			}
			const size_t begin = poss.size();
			int x = xr, y = yr;
			for (;;) { // forward: append and erase (:724-728)
				push(y, x);
				e[static_cast<size_t>(y) * S + x] = 0;
				if (!nextPixel(e, S, Wi, Hi, x, y)) break;
			}
			const size_t rev = poss.size();
			x = xr; y = yr;
			if (nextPixel(e, S, Wi, Hi, x, y)) { // backward from the reference pixel (:733-746)
				for (;;) {
					push(y, x);
					e[static_cast<size_t>(y) * S + x] = 0;
					if (!nextPixel(e, S, Wi, Hi, x, y)) break;
				}
			}
			const size_t end = poss.size();
			if ((end - begin) >= minSize) {
				std::reverse(poss.begin() + begin, poss.begin() + rev); // the string then runs end to end (:751-755)
				KhtRange r; r.begin = begin; r.end = end; strings.push_back(r);
			}
			else poss.resize(begin);
		}
	}
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
void khtPeaks(const KhtAxes& ax, std::vector<KhtCell>& cells, int maxLines, std::vector<KhtLine>& lines)
{
	lines.clear();
	std::sort(cells.begin(), cells.end(), [](const KhtCell& a, const KhtCell& b) { return a.order < b.order; });
	std::sort(cells.begin(), cells.end(), [](const KhtCell& a, const KhtCell& b) { return a.count > b.count; });
	std::vector<double> rho, theta;
	khtFillAxes(ax, rho, theta);
	const size_t vs = ax.rhoN + 2;
	std::vector<uint8_t> visited((ax.T + 2) * vs, 0);
	for (const KhtCell& c : cells) {
		uint8_t* p = visited.data() + static_cast<size_t>(c.thetaIndex) * vs + c.rhoIndex;
		const uint8_t *t = p - vs, *b = p + vs;
		const bool seen = t[-1] || t[0] || t[1] || p[-1] || p[1] || b[-1] || b[0] || b[1];
		if (!seen) {
			KhtLine l;
			l.rho = static_cast<float>(rho[c.rhoIndex]);
			l.theta = static_cast<float>((theta[c.thetaIndex] * kPi) / 180.0); // COMPV_MATH_DEGREE_TO_RADIAN
			l.strength = c.count;
			l.rhoIndex = static_cast<int32_t>(c.rhoIndex); l.thetaIndex = static_cast<int32_t>(c.thetaIndex);
			lines.push_back(l);
		}
		*p = 0xff;
	}
	if (maxLines > 0 && lines.size() > static_cast<size_t>(maxLines)) lines.resize(static_cast<size_t>(maxLines));
}

} // namespace compvhip
