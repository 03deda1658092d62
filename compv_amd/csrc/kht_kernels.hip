// kht_kernels.hip -- GPU stages of the kernel-based Hough transform for gfx950.
//
//   kht_subdivide_kernel  clusters_find / clusters_subdivision (:762-832): every string is split recursively at its point of largest
//                     deviation from the chord while a half scores a better length / deviation ratio.  One wave = one string: lane 0
//                     runs the recursion on an explicit stack in global memory (depth <= the number of clusters the string can have;
//                     float64 with __ddiv_rn / __dsqrt_rn), all 64 lanes scan the points for the largest integer deviation.
//                     kht_gather_clusters_kernel (one workgroup per frame of the batch) puts the strings' clusters into one list per frame, in string order.
//   kht_stats_kernel  voting_Algorithm2_Kernels + CompVHoughKhtKernelHeight_* + CompVMathEigen<double>::find2x2 (:849-1026,
//                     base/math/compv_math_eigen.cxx:285-342): centroid, covariance, closed-form eigenvectors, rho, Eq. 14 terms and
//                     the kernel height of every cluster.  One thread = one cluster: the float64 sums run in the reference's
//                     order (a tree reduction would round differently), clusters are independent.
//   kht_vote_kernel   vote_Algorithm4 (core/features/hough/compv_core_feature_houghkht.cxx:1088-1148): every elliptical
//                     Gaussian kernel is rasterised into the int32 (rho,theta) count map in four quadrant walks.  One
//                     thread = one (kernel, quadrant) walk; votes are added with global int32 atomics, exactly like the
//                     reference's __sync_fetch_and_add, so the map is independent of the execution order.
//   kht_peaks_kernel  peaks_Section3_4_VotesCount (:1151-1192,1282-1308 and intrin_sse2.cxx:20-96): 3x3 binomial
//                     smoothing of the non-zero cells, threshold, compaction of the surviving cells (with their position
//                     in the reference's emission order, which the host needs for the order-dependent sweep).
//
// Built with -ffp-contract=off: the float64 expressions below must round once per operation, like the reference's SSE2
// code, or the integer votes differ.  The voting loop has no division or square root (KhtVoteParams holds them); the statistics
// kernel uses the correctly rounded __ddiv_rn / __dsqrt_rn, which are bit-identical to the host's IEEE operations.
#include "kht.hpp"

namespace compvhip {

// static_cast<int32_t>(double) as the reference's x86 build performs it (houghkht.cxx:1123, cvttsd2si): a value that does not fit -- the vote of a
// kernel whose points are EXACTLY collinear (synthetic checkerboards: sigma -> the floor, the Gaussian's peak beyond 2^31) -- or a NaN becomes
// INT_MIN, "no vote" for the loops below.  v_cvt_i32_f64 saturates to INT_MAX instead and the walk would go on voting (found by tools/fuzz_parity.py).
__device__ __forceinline__ int cvttsd2si(double v)
{
	return (v < 2147483648.0 && v > -2147483649.0) ? (int)v : (int)0x80000000u;
}

__device__ __forceinline__ double exp_fast_small(double x)
{
	// (1 + x/1024)^1024, houghkht.cxx:77-88
	x = 1.0 + (x * (1.0 / 1024.0));
	x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
	return x;
}


__global__ __launch_bounds__(64) void kht_subdivide_kernel(KhtSubdivArgs a)
{
	// One WAVE per string.  Lane 0 runs the recursion (explicit stack and cluster list in global memory, float64 ratios); the only
	// O(length) part of a call -- the point of largest deviation from the chord, first index on ties (:779-787) -- is scanned by all 64
	// lanes and reduced (integer arithmetic: order independent).
	const int sidx = blockIdx.x;
	const int lane = threadIdx.x;
	const KhtStringDesc sd = a.strings[sidx];
	const KhtPoint* __restrict__ P = a.pts + sd.begin;
	KhtSpan* __restrict__ out = a.scratch + sd.slot;
	KhtSubdivFrame* __restrict__ st = a.stack + sd.slot;
	const int maxDepth = (int)khtSubdivSlots(sd.end - sd.begin, (size_t)a.minSize);
	int outCount = 0, sp = 0; // lane 0 only
	double ret = 0.0;
	if (lane == 0) { st[0].s = 0; st[0].e = (int)(sd.end - sd.begin) - 1; st[0].state = 0; sp = 1; }
	for (;;) {
		// lane 0 unwinds until the frame on top needs its scan (state 0) or the stack is empty
		int s = -1, e = -1;
		if (lane == 0) {
			while (sp > 0 && st[sp - 1].state != 0) {
				KhtSubdivFrame& f = st[sp - 1];
				if (f.state == 1) {
					f.rl = ret; f.state = 2;
					st[sp].s = f.m; st[sp].e = f.e; st[sp].state = 0; ++sp;
				}
				else {
					const double rl = f.rl, rr = ret;
					if (rl > f.ratio || rr > f.ratio) ret = rl > rr ? rl : rr; // the halves stay
					else {
						outCount = f.keep;
						out[outCount].begin = sd.begin + (uint32_t)f.s; out[outCount].end = sd.begin + (uint32_t)f.e + 1u; ++outCount;
						ret = f.ratio;
					}
					--sp;
				}
			}
			if (sp > 0) { s = st[sp - 1].s; e = st[sp - 1].e; }
		}
		s = __builtin_amdgcn_readfirstlane(s); e = __builtin_amdgcn_readfirstlane(e);
		if (s < 0) break;
		const KhtPoint ps = P[s], pe = P[e];
		const int diffx = ps.x - pe.x, diffy = ps.y - pe.y;
		// key = deviation << 32 | ~index: the largest key is the largest deviation at its FIRST index
		unsigned long long best = 0ull;
		for (int i = s + 1 + lane; i < e; i += 64) {
			const KhtPoint pi = P[i];
			const int d = ((ps.x - pi.x) * diffy) - ((ps.y - pi.y) * diffx);
			const unsigned long long key = ((unsigned long long)(uint32_t)(d < 0 ? -d : d) << 32) | (uint32_t)~(uint32_t)i;
			best = key > best ? key : best;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const unsigned long long other = __shfl_xor(best, o);
			best = other > best ? other : best;
		}
		if (lane == 0) {
			KhtSubdivFrame& f = st[sp - 1];
			const int maxDev = (int)(best >> 32);
			const int maxIndex = maxDev > 0 ? (int)~(uint32_t)(best & 0xffffffffull) : s;
			const double length = __dsqrt_rn((double)((diffx * diffx) + (diffy * diffy)));
			const double q = __ddiv_rn((double)maxDev, length);
			f.ratio = __ddiv_rn(length, (q < a.minDev) ? a.minDev : q); // length / std::max(maxDev / length, minDev), operand order included
			f.keep = outCount; f.m = maxIndex;
			const bool split = (maxIndex - s + 1) >= a.minSize && (e - maxIndex + 1) >= a.minSize && maxIndex > s;
			if (split && sp >= maxDepth) a.total[a.flagIndex] = 1u;   // out of recursion slots (khtSubdivSlots bounds the depth for minSize >= 2): the call fails
			if (split && sp < maxDepth) {
				f.state = 1;
				st[sp].s = s; st[sp].e = maxIndex; st[sp].state = 0; ++sp;
			}
			else {
				outCount = f.keep;
				out[outCount].begin = sd.begin + (uint32_t)s; out[outCount].end = sd.begin + (uint32_t)e + 1u; ++outCount;
				ret = f.ratio; --sp;
			}
		}
	}
	if (lane == 0) a.counts[sidx] = (uint32_t)outCount;
}

// one workgroup per frame: exclusive scan of the per-string counts of the frame's strings, then its clusters are copied into one list in string order
__global__ __launch_bounds__(1024) void kht_gather_clusters_kernel(KhtSubdivArgs a, KhtBatchStrings tab)
{
	__shared__ uint32_t s_part[1024];
	__shared__ uint32_t s_carry;
	const int f = blockIdx.x;
	const int sb = (int)tab.stringBegin[f], se = (int)tab.stringBegin[f + 1];
	KhtSpan* __restrict__ dst = a.clusters + tab.clusterBase[f];
	if (threadIdx.x == 0) s_carry = 0;
	__syncthreads();
	for (int base = sb; base < se; base += 1024) {
		const int i = base + (int)threadIdx.x;
		const uint32_t c = i < se ? a.counts[i] : 0u;
		s_part[threadIdx.x] = c;
		__syncthreads();
		for (int o = 1; o < 1024; o <<= 1) { // Hillis-Steele inclusive scan
			const uint32_t v = threadIdx.x >= (unsigned)o ? s_part[threadIdx.x - o] : 0u;
			__syncthreads();
			s_part[threadIdx.x] += v;
			__syncthreads();
		}
		const uint32_t off = s_carry + s_part[threadIdx.x] - c;
		if (i < se) {
			const KhtSpan* __restrict__ src = a.scratch + a.strings[i].slot;
			for (uint32_t k = 0; k < c; ++k) dst[off + k] = src[k];
		}
		__syncthreads();
		if (threadIdx.x == 1023) s_carry += s_part[1023];
		__syncthreads();
	}
	if (threadIdx.x == 0) a.total[f] = s_carry;
}

// CompVMathEigen<double>::find2x2 (base/math/compv_math_eigen.cxx:285-342), sort = norm = true
__device__ __forceinline__ void find2x2_dev(double a0, double a1, double a2, double a3, double (&D)[4], double (&Q)[4])
{
	bool norm = true;
	const double trace = a0 + a3;
	const double traceDiv2 = __ddiv_rn(trace, 2.0);
	const double det = (a0 * a3) - (a1 * a2);
	const double sq = __dsqrt_rn(__ddiv_rn(trace * trace, 4.0) - det);
	D[1] = D[2] = 0.0;
	D[0] = traceDiv2 + sq;
	D[3] = traceDiv2 - sq;
	if (a2 != 0) { Q[0] = D[0] - a3; Q[2] = a2; Q[1] = D[3] - a3; Q[3] = a2; }
	else if (a1 != 0) { Q[0] = a1; Q[2] = D[0] - a0; Q[1] = a1; Q[3] = D[3] - a0; }
	else {
		norm = false;
		if (a3 != 0.0) { Q[0] = 0.0; Q[2] = 1.0; Q[1] = 1.0; Q[3] = 0.0; }
		else { Q[0] = 1.0; Q[2] = 0.0; Q[1] = 0.0; Q[3] = 1.0; }
	}
	if (norm) {
		const double m02 = __ddiv_rn(1.0, __dsqrt_rn(Q[0] * Q[0] + Q[2] * Q[2]));
		const double m13 = __ddiv_rn(1.0, __dsqrt_rn(Q[1] * Q[1] + Q[3] * Q[3]));
		Q[0] *= m02; Q[2] *= m02; Q[1] *= m13; Q[3] *= m13;
	}
	if (D[0] < D[3]) {
		double a = Q[0], b = Q[2];
		Q[0] = Q[1]; Q[2] = Q[3]; Q[1] = a; Q[3] = b;
		a = D[0]; D[0] = D[3]; D[3] = a;
	}
}

__global__ __launch_bounds__(64) void kht_stats_kernel(KhtStatsArgs a, KhtBatchStats tab)
{
	const int k = blockIdx.x * blockDim.x + threadIdx.x;   // cluster of frame blockIdx.y
	const int f = blockIdx.y;
	if (k >= tab.n[f]) return;
	const double kRadToDeg = 180.0 / 3.14159265358979323846, kTwoPi = 2.0 * 3.14159265358979323846;
	const KhtSpan c = a.clusters[tab.clusterBase[f] + k];
	const KhtPoint* __restrict__ b = a.pts + c.begin;
	const uint32_t cnt = c.end - c.begin;
	const double nScale = __ddiv_rn(1.0, (double)cnt);
	// centred coordinates as CompVHoughKhtPos stores them (:557-560): cx = x - W/2, cy = y - H/2 (exact)
	double mx = 0, my = 0;
	for (uint32_t i = 0; i < cnt; ++i) { const KhtPoint p = b[i]; mx += (double)p.x - a.hw; my += (double)p.y - a.hh; }
	mx *= nScale; my *= nScale;
	double cxx = 0, cyy = 0, cxy = 0;
	for (uint32_t i = 0; i < cnt; ++i) {
		const KhtPoint p = b[i];
		const double cx = ((double)p.x - a.hw) - mx, cy = ((double)p.y - a.hh) - my;
		cxx += cx * cx; cyy += cy * cy; cxy += cx * cy;
	}
	double D[4], Q[4];
	find2x2_dev(cxx, cxy, cxy, cyy, D, Q);
	const double ux = Q[0], uy = Q[2];
	double vx = Q[1], vy = Q[3];
	if (vy < 0.0) { vx = -vx; vy = -vy; }
	KhtKernel K;
	K.rho = (vx * mx) + (vy * my);
	K.theta = vx; // acos on the host
	const double sq = __dsqrt_rn(1.0 - (vx * vx));
	const double M0 = -(ux * mx) - (uy * my);
	const double M2 = (sq == 0.0) ? 0.0 : (__ddiv_rn(ux, sq) * kRadToDeg);
	double r0 = 0.0;
	for (uint32_t i = 0; i < cnt; ++i) {
		const KhtPoint p = b[i];
		const double r1 = (ux * (((double)p.x - a.hw) - mx)) + (uy * (((double)p.y - a.hh) - my));
		r0 += r1 * r1;
	}
	const double inv = __ddiv_rn(1.0, r0);
	const double r1 = M0 * inv, r2 = M2 * inv;
	double srs = r1 * M0 + nScale;
	const double srt = r1 * M2;
	const double m2 = r2 * M0;
	double sts = r2 * M2;
	if (sts == 0.0) sts = 0.1;
	srs *= 4.0; sts *= 4.0;
	const double s = __dsqrt_rn(srs) * __dsqrt_rn(sts);
	const double rr = __ddiv_rn(srt, s);
	const double omr = 1.0 - (rr * rr);
	// kernel height: SIMD operation order 1/((sqrt(1-r^2)*s)*2pi) for the clusters the reference's vector loop takes
	// (intrin_avx.cxx:42-63, intrin_sse2.cxx:118-142), the C order 1/(2pi*s*sqrt(1-r^2)) for its remainder (:849-883)
	const double h = (k < tab.simdEnd[f]) ? __ddiv_rn(1.0, (__dsqrt_rn(omr) * s) * kTwoPi) : __ddiv_rn(1.0, kTwoPi * s * __dsqrt_rn(omr));
	K.sigmaRhoSquare = srs; K.sigmaRhoTimesTheta = srt; K.m2 = m2; K.sigmaThetaSquare = sts; K.h = h;
	a.out[tab.clusterBase[f] + k] = K;
}

__global__ __launch_bounds__(64) void kht_vote_kernel(KhtGpuArgs a, KhtBatchVote tab)
{
	const int id = blockIdx.x * blockDim.x + threadIdx.x;   // (kernel, quadrant) of frame blockIdx.y
	const int f = blockIdx.y;
	if (id >= tab.nKernels[f] * 4) return;
	const KhtVoteParams p = a.params[tab.paramsBase[f] + (id >> 2)];
	a.counts += (size_t)f * tab.mapElems;
	a.gs = tab.gs[f];
	const int quad = id & 3;
	// the four quadrants (:1080-1083)
	int incRhoIndex = (quad & 2) ? -1 : 1;
	const int incThetaIndex = (quad & 1) ? -1 : 1;
	unsigned long long rhoStartIndex = (unsigned long long)p.rhoIndex - ((quad & 2) ? 1u : 0u);
	unsigned long long thetaIndex = (unsigned long long)p.thetaIndex - ((quad & 1) ? 1u : 0u);
	const double rhoStart = (quad & 2) ? -a.dRho : 0.0;
	const double thetaStart = (quad & 1) ? -a.dThetaDeg : 0.0;

	const unsigned long long rhoSize = (unsigned long long)a.rhoN, thetaSize = (unsigned long long)a.T;
	const double incRho = a.dRho * incRhoIndex;      // fixed before any wrap-around flips incRhoIndex (:1092)
	const double incTheta = a.dThetaDeg * incThetaIndex;
	double theta = thetaStart, rho;
	unsigned long long thetaCount = 0;
	do {
		// kernel exceeds the parameter-space limits: wrap theta, mirror rho (:1114-1118)
		if (!thetaIndex || thetaIndex > thetaSize) {
			rhoStartIndex = (rhoSize - rhoStartIndex) + 1;
			thetaIndex = thetaIndex ? 1 : thetaSize;
			incRhoIndex = -incRhoIndex;
		}
		if (rhoStartIndex >= 1) {
			int32_t* pcount = a.counts + thetaIndex * (unsigned long long)a.stride;
			unsigned long long rhoIndex = rhoStartIndex;
			rho = rhoStart;
			const double w = (theta * theta) * p.stsScale;
			const double k = p.r2 * theta * p.sScale;
			double krho = k * rho;
			const double ki = k * incRho;
			double z = ((rho * rho) * p.srsScale) - krho + w;
			int votes;
			while ((rhoIndex <= rhoSize) && (votes = cvttsd2si(((p.x * exp_fast_small(-z * p.y)) * a.gs) + 0.5)) > 0) {
				atomicAdd(&pcount[rhoIndex], votes);
				rhoIndex += (long long)incRhoIndex;
				rho += incRho;
				krho += ki;
				z = ((rho * rho) * p.srsScale) - krho + w;
			}
			thetaIndex += (long long)incThetaIndex;
			theta += incTheta;
		}
		else break;
	} while ((rho != rhoStart) && (++thetaCount < thetaSize));
}

__device__ __forceinline__ int smooth3x3(const int32_t* c, int stride)
{
	const int32_t *t = c - stride, *b = c + stride;
	return t[-1] + (t[0] << 1) + t[1] + b[-1] + (b[0] << 1) + b[1] + (c[-1] << 1) + (c[0] << 2) + (c[1] << 1);
}

__global__ __launch_bounds__(256) void kht_peaks_kernel(KhtGpuArgs a, KhtBatchVote tab, int sseCovEnd, int consumed, int remains, int simd)
{
	const int ti = blockIdx.y + 1;                           // theta index 1 .. T-1 (:437)
	const int c = blockIdx.x * blockDim.x + threadIdx.x + 1; // column 1 .. rhoN
	const int f = blockIdx.z;
	if (ti >= a.T || c > a.rhoN || tab.nKernels[f] <= 0) return;   // a frame without kernels has no votes (and the reference returns before this stage)
	a.counts += (size_t)f * tab.mapElems; a.cells += (size_t)f * tab.cellCap; a.cellCount += f;
	const int vs = a.rhoN + 2;
	const int32_t* cell = a.counts + (size_t)ti * a.stride + c;
	const int v = *cell;
	if (!v) return;
	int emitRho = -1; uint32_t order = 0;
	if (simd) {
		if (c < sseCovEnd) { if (v > 0) { emitRho = c; order = (uint32_t)ti * 2u * vs + c; } }
		else if (c > consumed && c < consumed + remains) { emitRho = c - consumed; order = (uint32_t)ti * 2u * vs + vs + (c - consumed); } // quirk Q6
	}
	else if (c < a.rhoN) { emitRho = c; order = (uint32_t)ti * 2u * vs + c; }
	if (emitRho < 0) return;
	const int s = smooth3x3(cell, a.stride);
	if (s < a.threshold) return;
	const int idx = atomicAdd(a.cellCount, 1);
	if (idx < a.cellCap) {
		KhtCell o; o.order = order; o.count = s;   // (rho index emitRho and theta index ti are in `order`)
		a.cells[idx] = o;
	}
}

hipError_t launch_kht_subdivide(const KhtSubdivArgs& a, const KhtBatchStrings& tab, hipStream_t stream)
{
	if (a.nStrings <= 0 || tab.frames <= 0) return hipSuccess;
	hipLaunchKernelGGL(kht_subdivide_kernel, dim3(a.nStrings), dim3(64), 0, stream, a);
	hipLaunchKernelGGL(kht_gather_clusters_kernel, dim3(tab.frames), dim3(1024), 0, stream, a, tab);
	return hipGetLastError();
}

hipError_t launch_kht_stats(const KhtStatsArgs& a, const KhtBatchStats& tab, hipStream_t stream)
{
	int most = 0;
	for (int f = 0; f < tab.frames; ++f) most = tab.n[f] > most ? tab.n[f] : most;
	if (most <= 0) return hipSuccess;
	hipLaunchKernelGGL(kht_stats_kernel, dim3((most + 63) / 64, tab.frames), dim3(64), 0, stream, a, tab);
	return hipGetLastError();
}

hipError_t launch_kht_vote(const KhtGpuArgs& a, const KhtBatchVote& tab, hipStream_t stream)
{
	int most = 0;
	for (int f = 0; f < tab.frames; ++f) most = tab.nKernels[f] > most ? tab.nKernels[f] : most;
	if (most <= 0) return hipSuccess;
	hipLaunchKernelGGL(kht_vote_kernel, dim3((most * 4 + 63) / 64, tab.frames), dim3(64), 0, stream, a, tab);
	return hipGetLastError();
}

hipError_t launch_kht_peaks(const KhtGpuArgs& a, const KhtBatchVote& tab, hipStream_t stream)
{
	// column coverage of the reference's scan (:1166-1187): SSE2 groups of 4 from column 1 while rho_index < rhoN-3, then a
	// scalar remainder that starts at (rhoN & ~3) + 1
	const int simd = a.rhoN > 4;
	int sseCovEnd = 1, consumed = a.rhoN + 1, remains = 0;
	if (simd) {
		const int sseEnd = a.rhoN - 3;
		const int iters = (sseEnd - 1 + 3) / 4; // ri = 1, 5, ... < sseEnd
		sseCovEnd = 1 + 4 * (iters > 0 ? iters : 0);
		consumed = (a.rhoN & ~3) + 1;
		remains = a.rhoN > consumed ? a.rhoN - consumed : 0;
	}
	dim3 grid((a.rhoN + 255) / 256, a.T > 1 ? a.T - 1 : 1, tab.frames);
	hipLaunchKernelGGL(kht_peaks_kernel, grid, dim3(256), 0, stream, a, tab, sseCovEnd, consumed, remains, simd);
	return hipGetLastError();
}

} // namespace compvhip

