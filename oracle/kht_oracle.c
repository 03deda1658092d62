/* TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C, float64) of CompV's kernel-based Hough transform
 * (CompVHoughKht, core/features/hough/compv_core_feature_houghkht.cxx; citations relative to /root/reference).
 *
 * Pinned against the real CompV library (oracle/_ref) by tests/test_kht.py: GS (COMPV_HOUGHKHT_GET_FLT64_GS) to the last
 * digit and the complete line set (rho, theta, strength, order).  The reference std::sort()s the vote cells by count only
 * (unstable) before an order-dependent sweep; that order is reproduced by calling the same libstdc++
 * std::sort on the cells in the reference's emission order (oracle/kht_sort.cpp), which is deterministic on one toolchain.
 *
 * The stages are exposed separately so that the HIP path can be checked stage by stage.
 */
#include "kht_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float kPiF = 3.1415926535897932384626433f; /* kfMathTrigPi, base/math/compv_math.cxx:27 */
#define kPiOver180F (kPiF / 180.f)

/* ---- axes: initCoords, houghkht.cxx:501-541 ---------------------------------------------------------------- */
int orc_kht_axes(size_t W, size_t H, float rho, float thetaDeg, orc_kht_axes_t* ax)
{
	if (!W || !H || !(rho > 0.f) || rho > 1.f || !(thetaDeg > 0.f)) return -1;
	ax->dRho = (double)(rho * 1.f);
	ax->dTheta_rad = (double)(thetaDeg * kPiOver180F);
	ax->dTheta_deg = (ax->dTheta_rad * 180.0) / M_PI;               /* COMPV_MATH_RADIAN_TO_DEGREE */
	ax->r = sqrt((double)((W * W) + (H * H)));
	ax->rhoN = (size_t)((ax->r + 1.0) / ax->dRho);
	ax->T = (size_t)(180.0 / ax->dTheta_deg);
	ax->W = W; ax->H = H;
	return 0;
}

void orc_kht_fill_axes(const orc_kht_axes_t* ax, double* rho /*rhoN*/, double* theta /*T*/)
{
	double r0 = -(ax->r * 0.5);
	rho[0] = 0.0;  /* never written by the reference (:519) */
	for (size_t i = 1; i < ax->rhoN; ++i, r0 += ax->dRho) rho[i] = r0;
	r0 = 0.0;
	theta[0] = 0.0;
	for (size_t i = 1; i < ax->T; ++i, r0 += ax->dTheta_deg) theta[i] = r0;
}

/* ---- linking: Appendix A + Algorithms 5/6, houghkht.cxx:544-760 ---------------------------------------------- */
typedef struct { orc_kht_pos* p; size_t n, cap; } posvec;
typedef struct { orc_kht_range* p; size_t n, cap; } rngvec;

static int pos_push(posvec* v, int y, int x, double hh, double hw)
{
	if (v->n == v->cap) {
		size_t nc = v->cap ? v->cap * 2 : 4096;
		orc_kht_pos* np = (orc_kht_pos*)realloc(v->p, nc * sizeof(orc_kht_pos));
		if (!np) return -1;
		v->p = np; v->cap = nc;
	}
	orc_kht_pos* q = &v->p[v->n++];
	q->y = y; q->x = x; q->cx = (double)x - hw; q->cy = (double)y - hh;
	return 0;
}
static int rng_push(rngvec* v, size_t b, size_t e)
{
	if (v->n == v->cap) {
		size_t nc = v->cap ? v->cap * 2 : 1024;
		orc_kht_range* np = (orc_kht_range*)realloc(v->p, nc * sizeof(orc_kht_range));
		if (!np) return -1;
		v->p = np; v->cap = nc;
	}
	v->p[v->n].begin = b; v->p[v->n].end = e; v->n++;
	return 0;
}

/* Algorithm 6: next 8-neighbour in the fixed priority TL,T,TR, L,R, BL,B,BR (:666-703). Returns 0 when none. */
static int next_px(const uint8_t* e, size_t S, int W, int H, int* x, int* y)
{
	const int xs = *x, ys = *y;
	const int left = xs > 0, right = (xs + 1) < W;
	const uint8_t* c = e + (size_t)ys * S + xs;
	if (ys > 0) {
		const uint8_t* t = c - S;
		if (left && t[-1]) { *x = xs - 1; *y = ys - 1; return 1; }
		else if (*t) { *y = ys - 1; return 1; }
		else if (right && t[1]) { *x = xs + 1; *y = ys - 1; return 1; }
	}
	if (left && c[-1]) { *x = xs - 1; return 1; }
	else if (right && c[1]) { *x = xs + 1; return 1; }
	else if ((ys + 1) < H) {
		const uint8_t* b = c + S;
		if (left && b[-1]) { *x = xs - 1; *y = ys + 1; return 1; }
		else if (*b) { *y = ys + 1; return 1; }
		else if (right && b[1]) { *x = xs + 1; *y = ys + 1; return 1; }
	}
	return 0;
}

static int link_from(uint8_t* e, size_t S, int W, int H, int xr, int yr, size_t minSize, posvec* poss, rngvec* strings)
{
	const double hw = (double)W * 0.5, hh = (double)H * 0.5;
	const size_t begin = poss->n;
	int x = xr, y = yr;
	/* forward: add pixels to the end of the string, zeroing them (:724-728) */
	for (;;) {
		if (pos_push(poss, y, x, hh, hw)) return -1;
		e[(size_t)y * S + x] = 0;
		if (!next_px(e, S, W, H, &x, &y)) break;
	}
	const size_t rev = poss->n;
	/* backward from the reference pixel (:733-746) */
	x = xr; y = yr;
	if (next_px(e, S, W, H, &x, &y)) {
		for (;;) {
			if (pos_push(poss, y, x, hh, hw)) return -1;
			e[(size_t)y * S + x] = 0;
			if (!next_px(e, S, W, H, &x, &y)) break;
		}
	}
	const size_t end = poss->n;
	if ((end - begin) >= minSize) {
		/* the forward part is reversed so that the string runs end to end (:751-755) */
		for (size_t i = begin, j = rev; i + 1 < j; ++i) { --j; orc_kht_pos t = poss->p[i]; poss->p[i] = poss->p[j]; poss->p[j] = t; }
		return rng_push(strings, begin, end);
	}
	poss->n = begin;
	return 0;
}

int orc_kht_link(const uint8_t* edges, size_t W, size_t H, size_t S, size_t minSize, orc_kht_pos** poss_out, size_t* nposs,
                 orc_kht_range** strings_out, size_t* nstrings)
{
	uint8_t* e = (uint8_t*)malloc(S * H);
	if (!e) return -1;
	memcpy(e, edges, S * H);
	posvec poss = { 0, 0, 0 }; rngvec strings = { 0, 0, 0 };
	/* raster scan rows 1..H-2, cols 1..W-2 (:552-556: maxi = cols-1, maxj = rows-1, both loops start at 1) */
	for (int y = 1; y < (int)H - 1; ++y) {
		for (int x = 1; x < (int)W - 1; ++x) {
			if (e[(size_t)y * S + x]) {
				if (link_from(e, S, (int)W, (int)H, x, y, minSize, &poss, &strings)) { free(e); free(poss.p); free(strings.p); return -1; }
			}
		}
	}
	free(e);
	*poss_out = poss.p; *nposs = poss.n; *strings_out = strings.p; *nstrings = strings.n;
	return 0;
}

/* ---- clusters: recursive subdivision, houghkht.cxx:762-832 ----------------------------------------------------- */
typedef struct { const orc_kht_pos* poss; rngvec* out; size_t minSize; double minDev; int err; } subdiv_ctx;

static double subdivide(subdiv_ctx* c, size_t sbegin, size_t s, size_t e)
{
	const size_t keep = c->out->n;
	const orc_kht_pos* P = c->poss + sbegin;
	const int diffx = P[s].x - P[e].x, diffy = P[s].y - P[e].y;
	const double length = sqrt((double)((diffx * diffx) + (diffy * diffy)));
	size_t maxi = s;
	int maxdev = 0;
	for (size_t i = s + 1; i < e; ++i) {
		const int dev = abs(((P[s].x - P[i].x) * diffy) - ((P[s].y - P[i].y) * diffx));
		if (dev > maxdev) { maxi = i; maxdev = dev; }
	}
	const double d = (double)maxdev / length;
	const double ratio = length / ((d < c->minDev) ? c->minDev : d);   /* std::max(dev/len, minDev) */
	if ((maxi - s + 1) >= c->minSize) {
		if ((e - maxi + 1) >= c->minSize) {
			const double rl = subdivide(c, sbegin, s, maxi);
			const double rr = subdivide(c, sbegin, maxi, e);
			if (rl > ratio || rr > ratio) return rl > rr ? rl : rr;
		}
	}
	c->out->n = keep;
	if (rng_push(c->out, sbegin + s, sbegin + e + 1)) c->err = -1;
	return ratio;
}

int orc_kht_clusters(const orc_kht_pos* poss, const orc_kht_range* strings, size_t nstrings, size_t minSize, double minDev,
                     orc_kht_range** clusters_out, size_t* nclusters)
{
	rngvec out = { 0, 0, 0 };
	subdiv_ctx c = { poss, &out, minSize, minDev, 0 };
	for (size_t k = 0; k < nstrings; ++k) subdivide(&c, strings[k].begin, 0, (strings[k].end - strings[k].begin) - 1);
	if (c.err) { free(out.p); return -1; }
	*clusters_out = out.p; *nclusters = out.n;
	return 0;
}

/* ---- 2x2 eigen decomposition: base/math/compv_math_eigen.cxx:285-342 (sort = norm = true) ----------------------- */
static void find2x2(const double A[4], double D[4], double Q[4])
{
	int norm = 1;
	const double trace = A[0] + A[3];
	const double trace_div2 = trace / 2.0;
	const double det = (A[0] * A[3]) - (A[1] * A[2]);
	const double sq = sqrt(((trace * trace) / 4.0) - det);
	D[1] = D[2] = 0.0;
	D[0] = trace_div2 + sq;
	D[3] = trace_div2 - sq;
	if (A[2] != 0) { Q[0] = D[0] - A[3]; Q[2] = A[2]; Q[1] = D[3] - A[3]; Q[3] = A[2]; }
	else if (A[1] != 0) { Q[0] = A[1]; Q[2] = D[0] - A[0]; Q[1] = A[1]; Q[3] = D[3] - A[0]; }
	else {
		norm = 0;
		if (A[3] != 0.0) { Q[0] = 0.0; Q[2] = 1.0; Q[1] = 1.0; Q[3] = 0.0; }
		else { Q[0] = 1.0; Q[2] = 0.0; Q[1] = 0.0; Q[3] = 1.0; }
	}
	if (norm) {
		const double m02 = 1.0 / sqrt(Q[0] * Q[0] + Q[2] * Q[2]);
		const double m13 = 1.0 / sqrt(Q[1] * Q[1] + Q[3] * Q[3]);
		Q[0] *= m02; Q[2] *= m02; Q[1] *= m13; Q[3] *= m13;
	}
	if (D[0] < D[3]) {
		double a = Q[0], b = Q[2];
		Q[0] = Q[1]; Q[2] = Q[3]; Q[1] = a; Q[3] = b;
		a = D[0]; D[0] = D[3]; D[3] = a;
	}
}

/* ---- kernels: Algorithm 2, houghkht.cxx:885-1026 ------------------------------------------------------------------
 * Kernel height: clusters [0, n & ~3) follow the AVX operation order 1/((sqrt(1-r^2)*s)*2pi)
 * (intrin/x86/compv_core_feature_houghkht_intrin_avx.cxx:42-63), the last n & 3 the C order 1/(2pi*s*sqrt(1-r^2))
 * (:849-883) -- quirk Q5; with fewer than 4 clusters the SSE2 path (pairs) applies, whose order equals the AVX one. */
int orc_kht_kernels(const orc_kht_pos* poss, const orc_kht_range* clusters, size_t n, orc_kht_kernel* kernels, double* hmax_out)
{
	static const double kTwoPi = 2.0 * M_PI, kRadToDeg = 180.0 / M_PI;
	double hmax = 0.0;
	const size_t pack = n >= 4 ? 4 : (n >= 2 ? 2 : 1);
	const size_t simdEnd = pack > 1 ? (n & ~(pack - 1)) : 0;
	for (size_t k = 0; k < n; ++k) {
		const orc_kht_pos* b = poss + clusters[k].begin;
		const size_t cnt = clusters[k].end - clusters[k].begin;
		const double n_scale = 1.0 / (double)cnt;
		double mx = 0, my = 0;
		for (size_t i = 0; i < cnt; ++i) { mx += b[i].cx; my += b[i].cy; }
		mx *= n_scale; my *= n_scale;
		double cxx = 0, cyy = 0, cxy = 0;
		for (size_t i = 0; i < cnt; ++i) {
			const double cx = b[i].cx - mx, cy = b[i].cy - my;
			cxx += cx * cx; cyy += cy * cy; cxy += cx * cy;
		}
		const double M[4] = { cxx, cxy, cxy, cyy };
		double D[4], Q[4];
		find2x2(M, D, Q);
		const double ux = Q[0], uy = Q[2];
		double vx = Q[1], vy = Q[3];
		if (vy < 0.0) { vx = -vx; vy = -vy; }
		orc_kht_kernel* K = &kernels[k];
		K->rho = (vx * mx) + (vy * my);
		K->theta = acos(vx) * kRadToDeg;
		const double sq = sqrt(1.0 - (vx * vx));
		const double M0 = -(ux * mx) - (uy * my);
		const double M2 = (sq == 0.0) ? 0.0 : ((ux / sq) * kRadToDeg);
		double r0 = 0.0;
		for (size_t i = 0; i < cnt; ++i) {
			const double r1 = (ux * (b[i].cx - mx)) + (uy * (b[i].cy - my));
			r0 += r1 * r1;
		}
		/* CompVHoughKhtKernelHeight_* */
		const double inv = 1.0 / r0;
		const double r1 = M0 * inv, r2 = M2 * inv;
		double srs = r1 * M0 + n_scale;
		const double srt = r1 * M2;
		const double m2 = r2 * M0;
		double sts = r2 * M2;
		if (sts == 0.0) sts = 0.1;
		srs *= 4.0; sts *= 4.0;
		const double s = sqrt(srs) * sqrt(sts);
		const double rr = srt / s;
		const double omr = 1.0 - (rr * rr);
		double h;
		if (k < simdEnd) h = 1.0 / ((sqrt(omr) * s) * kTwoPi);
		else h = 1.0 / (kTwoPi * s * sqrt(omr));
		K->sigma_rho_square = srs; K->sigma_rho_times_theta = srt; K->m2 = m2; K->sigma_theta_square = sts; K->h = h;
		/* std::max / _mm256_max_pd: a NaN height never replaces hmax */
		if (h > hmax) hmax = h;
	}
	*hmax_out = hmax;
	return 0;
}

/* (1 + x/1024)^1024, houghkht.cxx:77-88 */
static double exp_fast_small(double x)
{
	x = 1.0 + (x * (1.0 / 1024.0));
	x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
	return x;
}

/* __gauss_Eq15, houghkht.cxx:834-846 */
static double gauss_eq15(double rho, double theta, const orc_kht_kernel* k)
{
	const double s = sqrt(k->sigma_rho_square) * sqrt(k->sigma_theta_square);
	const double sscale = 1.0 / s;
	const double r = k->sigma_rho_times_theta * sscale;
	const double omr = 1.0 - (r * r);
	const double x = 1.0 / ((2.0 * M_PI) * s * sqrt(omr));
	const double y = 1.0 / (2.0 * omr);
	const double z = ((rho * rho) / k->sigma_rho_square) - (((r * 2.0) * rho * theta) * sscale) + ((theta * theta) / k->sigma_theta_square);
	return x * exp_fast_small(-z * y);
}

/* discard short kernels (:1029-1041) + Gmin (:1044-1062) + GS (:377). Compacts `kernels` in place. */
int orc_kht_prune_gs(orc_kht_kernel* kernels, size_t* n, double hmax, double minHeight, double* gs_out)
{
	const double hscale = 1.0 / hmax;
	size_t m = 0;
	for (size_t k = 0; k < *n; ++k) {
		if (!((kernels[k].h * hscale) < minHeight)) kernels[m++] = kernels[k];
	}
	*n = m;
	double Gmin = DBL_MAX;
	for (size_t k = 0; k < m; ++k) {
		const double M[4] = { kernels[k].sigma_rho_square, kernels[k].sigma_rho_times_theta, kernels[k].m2, kernels[k].sigma_theta_square };
		double D[4], Q[4];
		find2x2(M, D, Q);
		const double w = sqrt(D[3]);
		const double g = gauss_eq15(Q[1] * w, Q[3] * w, &kernels[k]);
		if (g < Gmin) Gmin = g;
	}
	const double inv = 1.0 / Gmin;
	*gs_out = (Gmin == 0.0) ? 1.0 : (inv > 1.0 ? inv : 1.0);
	return 0;
}

/* vote_Algorithm4, houghkht.cxx:1088-1148 */
static void vote4(int32_t* counts, size_t stride, const orc_kht_axes_t* ax, size_t rho_start_index, size_t theta_start_index,
                  double rho_start, double theta_start, int inc_rho_index, int inc_theta_index, double scale, const orc_kht_kernel* kr)
{
	const size_t rho_size = ax->rhoN, theta_size = ax->T;
	double inc_rho = ax->dRho * inc_rho_index;
	const double inc_theta = ax->dTheta_deg * inc_theta_index;
	const double srs_scale = 1.0 / kr->sigma_rho_square, sts_scale = 1.0 / kr->sigma_theta_square;
	const double s = sqrt(kr->sigma_rho_square) * sqrt(kr->sigma_theta_square);
	const double s_scale = 1.0 / s;
	const double r = kr->sigma_rho_times_theta * s_scale;
	const double omr = 1.0 - (r * r);
	const double r2 = r * 2.0;
	const double x = 1.0 / ((2.0 * M_PI) * s * sqrt(omr));
	const double y = 1.0 / (2.0 * omr);
	(void)inc_rho;
	inc_rho = ax->dRho * inc_rho_index; /* computed once, BEFORE any wrap-around flips inc_rho_index (:1092) */
	size_t theta_index = theta_start_index, theta_count = 0;
	double theta = theta_start, rho;
	do {
		if (!theta_index || theta_index > theta_size) {
			rho_start_index = (rho_size - rho_start_index) + 1;
			theta_index = theta_index ? 1 : theta_size;
			inc_rho_index = -inc_rho_index;
		}
		if (rho_start_index >= 1) {
			int32_t* pcount = counts + theta_index * stride;
			size_t rho_index = rho_start_index;
			rho = rho_start;
			const double w = (theta * theta) * sts_scale;
			const double k = r2 * theta * s_scale;
			double krho = k * rho;
			const double ki = k * inc_rho;
			double z = ((rho * rho) * srs_scale) - krho + w;
			int32_t votes;
			while ((rho_index <= rho_size) && (votes = (int32_t)(((x * exp_fast_small(-z * y)) * scale) + 0.5)) > 0) {
				pcount[rho_index] += votes;
				rho_index += inc_rho_index;
				rho += inc_rho;
				krho += ki;
				z = ((rho * rho) * srs_scale) - krho + w;
			}
			theta_index += inc_theta_index;
			theta += inc_theta;
		}
		else break;
	} while ((rho != rho_start) && (++theta_count < theta_size));
}

/* voting_Algorithm2_Count, houghkht.cxx:1065-1086. counts: (T+2) rows x stride (>= rhoN+2), zeroed by the caller. */
int orc_kht_vote(const orc_kht_axes_t* ax, const orc_kht_kernel* kernels, size_t n, double gs, int32_t* counts, size_t stride)
{
	const double rho_scale = 1.0 / ax->dRho, theta_scale = 1.0 / ax->dTheta_deg;
	const double rho_max_neg = -(ax->r * 0.5); /* m_rho[1] */
	for (size_t k = 0; k < n; ++k) {
		const orc_kht_kernel* kr = &kernels[k];
		const size_t ri = (size_t)(fabs((kr->rho - rho_max_neg) * rho_scale) + 0.5) + 1;
		const size_t ti = (size_t)(fabs(kr->theta * theta_scale) + 0.5) + 1;
		vote4(counts, stride, ax, ri, ti, 0.0, 0.0, 1, 1, gs, kr);
		vote4(counts, stride, ax, ri, ti - 1, 0.0, -ax->dTheta_deg, 1, -1, gs, kr);
		vote4(counts, stride, ax, ri - 1, ti, -ax->dRho, 0.0, -1, 1, gs, kr);
		vote4(counts, stride, ax, ri - 1, ti - 1, -ax->dRho, -ax->dTheta_deg, -1, -1, gs, kr);
	}
	return 0;
}

/* peaks: Section 3.4, houghkht.cxx:1151-1247,1282-1308 with the SSE2 scan (intrin_sse2.cxx:20-96) and quirk Q6 */
/* std::sort on the count alone, in the reference's emission order: oracle/kht_sort.cpp */
void orc_kht_sort_cells(orc_kht_cell* cells, size_t n);

static int32_t smooth3x3(const int32_t* c, size_t stride)
{
	const int32_t *t = c - stride, *b = c + stride;
	return t[-1] + (t[0] << 1) + t[1] + b[-1] + (b[0] << 1) + b[1] + (c[-1] << 1) + (c[0] << 2) + (c[1] << 1);
}

int orc_kht_peak_votes(const orc_kht_axes_t* ax, const int32_t* counts, size_t stride, int32_t threshold, orc_kht_cell** votes_out, size_t* nvotes)
{
	const size_t rhoN = ax->rhoN, T = ax->T;
	size_t cap = 4096, n = 0;
	orc_kht_cell* v = (orc_kht_cell*)malloc(cap * sizeof(orc_kht_cell));
	if (!v) return -1;
#define PUSH(ri, ti, cnt) do { if (n == cap) { cap *= 2; orc_kht_cell* nv = (orc_kht_cell*)realloc(v, cap * sizeof(orc_kht_cell)); if (!nv) { free(v); return -1; } v = nv; } \
	v[n].rho_index = (ri); v[n].theta_index = (ti); v[n].count = (cnt); ++n; } while (0)
	const int simd = rhoN > 4;
	const size_t sseEnd = simd ? (rhoN > 3 ? rhoN - 3 : 0) : 0;
	const size_t consumed = (simd ? (rhoN & ~(size_t)3) : rhoN) + 1;
	const size_t remains = rhoN > consumed ? rhoN - consumed : 0;
	for (size_t ti = 1; ti < T; ++ti) {
		const int32_t* row = counts + ti * stride;
		if (simd) {
			for (size_t ri = 1; ri < sseEnd; ri += 4) {
				for (size_t j = 0; j < 4; ++j) {
					if (row[ri + j] > 0) { /* _mm_cmpgt_epi32(count, 0) */
						const int32_t c = smooth3x3(row + ri + j, stride);
						if (c >= threshold) PUSH(ri + j, ti, c);
					}
				}
			}
			if (remains) { /* scalar remainder on an offset pointer, indices pushed RELATIVE to it (quirk Q6, :1180-1187) */
				const int32_t* off = row + consumed;
				for (size_t ri = 1; ri < remains; ++ri) {
					if (off[ri]) {
						const int32_t c = smooth3x3(off + ri, stride);
						if (c >= threshold) PUSH(ri, ti, c);
					}
				}
			}
		}
		else {
			for (size_t ri = 1; ri < rhoN; ++ri) {
				if (row[ri]) {
					const int32_t c = smooth3x3(row + ri, stride);
					if (c >= threshold) PUSH(ri, ti, c);
				}
			}
		}
	}
#undef PUSH
	orc_kht_sort_cells(v, n);
	*votes_out = v; *nvotes = n;
	return 0;
}

int orc_kht_peak_lines(const orc_kht_axes_t* ax, const orc_kht_cell* votes, size_t nvotes, int maxLines, orc_kht_line* lines, size_t cap, size_t* nlines)
{
	const size_t vs = ax->rhoN + 2;
	uint8_t* visited = (uint8_t*)calloc((ax->T + 2) * vs, 1);
	double* rho = (double*)malloc(ax->rhoN * sizeof(double));
	double* theta = (double*)malloc(ax->T * sizeof(double));
	if (!visited || !rho || !theta) { free(visited); free(rho); free(theta); return -1; }
	orc_kht_fill_axes(ax, rho, theta);
	size_t n = 0;
	for (size_t i = 0; i < nvotes; ++i) {
		uint8_t* p = visited + votes[i].theta_index * vs + votes[i].rho_index;
		const uint8_t *t = p - vs, *b = p + vs;
		const int bv = t[-1] || t[0] || t[1] || p[-1] || p[1] || b[-1] || b[0] || b[1];
		if (!bv) {
			if (n < cap) {
				lines[n].rho = (float)rho[votes[i].rho_index];
				lines[n].theta = (float)((theta[votes[i].theta_index] * M_PI) / 180.0); /* COMPV_MATH_DEGREE_TO_RADIAN */
				lines[n].strength = votes[i].count;
				lines[n].rho_index = (int32_t)votes[i].rho_index;
				lines[n].theta_index = (int32_t)votes[i].theta_index;
			}
			++n;
		}
		*p = 0xff;
	}
	free(visited); free(rho); free(theta);
	if (maxLines > 0 && n > (size_t)maxLines) n = (size_t)maxLines;
	*nlines = n;
	return 0;
}

/* whole CompVHoughKht::process (single-threaded branch, houghkht.cxx:208-447) */
int orc_kht(const uint8_t* edges, size_t W, size_t H, size_t S, float rho, float thetaDeg, int32_t threshold, int maxLines,
            double minDev, size_t minSize, double minHeight, orc_kht_line* lines, size_t cap, size_t* nlines, double* gs_out)
{
	orc_kht_axes_t ax;
	*nlines = 0;
	if (gs_out) *gs_out = 1.0;
	if (orc_kht_axes(W, H, rho, thetaDeg, &ax)) return -1;
	orc_kht_pos* poss = 0; orc_kht_range *strings = 0, *clusters = 0; size_t np = 0, ns = 0, nc = 0;
	orc_kht_kernel* kernels = 0; int32_t* counts = 0; orc_kht_cell* votes = 0; size_t nv = 0;
	int rc = -1;
	if (orc_kht_link(edges, W, H, S, minSize, &poss, &np, &strings, &ns)) goto done;
	rc = 0;
	if (!ns) goto done;
	if (orc_kht_clusters(poss, strings, ns, minSize, minDev, &clusters, &nc)) { rc = -1; goto done; }
	if (!nc) goto done;
	kernels = (orc_kht_kernel*)malloc(nc * sizeof(orc_kht_kernel));
	if (!kernels) { rc = -1; goto done; }
	double hmax, gs;
	orc_kht_kernels(poss, clusters, nc, kernels, &hmax);
	size_t nk = nc;
	orc_kht_prune_gs(kernels, &nk, hmax, minHeight, &gs);
	if (!nk) goto done;
	if (gs_out) *gs_out = gs;
	const size_t stride = ax.rhoN + 2;
	counts = (int32_t*)calloc((ax.T + 2) * stride, sizeof(int32_t));
	if (!counts) { rc = -1; goto done; }
	orc_kht_vote(&ax, kernels, nk, gs, counts, stride);
	if (orc_kht_peak_votes(&ax, counts, stride, threshold, &votes, &nv)) { rc = -1; goto done; }
	rc = orc_kht_peak_lines(&ax, votes, nv, maxLines, lines, cap, nlines);
done:
	free(poss); free(strings); free(clusters); free(kernels); free(counts); free(votes);
	return rc;
}

void orc_free(void* p) { free(p); }
