/* TEST INFRASTRUCTURE ONLY -- see kht_oracle.c. */
#ifndef KHT_ORACLE_H
#define KHT_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_kht_axes_t { double dRho, dTheta_rad, dTheta_deg, r; size_t rhoN, T, W, H; } orc_kht_axes_t;
typedef struct orc_kht_pos { int y, x; double cy, cx; } orc_kht_pos;
typedef struct orc_kht_range { size_t begin, end; } orc_kht_range;
typedef struct orc_kht_kernel { double rho, theta, h, sigma_theta_square, sigma_rho_square, m2, sigma_rho_times_theta; } orc_kht_kernel;
typedef struct orc_kht_cell { size_t rho_index, theta_index; int32_t count; } orc_kht_cell;
typedef struct orc_kht_line { float rho, theta; int32_t strength, rho_index, theta_index; } orc_kht_line;

int orc_kht_axes(size_t W, size_t H, float rho, float thetaDeg, orc_kht_axes_t* ax);
void orc_kht_fill_axes(const orc_kht_axes_t* ax, double* rho, double* theta);
int orc_kht_link(const uint8_t* edges, size_t W, size_t H, size_t S, size_t minSize, orc_kht_pos** poss, size_t* nposs, orc_kht_range** strings, size_t* nstrings);
int orc_kht_clusters(const orc_kht_pos* poss, const orc_kht_range* strings, size_t nstrings, size_t minSize, double minDev, orc_kht_range** clusters, size_t* nclusters);
int orc_kht_kernels(const orc_kht_pos* poss, const orc_kht_range* clusters, size_t n, orc_kht_kernel* kernels, double* hmax);
int orc_kht_prune_gs(orc_kht_kernel* kernels, size_t* n, double hmax, double minHeight, double* gs);
int orc_kht_vote(const orc_kht_axes_t* ax, const orc_kht_kernel* kernels, size_t n, double gs, int32_t* counts, size_t stride);
int orc_kht_peak_votes(const orc_kht_axes_t* ax, const int32_t* counts, size_t stride, int32_t threshold, orc_kht_cell** votes, size_t* nvotes);
int orc_kht_peak_lines(const orc_kht_axes_t* ax, const orc_kht_cell* votes, size_t nvotes, int maxLines, orc_kht_line* lines, size_t cap, size_t* nlines);
int orc_kht(const uint8_t* edges, size_t W, size_t H, size_t S, float rho, float thetaDeg, int32_t threshold, int maxLines,
            double minDev, size_t minSize, double minHeight, orc_kht_line* lines, size_t cap, size_t* nlines, double* gs);
void orc_free(void* p);
#ifdef __cplusplus
}
#endif
#endif
