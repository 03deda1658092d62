// kht.hpp -- types shared by the host stages (kht_host.cpp), the GPU stages (kht_kernels.hip) and the C ABI (api.cpp) of
// the kernel-based Hough transform.  Reference: core/features/hough/compv_core_feature_houghkht.{h,cxx}.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace compvhip {

struct KhtAxes { double dRho, dThetaRad, dThetaDeg, r; size_t rhoN, T, W, H; };
struct KhtRange { size_t begin, end; };                     // CompVHoughKhtString / Cluster
struct KhtKernel { double rho, theta, h, sigmaThetaSquare, sigmaRhoSquare, m2, sigmaRhoTimesTheta; }; // CompVHoughKhtKernel (:52-62)
struct KhtLine { float rho, theta; int32_t strength, rhoIndex, thetaIndex; };

// per-kernel constants of vote_Algorithm4, precomputed on the host (all divisions / square roots)
struct KhtVoteParams { double srsScale, stsScale, sScale, r2, x, y; unsigned rhoIndex, thetaIndex; };
// one vote cell that passed the 3x3 smoothing + threshold; `order` = position in the reference's emission order = thetaIndex * 2 (rhoN + 2) + rhoIndex, or
// ... + (rhoN + 2) + rhoIndex for the cells the reference's scalar remainder pushes (quirk Q6): the indices are recovered on the host (8 bytes per cell cross PCIe)
struct KhtCell { uint32_t order; int32_t count; };
static_assert(sizeof(KhtCell) == 8, "vote cells cross PCIe as 8-byte records");

bool khtAxes(size_t W, size_t H, float rho, float thetaDeg, KhtAxes& ax);
void khtFillAxes(const KhtAxes& ax, std::vector<double>& rho, std::vector<double>& theta);
struct KhtPoint { int16_t x, y; };   // W, H <= 32 767 (the reference keeps int16 coordinates too, canny hysteresis :628); 4 bytes per point: half the upload of the strings
static_assert(sizeof(KhtPoint) == 4, "points cross PCIe as 4-byte records");
// the edge map of one frame, one bit per pixel, with a zero border: row y (-1 <= y <= H) starts at row(y) with one zero pad word (8 bytes); pixel x is bit 64 + x of the row
struct KhtBitPlane {
	std::vector<uint8_t> buf; size_t pitch = 0, W = 0, H = 0;
	void reset(size_t W, size_t H);
	uint8_t* row(int y) { return buf.data() + static_cast<size_t>(y + 1) * pitch; }
};
void khtPackBytes(const uint8_t* edges, size_t W, size_t H, size_t S, KhtBitPlane& plane);
void khtPlaneFromWords(const uint32_t* words, size_t wordsPerRow, size_t W, size_t H, KhtBitPlane& plane);
// Appendix A: strings of linked pixels, in the reference's order; destroys the plane
size_t khtPlaneCount(const KhtBitPlane& plane);   // set pixels = the most points khtLink can write
// pts: room for khtPlaneCount(plane) points; returns the number of points written (string after string)
size_t khtLink(KhtBitPlane& plane, size_t minSize, KhtPoint* pts, std::vector<KhtRange>& strings);
void khtFinishKernels(std::vector<KhtKernel>& kernels, double& hmax);
double khtPruneAndScale(std::vector<KhtKernel>& kernels, double hmax, double minHeight);
void khtVoteParams(const KhtAxes& ax, const std::vector<KhtKernel>& kernels, std::vector<KhtVoteParams>& params);
// buffers of the peak stage that survive from frame to frame (one per KhtScratch)
struct KhtPeaksWork {
	struct Rec { int32_t count; uint32_t pos; };
	struct Idx { uint32_t rho, theta; };
	std::vector<KhtCell> tmp; std::vector<Rec> recs; std::vector<Idx> idx; std::vector<uint8_t> visited; std::vector<double> rho, theta;
	size_t axW = 0, axH = 0; double axRho = 0.0, axTheta = 0.0;
};
void khtPeaks(const KhtAxes& ax, std::vector<KhtCell>& cells, int maxLines, std::vector<KhtLine>& lines, KhtPeaksWork& work);

// GPU stages.  Every launch covers the frames of a BATCH (compvhip_plan_houghkht: one launch per stage for up to kKhtBatch frames instead of one per frame;
// the host entry point is a batch of one): strings, clusters and kernels of the frames sit one behind the other in shared arrays, the per-frame tables
// travel by value in the kernel arguments (blockIdx.y / .z = frame).
constexpr int kKhtBatch = 32;      // frames a launch can cover (size of the by-value tables)
constexpr int kKhtGroup = 8;       // frames compvhip_plan_houghkht puts through the stages together
constexpr int kKhtMaxInFlight = 8; // groups in flight at most (controllers: own stream, buffers and bit planes each)
struct KhtBatchStrings { uint32_t stringBegin[kKhtBatch + 1]; uint32_t clusterBase[kKhtBatch]; int frames; };   // strings of frame f: [stringBegin[f], stringBegin[f + 1]); its clusters start at clusterBase[f]
struct KhtBatchStats { uint32_t clusterBase[kKhtBatch]; int n[kKhtBatch]; int simdEnd[kKhtBatch]; int frames; };   // simdEnd: clusters [0, simdEnd) of the frame use the SIMD operation order of the kernel height
struct KhtBatchVote { uint32_t paramsBase[kKhtBatch]; int nKernels[kKhtBatch]; double gs[kKhtBatch]; int frames; size_t mapElems, cellCap; };   // vote map / cell list of frame f at f * mapElems / f * cellCap
struct KhtGpuArgs {
	const KhtVoteParams* params; int nKernels;
	int32_t* counts;          // [frames][(T+2) x stride], zeroed
	int stride;               // >= rhoN + 2
	int rhoN, T;
	double dRho, dThetaDeg, gs;
	int32_t threshold;
	KhtCell* cells; int* cellCount; int cellCap;   // [frames][cellCap], [frames]
};
// Algorithm 2, per-cluster statistics (kht_stats_kernel): one thread per cluster, float64, the reference's operation order
struct KhtSpan { uint32_t begin, end; };
struct KhtStatsArgs {
	const KhtPoint* pts; const KhtSpan* clusters;
	double hw, hh;            // W/2, H/2
	KhtKernel* out;           // cluster order; .theta holds vx: acos() is taken by the host libm (khtFinishKernels)
};
hipError_t launch_kht_stats(const KhtStatsArgs& a, const KhtBatchStats& tab, hipStream_t stream);

// clusters_find / clusters_subdivision (houghkht.cxx:762-832) on the GPU: one thread per string, explicit recursion stack
struct KhtStringDesc { uint32_t begin, end, slot; }; // points [begin, end) of the string; slot = first cluster / stack slot of its private regions
struct KhtSubdivFrame { int32_t s, e, m, keep, state; double ratio, rl; };
struct KhtSubdivArgs {
	const KhtPoint* pts; const KhtStringDesc* strings; int nStrings;
	int minSize; double minDev;
	KhtSpan* scratch;          // per-string cluster regions (capacity: see khtSubdivSlots)
	KhtSubdivFrame* stack;     // per-string recursion stacks (same slot layout, + 2 frames per string)
	uint32_t* counts;          // [nStrings] clusters found per string
	KhtSpan* clusters;         // compacted, string order, frame f from clusterBase[f]
	uint32_t* total;           // [frames + 1]: clusters found per frame; [frames] = flag "a recursion ran out of stack slots" (zeroed by the caller)
	int flagIndex;             // = frames
};
// upper bound on the clusters (and on the recursion depth) of a string of `len` points
__host__ __device__ inline size_t khtSubdivSlots(size_t len, size_t minSize) { const size_t m = minSize < 2 ? 2 : minSize; return (len > m ? (len - 1) / (m - 1) : 1) + 2; }
hipError_t launch_kht_subdivide(const KhtSubdivArgs& a, const KhtBatchStrings& tab, hipStream_t stream);
hipError_t launch_kht_vote(const KhtGpuArgs& a, const KhtBatchVote& tab, hipStream_t stream);
hipError_t launch_kht_peaks(const KhtGpuArgs& a, const KhtBatchVote& tab, hipStream_t stream);

} // namespace compvhip
