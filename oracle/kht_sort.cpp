// TEST INFRASTRUCTURE ONLY -- the one step of CompVHoughKht that cannot be restated in C: the reference orders its vote
// cells with std::sort on the count alone (core/features/hough/compv_core_feature_houghkht.cxx:1195-1204).  std::sort is
// unstable but DETERMINISTIC for a given libstdc++ and input order, and the sweep that follows (:1207-1247) depends on the
// order inside equal-count groups.  Calling the same std::sort on the cells in the reference's emission order
// (theta-major, rho ascending, :1151-1192) therefore reproduces the reference's line set exactly on this toolchain.
#include "kht_oracle.h"

#include <algorithm>

extern "C" void orc_kht_sort_cells(orc_kht_cell* cells, size_t n)
{
	std::sort(cells, cells + n, [](const orc_kht_cell& a, const orc_kht_cell& b) { return a.count > b.count; });
}
