// TEST INFRASTRUCTURE ONLY -- the one step of CompVHoughKht that cannot be restated in C: the reference orders its vote
// cells with std::sort on the count alone (core/features/hough/compv_core_feature_houghkht.cxx:1195-1204).  std::sort is
// unstable but DETERMINISTIC for a given libstdc++ and input order, and the sweep that follows (:1207-1247) depends on the
// order inside equal-count groups.  Calling the same std::sort on the cells in the reference's emission order
// (theta-major, rho ascending, :1151-1192) therefore reproduces the reference's line set exactly on this toolchain.
#include "kht_oracle.h"
#include "compv_oracle.h"

#include <algorithm>

extern "C" void orc_kht_sort_cells(orc_kht_cell* cells, size_t n)
{
	std::sort(cells, cells + n, [](const orc_kht_cell& a, const orc_kht_cell& b) { return a.count > b.count; });
}

// The same for CompVHoughSht (core/features/hough/compv_core_feature_houghsht.cxx:241-249): nms_apply pushes the surviving cells row by
// row, column by column (:546-562, intrin/x86/compv_core_feature_houghsht_intrin_sse2.cxx:60-100; with several threads the per-thread
// vectors are concatenated in row order, :228-234), then std::sort orders them by strength alone and the first maxLines are kept.
// orc_sht_lines() returns the canonical order (strength, then cell); this puts `n` lines -- ALL lines of the frame, not a truncated
// set -- into the order the reference built with this libstdc++ returns them in.
extern "C" void orc_sht_reference_order(orc_line* lines, size_t n)
{
	std::sort(lines, lines + n, [](const orc_line& a, const orc_line& b) { return a.row != b.row ? a.row < b.row : a.col < b.col; }); // emission order
	std::sort(lines, lines + n, [](const orc_line& a, const orc_line& b) { return a.strength > b.strength; });
}
