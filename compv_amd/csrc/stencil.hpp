// stencil.hpp -- what the tile kernels of the gradient stage share (gfx950, wave64): the XCD-aware tile placement.
// (The register-resident 32-bit row rings of the first kernel generation -- Grad3Ring / Grad5State, 512x64 tiles, 8 px per lane -- lived here until the
// last kernel that used them, the 5x5 Canny tile kernel, moved to the packed u16 arithmetic of canny_swar_kernels.hip in round 5.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace compvhip {

// XCD-aware tile mapping.  Workgroup b is observed to run on XCD b % 8, each XCD with a private L2.  Horizontally adjacent
// tiles share their 4-byte column halos (one extra 128-byte line each side per row), so all tilesX tiles of one "row group"
// (same frame, same block-row) are given to the SAME XCD, consecutively: the halo lines then hit in that XCD's L2 instead
// of being fetched from the fabric twice.  A pure performance remap: any placement is correct.
// grid = 8 * ceil(groups/8) * tilesX workgroups (1-D); returns false for padding workgroups.
__device__ __forceinline__ bool xcd_tile_map(int b, int tilesX, int groups, int& tileX, int& group)
{
	const int xcd = b & 7;
	const int k = b >> 3;             // index of this workgroup inside its XCD's queue
	group = (k / tilesX) * 8 + xcd;
	tileX = k - (k / tilesX) * tilesX;
	return group < groups;
}

} // namespace compvhip
