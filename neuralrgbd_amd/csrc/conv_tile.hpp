// conv_tile.hpp — pieces shared by the matrix-core convolution kernels (conv3d.hip, conv2d.hip):
// the swizzled LDS image of a channels-last halo tile and the MFMA-row <-> pixel mapping.
#pragma once
#include "common.hpp"

namespace nrgbd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCB = 16;   // channels per K block
constexpr int kSV = kCB;  // LDS voxel stride (floats): 64 B, XOR-swizzled

// LDS image of a halo tile: [voxel][4 x 16 B], the 16-B slot s of voxel v stored at slot s ^ ((v >> 2) & 3).
// A ds_read_b128 lane group reads one logical slot of 16 consecutive voxels: voxels v, v+4, v+8, v+12 share a
// bank quad at a 64-B stride, the swizzle sends them to 4 different slots -> conflict-free without padding.
__device__ __forceinline__ int lds_slot(int voxel, int slot) { return voxel * kSV + ((slot ^ ((voxel >> 2) & 3)) << 2); }

// position of MFMA row i (0..31) inside its 2-row x 16-x patch: rows of lane group G0 = {0-3,12-15,20-27}
// take x = 0..15 of the first row, G1 = {4-11,16-19,28-31} of the second row
__device__ __forceinline__ void row_to_yx(int i, int& dy, int& x) {
    dy = ((i >= 4 && i < 12) || (i >= 16 && i < 20) || (i >= 28)) ? 1 : 0;
    x = (i < 4) ? i : (i < 12) ? i - 4 : (i < 16) ? i - 8 : (i < 20) ? i - 8 : (i < 28) ? i - 12 : i - 16;
}

// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  With xcd != 0 the launch order is
// re-mapped so that XCD k works on the k-th contiguous eighth of the tile list: neighbouring tiles (which share
// their halo voxels) then hit the same L2 instead of each XCD fetching the halo from HBM again.
__device__ __forceinline__ int xcd_tile(int bid, int nwg, int xcd) {
    if (!xcd) return bid;
    const int x = bid & 7, chunk = nwg >> 3, rem = nwg & 7;
    return x * chunk + (x < rem ? x : rem) + (bid >> 3);
}

}  // namespace nrgbd
