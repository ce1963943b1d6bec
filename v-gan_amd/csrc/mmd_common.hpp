// Types shared by the fp32 (mmd.hip) and split-bf16 (mmd_bf16.hip) MMD kernels.
#pragma once
#include "vgan_common.hpp"

namespace vgan {

struct TileDesc {
    int r0, c0, rlim, clim, flags, pad0, pad1, pad2;
};
static_assert(sizeof(TileDesc) == VGAN_TILE_INTS * 4, "tile descriptor layout");

// Column arg-max job that may ride in the Gram launch: workgroups with blockIdx.x >= ntiles each do one (64-column,
// 64-row-chunk) cell of it.  It is independent of the Gram, tiny, and the Gram grid (528 tiles at n = 1024) leaves most
// CUs idle during its last third, so this removes a launch from the step's critical path for free.
struct ColmaxJob {
    const float* S;
    unsigned long long* part;
    int lds, row_offset, n, d, from_softmax, nbx;  // nbx = ceil(d / 64); job is empty when S == nullptr
};

}  // namespace vgan
