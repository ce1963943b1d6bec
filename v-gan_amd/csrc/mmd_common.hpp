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

// K split of the LAST tiles of a wide Gram launch (vgan_mmd_gram_bf3, tile = 256): one 768-thread workgroup holds a CU, so a
// table of T tiles runs in ceil(T / CUs) rounds and a short last round idles most of the chip for a whole tile time (c4: 1 040
// tiles on 256 CUs = 4 rounds + 16 tiles).  Tiles [first, ntiles) are therefore computed by `parts` workgroups each, part q over
// K columns [q kchunk, (q + 1) kchunk); the partial products meet in `slabs` (one 128 KB slab per part) and the last part to
// arrive at the tile's ticket finishes the tile.  first = ntiles, parts = 1: no split.
struct TailSplit {
    float* slabs;
    int* tickets;  // one per split tile, zero between launches (the finishing workgroup resets it)
    int first, parts, kchunk;
};
constexpr int kTailSlabs = 256;                       // at most this many parts per launch
constexpr long kTailSlabBytes = 256L * 128 * 4;       // one part's partial products
constexpr long kTailTicketBytes = 4096;               // tickets sit behind the slabs


// The single-rank step tail (vgan_mmd_finalize): partial[] -> stats[4]; colpart[chunks * d] -> colkey[d]; loss and its
// bookkeeping.  Runs in one workgroup of any size up to 1024 threads: stand-alone (mmd_finalize_kernel) or as the
// surplus workgroup of a backward launch.
__device__ __forceinline__ void finalize_body(const vgan_finalize_job& job) {
    const float* __restrict__ partial = job.partial;
    const TileDesc* __restrict__ tiles = reinterpret_cast<const TileDesc*>(job.tiles);
    const unsigned long long* colpart = reinterpret_cast<const unsigned long long*>(job.colpart);
    unsigned long long* __restrict__ colkey = reinterpret_cast<unsigned long long*>(job.colkey);
    double* __restrict__ stats = job.stats;
    float* __restrict__ loss = job.loss;
    float* __restrict__ loss_accum = job.loss_accum;
    unsigned long long* __restrict__ step_counter = reinterpret_cast<unsigned long long*>(job.step_counter);
    const int ntiles = job.ntiles, chunks = job.chunks, n = job.n, d = job.d;
    const float weight = job.weight, accum_scale = job.accum_scale;
    // split tail (vgan_finalize_job.mode): 1 = everything over the tiles [0, ntiles_main) -- those of the Gram launch, which may
    // include some X-X tiles; 2 = the block sums of the late tiles [ntiles_main, ntiles) (X-X only) and the loss.  stats[3]
    // carries the loss-so-far and stats[0] the X-X sum so far from 1 to 2, in double precision.
    const int mode = job.mode;
    const int t_lo = mode == 2 ? job.ntiles_main : 0, t_hi = mode == 1 ? job.ntiles_main : ntiles;
    if (mode == 2) colpart = nullptr;

    __shared__ double red[16][5];  // up to 1024 threads
    double s[5] = {0, 0, 0, 0, 0};  // Sxx, Sxy, Syy, sumL, penalty
    for (int t = t_lo + threadIdx.x; t < t_hi; t += blockDim.x) {
        const int fl = tiles[t].flags;
        const double w = (fl & VGAN_TF_TWICE) ? 2.0 : 1.0;
        const float4 pv = reinterpret_cast<const float4*>(partial)[t];
        s[fl & VGAN_TF_SLOT_MASK] += w * (double)pv.x;
        s[3] += (((fl & VGAN_TF_SLOT_MASK) == 1) ? 2.0 : w) * (double)pv.y;
    }
    if (colpart != nullptr) {
        for (int j = threadIdx.x; j < d; j += blockDim.x) {
            unsigned long long b = 0ull;
            for (int c0 = 0; c0 < chunks; c0 += 16) {  // 16 independent loads in flight (this block is latency-bound)
                unsigned long long k[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) k[e] = colpart[(long)min(c0 + e, chunks - 1) * d + j];
#pragma unroll
                for (int e = 0; e < 16; ++e) b = k[e] > b ? k[e] : b;
            }
            colkey[j] = b;
            s[4] += 1.0 - (double)colkey_value(b);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        s[q] = wave_sum(s[q]);
        if (lane == 0) red[wave][q] = s[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5] = {0, 0, 0, 0, 0};
        const int nw = blockDim.x >> 6;
        for (int w = 0; w < nw; ++w)
            for (int q = 0; q < 5; ++q) t[q] += red[w][q];
        const double nn = (double)n * (double)n;
        if (mode == 1) {
            stats[0] = t[0];
            stats[1] = t[1];
            stats[2] = t[2];
            stats[3] = (t[0] - 2.0 * t[1] + t[2]) / nn + (colpart ? (double)weight * t[4] / (double)d : 0.0);
            if (step_counter) step_counter[0] += 1ull;
        } else {
            double v;
            if (mode == 2) {
                stats[0] += t[0];
                v = stats[3] + t[0] / nn;
            } else {
                for (int q = 0; q < 4; ++q) stats[q] = t[q];
                v = (t[0] - 2.0 * t[1] + t[2]) / nn + (colpart ? (double)weight * t[4] / (double)d : 0.0);
                if (step_counter) step_counter[0] += 1ull;
            }
            loss[0] = (float)v;
            if (loss_accum) loss_accum[0] += (float)(v * (double)accum_scale);
        }
    }
}

inline bool finalize_job_ok(const vgan_finalize_job& j) {
    return j.partial && j.tiles && j.ntiles > 0 && j.stats && j.loss && j.n > 0 && j.d > 0 &&
           (j.colpart == nullptr || (j.colkey != nullptr && j.chunks > 0)) && j.mode >= 0 && j.mode <= 2 &&
           (j.mode == 0 || (j.ntiles_main >= 0 && j.ntiles_main <= j.ntiles));
}

}  // namespace vgan
