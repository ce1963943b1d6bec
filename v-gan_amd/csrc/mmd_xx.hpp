// X-X tiles of the Gram (block sums only: they feed the reported loss, never a gradient) as a job that rides in ANOTHER launch.
//
// They depend on nothing the step produces: their operand is the batch's rows of the data set, whose centred split images
// (Dh, Dl) and norms (dsq) are prepared ONCE per fit; the batch is gathered by index (the epoch's table + the device-side step
// counter) while the tiles are staged.  Taken out of the Gram launch they leave it with 392 instead of 528 tiles at n = 1024 --
// one round on the 512 resident slots instead of two: 24.2 -> 17.5 us.  MEASURED on MI355X: inside the mask / projection launch
// the tiles are NOT free -- their operand, gathered from the data set's images, is HBM-cold (inside the Gram it is the L2-hot
// image the forward has just written), the K loop turns latency-bound and the carrying launch grows from 9.4 to 18.3 us: a net
// loss of ~1.5 % of the step rate.  Running them on a side stream was slower still (profiles/r02_overlap_schedules.txt).  The
// job is therefore opt-in (VGAN_XX_RIDE=1 in v-gan_amd/trainer.py); kept because it is the piece a warm-operand carrier needs.
#pragma once
#include "gemm_bf3.hpp"
#include "mmd_common.hpp"

namespace vgan {

struct XXJob {
    const unsigned short *Dh, *Dl;   // [rows of the data set, ldd] split images of (data - centre); pad columns zero
    const float* dsq;                // their squared norms (of the split values)
    const int* rows;                 // epoch table of batch indices, or nullptr: batch row i = data row i
    const unsigned long long* cursor;
    const TileDesc* tiles;           // the X-X tiles (r0, c0 < n)
    const float* bw;
    float* partial;                  // [ntiles][4]
    int ldd, row_batches, row_stride, ntiles, n;
};

// One tile per workgroup on 256 threads.  The host launch may use larger workgroups (the mask / projection kernel's have 512
// threads): the surplus waves return at once -- s_barrier waits on the SURVIVING waves of a workgroup only, so the tile's
// barriers keep working.  ASSUMPTION, recorded here because the HIP programming model calls a barrier that not every thread of
// the block reaches undefined: it holds on gfx950 / ROCm 7.2 because an ended wave leaves the workgroup's barrier count, and it
// is exercised only by the 512- and 1024-thread carriers (mask_forward_bf3_kernel<.., XX>: opt-in; linear_bwd_params_ks_xx: the
// late X-X tiles when the MMD backward launch has no room).  The DEFAULT carrier, mmd_backward_bf3_kernel<64, true>, has exactly
// 256 threads: no wave returns early there.  Each carrier has a GPU parity test in the default tier
// (test_backward_bf3_rowmajor_operand_equals_transposed_operand, test_xx_tiles_*, the VGAN_GRAM_SLOTS-forced step tests); should a
// compiler change break the early return, let the surplus waves idle THROUGH the tile's barriers instead.  One tile per CU is the point: alone on a CU a tile takes ~7 us, two sharing one ~15 (measured with
// tile PAIRS per workgroup: the carrying launch went from 9.4 to 18.2 us).  lds: GemmBF3<64>::kLdsBytes.
__device__ __forceinline__ void xx_tile_body(const XXJob& job, int t, char* lds, float* red /* [4] */) {
    using G = GemmBF3<64>;
    if (threadIdx.x >= kBlock) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TileDesc td = job.tiles[t];
    const int* map = job.rows;
    if (map != nullptr) map += (long)(job.cursor ? (long)(job.cursor[0] % (unsigned long long)job.row_batches) : 0l) * job.row_stride;
    auto row_of = [&](int g) { return map ? map[g] : g; };
    const int j = td.c0 + G::sub_col();
    const bool jok = j < td.clim;
    const float sj = job.dsq[row_of(min(j, td.clim - 1))];
    const float bw = job.bw[0];
    float si_pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) si_pre[r] = job.dsq[row_of(min(td.r0 + G::sub_row(r), td.rlim - 1))];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    G::template run<false>(job.Dh, job.Dl, job.ldd, job.Dh, job.Dl, job.ldd, td.r0, td.c0, td.rlim, td.clim, job.ldd, lds, nullptr, acc, map,
                           map);
    const float c2 = -1.4426950408889634f / (4.f * bw);
    float ksum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = td.r0 + G::sub_row(r);
        const bool ok = jok && (i < td.rlim);
        const float L = fmaxf(si_pre[r] + sj - 2.f * acc[r], 0.f);
        const float tt = __builtin_amdgcn_exp2f(L * c2);
        const float t2 = tt * tt, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
        ksum += ok ? ((tt + t2) + (t4 + t8)) + t16 : 0.f;
    }
    ksum = wave_sum(ksum);
    if (lane == 0) red[wave] = ksum;
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<float4*>(job.partial)[t] = make_float4((red[0] + red[1]) + (red[2] + red[3]), 0.f, 0.f, 0.f);
}

}  // namespace vgan
