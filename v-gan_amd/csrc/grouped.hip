// Grouped small-GEMM launch: up to four independent row-major products in ONE launch.
//
// The collapsed generator (src/models/Generator.py:61-66 has no activation between its Linear layers) is a chain of
// small matrix products; each launch of a dependent chain costs ~5 us whatever its size, so products that do not depend
// on each other share a launch.  Per problem the library picks the tile engine: 32x32 tiles with K split over the waves
// (GemmTileKS) when the 64x64 grid would be a handful of long-K tiles, 64x64x32 tiles otherwise.
#include "gemm_core.hpp"

namespace vgan {

constexpr int QBM = 64, QBK = 32, QKS = 128;

struct GroupedArgs {
    vgan_gemm_problem p[VGAN_GEMM_MAX_GROUP];
    int tile_start[VGAN_GEMM_MAX_GROUP + 1];
    int ks[VGAN_GEMM_MAX_GROUP];
    int count;
};

template <int LA, int LB, int VEC>
__device__ __forceinline__ void tile64(const vgan_gemm_problem& q, int t, float* lds) {
    using G = GemmTile<QBM, QBM, QBK, LA, LB, VEC>;
    const int gx = (q.n + QBM - 1) / QBM;
    const int m0 = (t / gx) * QBM, n0 = (t % gx) * QBM;
    f32x16 acc[1][1];
    zero_acc(acc);
    G::template run<false>(q.a, q.lda, q.b, q.ldb, m0, n0, q.m, q.n, q.k, lds, nullptr, acc);
    const int col = n0 + G::sub_col(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + G::sub_row(0, r);
        if (row < q.m && col < q.n) q.c[(long)row * q.ldc + col] = acc[0][0][r];
    }
}

template <int LA, int LB, int VEC, int NW = 4>
__device__ __forceinline__ void tile_ks(const vgan_gemm_problem& q, int t, float* lds) {
    using G = GemmTileKS<QKS, LA, LB, VEC, NW>;
    const int gx = (q.n + 31) / 32;
    const int m0 = (t / gx) * 32, n0 = (t % gx) * 32;
    float o[G::NR];
    G::run(q.a, q.lda, q.b, q.ldb, m0, n0, q.m, q.n, q.k, lds, o);
    const int col = n0 + G::col_of();
#pragma unroll
    for (int rr = 0; rr < G::NR; ++rr) {
        const int row = m0 + G::row_of(rr);
        if (row < q.m && col < q.n) q.c[(long)row * q.ldc + col] = o[rr];
    }
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int VEC>
__global__ __launch_bounds__(kBlock, 2) void gemm_grouped_kernel(GroupedArgs g) {
    constexpr int kLds = cmax(cmax(GemmTile<QBM, QBM, QBK, KC, MC, VEC>::kLdsFloats, GemmTile<QBM, QBM, QBK, MC, MC, VEC>::kLdsFloats),
                              cmax(cmax(GemmTileKS<QKS, KC, MC, VEC>::kLdsFloats, GemmTileKS<QKS, MC, MC, VEC>::kLdsFloats),
                                   cmax(GemmTile<QBM, QBM, QBK, KC, KC, VEC>::kLdsFloats, GemmTileKS<QKS, KC, KC, VEC>::kLdsFloats)));
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    int qi = 0;  // block-uniform
#pragma unroll
    for (int i = 1; i < VGAN_GEMM_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.tile_start[i]) qi = i;
    const vgan_gemm_problem& q = g.p[qi];
    const int t = blockIdx.x - g.tile_start[qi];
    // operand images: KC = contraction index contiguous ([mn][K]), MC = output index contiguous ([K][mn])
    if (q.kind == VGAN_GEMM_NN) {         // C = A[m,k] . B[k,n]
        if (g.ks[qi]) tile_ks<KC, MC, VEC>(q, t, lds); else tile64<KC, MC, VEC>(q, t, lds);
    } else if (q.kind == VGAN_GEMM_NT) {  // C = A[m,k] . B[n,k]^T
        if (g.ks[qi]) tile_ks<KC, KC, VEC>(q, t, lds); else tile64<KC, KC, VEC>(q, t, lds);
    } else {                              // C = A[k,m]^T . B[k,n]
        if (g.ks[qi]) tile_ks<MC, MC, VEC>(q, t, lds); else tile64<MC, MC, VEC>(q, t, lds);
    }
}

// Every problem on 32x32 tiles with 16 waves splitting K (1024-thread workgroups): launches whose products all have a long
// contraction, where the per-wave MFMA chain is the critical path.
__global__ __launch_bounds__(1024, 1) void gemm_grouped_ks16_kernel(GroupedArgs g) {
    constexpr int kLds = cmax(cmax(GemmTileKS<QKS, KC, MC, 4, 16>::kLdsFloats, GemmTileKS<QKS, MC, MC, 4, 16>::kLdsFloats),
                              GemmTileKS<QKS, KC, KC, 4, 16>::kLdsFloats);
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    int qi = 0;
#pragma unroll
    for (int i = 1; i < VGAN_GEMM_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.tile_start[i]) qi = i;
    const vgan_gemm_problem& q = g.p[qi];
    const int t = blockIdx.x - g.tile_start[qi];
    if (q.kind == VGAN_GEMM_NN) tile_ks<KC, MC, 4, 16>(q, t, lds);
    else if (q.kind == VGAN_GEMM_NT) tile_ks<KC, KC, 4, 16>(q, t, lds);
    else tile_ks<MC, MC, 4, 16>(q, t, lds);
}

}  // namespace vgan

using namespace vgan;

extern "C" int vgan_gemm_grouped(const vgan_gemm_problem* problems, int count, vgan_stream_t stream) {
    VGAN_CHECK_ARG(problems && count >= 1 && count <= VGAN_GEMM_MAX_GROUP);
    GroupedArgs g{};
    g.count = count;
    bool vec = true;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        const vgan_gemm_problem& q = problems[i];
        VGAN_CHECK_ARG(q.a && q.b && q.c && q.m > 0 && q.n > 0 && q.k > 0 && q.ldc >= q.n);
        VGAN_CHECK_ARG(q.kind == VGAN_GEMM_NN || q.kind == VGAN_GEMM_NT || q.kind == VGAN_GEMM_TN);
        VGAN_CHECK_ARG(q.lda >= (q.kind == VGAN_GEMM_TN ? q.m : q.k) && q.ldb >= (q.kind == VGAN_GEMM_NT ? q.k : q.n));
        vec = vec && (q.m % 4 == 0) && (q.n % 4 == 0) && (q.k % 4 == 0) && (q.lda % 4 == 0) && (q.ldb % 4 == 0) && aligned16(q.a) &&
              aligned16(q.b);
        const long t64 = (long)((q.m + 63) / 64) * ((q.n + 63) / 64);
        // a long contraction over few tiles: the K loop is the critical path -> 32x32 tiles with K split over the waves, as
        // long as they still fit the chip in one round (2 workgroups per CU)
        const long t32 = (long)((q.m + 31) / 32) * ((q.n + 31) / 32);
        g.ks[i] = (q.k >= 128 && t32 <= 512) ? 1 : 0;
        g.p[i] = q;
        g.tile_start[i] = tiles;
        tiles += g.ks[i] ? ((q.m + 31) / 32) * ((q.n + 31) / 32) : (int)t64;
    }
    for (int i = count; i <= VGAN_GEMM_MAX_GROUP; ++i) g.tile_start[i] = tiles;
    // all products long-K and few tiles: one 1024-thread launch with every problem on 32x32 tiles
    int kmin = problems[0].k, t32 = 0;
    for (int i = 0; i < count; ++i) {
        kmin = problems[i].k < kmin ? problems[i].k : kmin;
        t32 += ((problems[i].m + 31) / 32) * ((problems[i].n + 31) / 32);
    }
    if (vec && kmin >= 96 && t32 <= 256) {
        int acc = 0;
        for (int i = 0; i < count; ++i) {
            g.ks[i] = 1;
            g.tile_start[i] = acc;
            acc += ((problems[i].m + 31) / 32) * ((problems[i].n + 31) / 32);
        }
        for (int i = count; i <= VGAN_GEMM_MAX_GROUP; ++i) g.tile_start[i] = acc;
        hipLaunchKernelGGL(gemm_grouped_ks16_kernel, dim3(acc), dim3(1024), 0, (hipStream_t)stream, g);
        VGAN_CHECK_LAUNCH();
        return VGAN_OK;
    }
    if (vec)
        hipLaunchKernelGGL(gemm_grouped_kernel<4>, dim3(tiles), dim3(kBlock), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(gemm_grouped_kernel<1>, dim3(tiles), dim3(kBlock), 0, (hipStream_t)stream, g);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}
