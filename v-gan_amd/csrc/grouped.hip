// Grouped small-GEMM launch: up to four independent row-major products in ONE launch.
//
// The collapsed generator (src/models/Generator.py:61-66 has no activation between its Linear layers) is a chain of
// small matrix products; each launch of a dependent chain costs ~5 us whatever its size, so products that do not depend
// on each other share a launch.  Per problem the library picks the tile engine: 32x32 tiles with K split over the waves
// (GemmTileKS) when the 64x64 grid would be a handful of long-K tiles, 64x64x32 tiles otherwise.
#include "gemm_core.hpp"
#include "mmd_common.hpp"
#include "optim_common.hpp"

namespace vgan {

constexpr int QBM = 64, QBK = 32, QKS = 128;

struct GroupedArgs {
    vgan_gemm_problem p[VGAN_GEMM_MAX_GROUP];
    int tile_start[VGAN_GEMM_MAX_GROUP + 1];
    int ks[VGAN_GEMM_MAX_GROUP];
    int split[VGAN_GEMM_MAX_GROUP];   // K slices of the problem (vgan_gemm_problem.splitk): slice s writes slab s of C
    int kchunk[VGAN_GEMM_MAX_GROUP];  // contraction indices per slice (a multiple of QBK)
    int count;
};

// One K slice of a split problem as a problem of its own: operands advanced along the contraction, output = slab `slice`.
__device__ __forceinline__ vgan_gemm_problem k_slice(vgan_gemm_problem q, int slice, int kchunk) {
    const int k0 = slice * kchunk;
    q.c += (long)slice * q.m * q.ldc;
    if (q.kind == VGAN_GEMM_NN) { q.a += k0; q.b += (long)k0 * q.ldb; }
    else if (q.kind == VGAN_GEMM_NT) { q.a += k0; q.b += k0; }
    else { q.a += (long)k0 * q.lda; q.b += (long)k0 * q.ldb; }
    q.k = min(kchunk, q.k - k0);
    return q;
}

// What may ride in a grouped launch besides its products (vgan_gemm_grouped_ex): a plain copy, the Adadelta update as
// the products' epilogue (+ one element-wise layer whose gradient already sits in memory), and the next step's noise draw.
struct GroupedExtras {
    const float* copy_src;
    float* copy_dst;
    long copy_count;
    int copy_blocks, extra_blocks, noise_blocks, adadelta;
    float *p, *sq, *acc;
    float lr, rho, eps, wd, gs;
    vgan_adadelta_layer layer[VGAN_GEMM_MAX_GROUP + 1];
    const float* g_extra;
    int ld_extra;
    float* z;
    int zrows, zcols, zld, zones;
    unsigned long long seed;
    const unsigned long long* step_counter;
    int fold_blocks;
    vgan_finalize_job fold;
};

// the optimiser step of one element of a packed gradient image [dW | db]: (row, col) -> flat parameter index
__device__ __forceinline__ void adadelta_packed_element(const GroupedExtras& x, const vgan_adadelta_layer& L, int row, int col, float g) {
    if (row >= L.out || col > L.in) return;
    const long idx = col < L.in ? L.off_w + (long)row * L.in + col : L.off_b + row;
    float pv = x.p[idx], v = x.sq[idx], a = x.acc[idx];
    adadelta_one(pv, g, v, a, x.lr, x.rho, x.eps, x.wd, x.gs);
    x.p[idx] = pv;
    x.sq[idx] = v;
    x.acc[idx] = a;
    L.w_packed[(long)row * L.ldp + col] = pv;
}

// The same for the NE elements a lane holds of an output tile (one column, NE rows), in two phases: the optimiser state is
// REQUESTED before the product's main loop (it does not depend on the product; requested after it the launch pays one more
// memory latency per tile -- measured 11.5 us for the launch instead of the product's ~6) and ALL loads precede the first store
// (the pointers may alias as far as the compiler knows: element-by-element code serialises NE round trips, 16.2 us).
template <int NE>
struct AdadeltaTile {
    long idx[NE];
    float pv[NE], v[NE], a[NE];
    bool ok[NE];
    template <typename RowOf>
    __device__ __forceinline__ void prefetch(const GroupedExtras& x, const vgan_adadelta_layer& L, int col, RowOf row_of) {
#pragma unroll
        for (int r = 0; r < NE; ++r) {
            const int row = row_of(r);
            ok[r] = row < L.out && col <= L.in;
            idx[r] = ok[r] ? (col < L.in ? L.off_w + (long)row * L.in + col : L.off_b + row) : L.off_w;  // clamped: unconditional loads
            pv[r] = x.p[idx[r]];
            v[r] = x.sq[idx[r]];
            a[r] = x.acc[idx[r]];
        }
    }
    template <typename RowOf>
    __device__ __forceinline__ void finish(const GroupedExtras& x, const vgan_adadelta_layer& L, int col, const float (&g)[NE], RowOf row_of) {
#pragma unroll
        for (int r = 0; r < NE; ++r) adadelta_one(pv[r], g[r], v[r], a[r], x.lr, x.rho, x.eps, x.wd, x.gs);
#pragma unroll
        for (int r = 0; r < NE; ++r)
            if (ok[r]) {
                x.p[idx[r]] = pv[r];
                x.sq[idx[r]] = v[r];
                x.acc[idx[r]] = a[r];
                L.w_packed[(long)row_of(r) * L.ldp + col] = pv[r];
            }
    }
};

// workgroups past the product tiles: copy | element-wise layer | noise (block-uniform dispatch)
__device__ __forceinline__ void grouped_extra_jobs(const GroupedExtras& x, int b, int layer_index) {
    if (b < x.copy_blocks) {
        for (long i = (long)b * blockDim.x + threadIdx.x; i < x.copy_count; i += (long)x.copy_blocks * blockDim.x) x.copy_dst[i] = x.copy_src[i];
        return;
    }
    b -= x.copy_blocks;
    if (b < x.extra_blocks) {
        const vgan_adadelta_layer& L = x.layer[layer_index];
        const long total = (long)L.out * (L.in + 1);
        for (long i = (long)b * blockDim.x + threadIdx.x; i < total; i += (long)x.extra_blocks * blockDim.x) {
            const int row = (int)(i / (L.in + 1)), col = (int)(i % (L.in + 1));
            adadelta_packed_element(x, L, row, col, x.g_extra[(long)row * x.ld_extra + col]);
        }
        return;
    }
    b -= x.extra_blocks;
    if (b < x.noise_blocks) {
        noise_normal_body(x.z, x.zrows, x.zcols, x.zld, x.zones, x.seed, x.step_counter, 0ull, b, x.noise_blocks);
        return;
    }
    b -= x.noise_blocks;
    if (b < x.fold_blocks) finalize_body(x.fold);
}

template <int LA, int LB, int VEC, bool EPI>
__device__ __forceinline__ void tile64(const vgan_gemm_problem& q, int t, float* lds, const GroupedExtras& x, int qi) {
    using G = GemmTile<QBM, QBM, QBK, LA, LB, VEC>;
    const int gx = (q.n + QBM - 1) / QBM;
    const int m0 = (t / gx) * QBM, n0 = (t % gx) * QBM;
    const int col = n0 + G::sub_col(0);
    auto row_of = [&](int r) { return m0 + G::sub_row(0, r); };
    AdadeltaTile<EPI ? 16 : 1> upd;
    if constexpr (EPI) upd.prefetch(x, x.layer[qi], col, row_of);
    f32x16 acc[1][1];
    zero_acc(acc);
    G::template run<false>(q.a, q.lda, q.b, q.ldb, m0, n0, q.m, q.n, q.k, lds, nullptr, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + G::sub_row(0, r);
        if (row < q.m && col < q.n) q.c[(long)row * q.ldc + col] = acc[0][0][r];
    }
    if constexpr (EPI) {
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = acc[0][0][r];
        upd.finish(x, x.layer[qi], col, g, row_of);
    }
}

template <int LA, int LB, int VEC, int NW = 4, bool EPI = false>
__device__ __forceinline__ void tile_ks(const vgan_gemm_problem& q, int t, float* lds, const GroupedExtras& x, int qi) {
    using G = GemmTileKS<QKS, LA, LB, VEC, NW>;
    const int gx = (q.n + 31) / 32;
    const int m0 = (t / gx) * 32, n0 = (t % gx) * 32;
    const int col = n0 + G::col_of();
    auto row_of = [&](int rr) { return m0 + G::row_of(rr); };
    AdadeltaTile<EPI ? G::NR : 1> upd;
    if constexpr (EPI) upd.prefetch(x, x.layer[qi], col, row_of);
    float o[G::NR];
    G::run(q.a, q.lda, q.b, q.ldb, m0, n0, q.m, q.n, q.k, lds, o);
#pragma unroll
    for (int rr = 0; rr < G::NR; ++rr) {
        const int row = m0 + G::row_of(rr);
        if (row < q.m && col < q.n) q.c[(long)row * q.ldc + col] = o[rr];
    }
    if constexpr (EPI) upd.finish(x, x.layer[qi], col, o, row_of);
}

// C tile = (A[m0 .., :k] . B[:k2, :k]^T) . D[n0 .., :k2]^T: the tile's 64 rows of H = A . B^T go through the problem's scratch
// region (global memory, L2-hot: written and re-read by this workgroup only, behind a __syncthreads, whose fence drains the
// stores), then the second product reads them as its A operand.
template <int VEC>
__device__ __forceinline__ void tile64_two_stage(const vgan_gemm_problem& q, int t, float* lds) {
    using G = GemmTile<QBM, QBM, QBK, KC, KC, VEC>;
    const int gx = (q.n + QBM - 1) / QBM;
    const int m0 = (t / gx) * QBM, n0 = (t % gx) * QBM;
    const int ldh = (q.k2 + 3) / 4 * 4;
    float* H = q.scratch + (long)t * QBM * ldh;
    const int rows = min(QBM, q.m - m0);
    for (int c0 = 0; c0 < q.k2; c0 += QBM) {
        f32x16 acc[1][1];
        zero_acc(acc);
        G::template run<false>(q.a, q.lda, q.b, q.ldb, m0, c0, q.m, q.k2, q.k, lds, nullptr, acc);
        const int col = c0 + G::sub_col(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = G::sub_row(0, r);
            if (row < rows && col < ldh) H[(long)row * ldh + col] = col < q.k2 ? acc[0][0][r] : 0.f;
        }
    }
    __syncthreads();
    f32x16 acc[1][1];
    zero_acc(acc);
    G::template run<false>(H, ldh, q.d, q.ldd, 0, n0, rows, q.n, q.k2, lds, nullptr, acc);
    const int col = n0 + G::sub_col(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + G::sub_row(0, r);
        if (row < q.m && col < q.n) q.c[(long)row * q.ldc + col] = acc[0][0][r];
    }
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int VEC, bool EPI>
__global__ __launch_bounds__(kBlock, 2) void gemm_grouped_kernel(GroupedArgs g, GroupedExtras x) {
    constexpr int kLds = cmax(cmax(GemmTile<QBM, QBM, QBK, KC, MC, VEC>::kLdsFloats, GemmTile<QBM, QBM, QBK, MC, MC, VEC>::kLdsFloats),
                              cmax(cmax(GemmTileKS<QKS, KC, MC, VEC>::kLdsFloats, GemmTileKS<QKS, MC, MC, VEC>::kLdsFloats),
                                   cmax(GemmTile<QBM, QBM, QBK, KC, KC, VEC>::kLdsFloats, GemmTileKS<QKS, KC, KC, VEC>::kLdsFloats)));
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    if ((int)blockIdx.x >= g.tile_start[VGAN_GEMM_MAX_GROUP]) {  // block-uniform: the jobs riding behind the product tiles
        grouped_extra_jobs(x, blockIdx.x - g.tile_start[VGAN_GEMM_MAX_GROUP], g.count);
        return;
    }
    int qi = 0;  // block-uniform
#pragma unroll
    for (int i = 1; i < VGAN_GEMM_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.tile_start[i]) qi = i;
    vgan_gemm_problem q = g.p[qi];
    int t = blockIdx.x - g.tile_start[qi];
    if (g.split[qi] > 1) {  // (block-uniform) tiles of slice 0, then of slice 1, ...
        const int per = (g.tile_start[qi + 1] - g.tile_start[qi]) / g.split[qi];
        q = k_slice(q, t / per, g.kchunk[qi]);
        t %= per;
    }
    // operand images: KC = contraction index contiguous ([mn][K]), MC = output index contiguous ([K][mn])
    if (q.kind == VGAN_GEMM_NN) {         // C = A[m,k] . B[k,n]
        if (g.ks[qi]) tile_ks<KC, MC, VEC, 4, EPI>(q, t, lds, x, qi); else tile64<KC, MC, VEC, EPI>(q, t, lds, x, qi);
    } else if (q.kind == VGAN_GEMM_NT) {  // C = A[m,k] . B[n,k]^T
        if (g.ks[qi]) tile_ks<KC, KC, VEC, 4, EPI>(q, t, lds, x, qi); else tile64<KC, KC, VEC, EPI>(q, t, lds, x, qi);
    } else if (q.kind == VGAN_GEMM_NT_NT) {  // C = (A[m,k] . B[k2,k]^T) . D[n,k2]^T
        tile64_two_stage<VEC>(q, t, lds);
    } else {                              // C = A[k,m]^T . B[k,n]
        if (g.ks[qi]) tile_ks<MC, MC, VEC, 4, EPI>(q, t, lds, x, qi); else tile64<MC, MC, VEC, EPI>(q, t, lds, x, qi);
    }
}

// Every problem on 32x32 tiles with 16 waves splitting K (1024-thread workgroups): launches whose products all have a long
// contraction, where the per-wave MFMA chain is the critical path.
__global__ __launch_bounds__(1024, 1) void gemm_grouped_ks16_kernel(GroupedArgs g, GroupedExtras x) {
    constexpr int kLds = cmax(cmax(GemmTileKS<QKS, KC, MC, 4, 16>::kLdsFloats, GemmTileKS<QKS, MC, MC, 4, 16>::kLdsFloats),
                              GemmTileKS<QKS, KC, KC, 4, 16>::kLdsFloats);
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    if ((int)blockIdx.x >= g.tile_start[VGAN_GEMM_MAX_GROUP]) {  // (this variant carries the copy job only)
        grouped_extra_jobs(x, blockIdx.x - g.tile_start[VGAN_GEMM_MAX_GROUP], g.count);
        return;
    }
    int qi = 0;
#pragma unroll
    for (int i = 1; i < VGAN_GEMM_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.tile_start[i]) qi = i;
    const vgan_gemm_problem& q = g.p[qi];
    const int t = blockIdx.x - g.tile_start[qi];
    if (q.kind == VGAN_GEMM_NN) tile_ks<KC, MC, 4, 16>(q, t, lds, x, qi);
    else if (q.kind == VGAN_GEMM_NT) tile_ks<KC, KC, 4, 16>(q, t, lds, x, qi);
    else tile_ks<MC, MC, 4, 16>(q, t, lds, x, qi);
}

}  // namespace vgan

using namespace vgan;

static inline int extra_grid(long work_items) {
    long g = (work_items + kBlock - 1) / kBlock;
    return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}

extern "C" int vgan_gemm_grouped_ex(const vgan_gemm_problem* problems, int count, const vgan_grouped_extras* extras,
                                    vgan_stream_t stream) {
    VGAN_CHECK_ARG(problems && count >= 1 && count <= VGAN_GEMM_MAX_GROUP);
    GroupedArgs g{};
    GroupedExtras x{};
    g.count = count;
    bool vec = true, any_split = false;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        const vgan_gemm_problem& q = problems[i];
        VGAN_CHECK_ARG(q.a && q.b && q.c && q.m > 0 && q.n > 0 && q.k > 0 && q.ldc >= q.n);
        VGAN_CHECK_ARG(q.kind == VGAN_GEMM_NN || q.kind == VGAN_GEMM_NT || q.kind == VGAN_GEMM_TN || q.kind == VGAN_GEMM_NT_NT);
        const bool two = q.kind == VGAN_GEMM_NT_NT;
        VGAN_CHECK_ARG(q.lda >= (q.kind == VGAN_GEMM_TN ? q.m : q.k) && q.ldb >= ((q.kind == VGAN_GEMM_NT || two) ? q.k : q.n));
        if (two) {
            VGAN_CHECK_ARG(q.d && q.scratch && q.k2 > 0 && q.ldd >= q.k2 && q.splitk <= 1 && aligned16(q.scratch));
            vec = vec && (q.k2 % 4 == 0) && (q.ldd % 4 == 0) && aligned16(q.d);
            any_split = true;  // (keeps the launch on the 64 x 64 tile kernel, without the optimiser epilogue)
        }
        vec = vec && (q.m % 4 == 0) && (q.n % 4 == 0) && (q.k % 4 == 0) && (q.lda % 4 == 0) && (q.ldb % 4 == 0) && aligned16(q.a) &&
              aligned16(q.b);
        const long t64 = (long)((q.m + 63) / 64) * ((q.n + 63) / 64);
        // a long contraction over few tiles: the K loop is the critical path -> 32x32 tiles with K split over the waves, as
        // long as they still fit the chip in one round (2 workgroups per CU)
        const long t32 = (long)((q.m + 31) / 32) * ((q.n + 31) / 32);
        g.ks[i] = (q.k >= 128 && t32 <= 512 && !two) ? 1 : 0;
        g.p[i] = q;
        g.tile_start[i] = tiles;
        // split-K across WORKGROUPS (splitk > 1): a product with a long contraction and too few 64 x 64 tiles to load the chip
        // evenly (c5: M_3 = Wt_4^T M_4 is 165 tiles of K = 4100) is cut into splitk slices of the contraction; slice s writes
        // its partial product to slab s of C (slabs m * ldc floats apart) and the caller sums them in fixed order
        // (vgan_reduce_slabs) -- no atomics, so replicas of a data-parallel run stay bit-identical.
        g.split[i] = q.splitk > 1 ? q.splitk : 1;
        if (g.split[i] > 1) {
            g.kchunk[i] = ((q.k + g.split[i] - 1) / g.split[i] + QBK - 1) / QBK * QBK;
            VGAN_CHECK_ARG(g.split[i] <= 64 && (long)(g.split[i] - 1) * g.kchunk[i] < q.k);  // every slice holds work
            g.ks[i] = 0;
            any_split = true;
        }
        tiles += g.ks[i] ? ((q.m + 31) / 32) * ((q.n + 31) / 32) : (int)t64 * g.split[i];
    }
    for (int i = count; i <= VGAN_GEMM_MAX_GROUP; ++i) g.tile_start[i] = tiles;
    bool epi = false;
    if (extras != nullptr) {
        const vgan_grouped_extras& e = *extras;
        if (e.copy_src != nullptr) {
            VGAN_CHECK_ARG(e.copy_dst && e.copy_count > 0);
            x.copy_src = e.copy_src;
            x.copy_dst = e.copy_dst;
            x.copy_count = (long)e.copy_count;
            x.copy_blocks = extra_grid(e.copy_count);
        }
        if (e.adadelta) {
            VGAN_CHECK_ARG(e.p && e.sq_avg && e.acc_delta && !any_split);  // (the optimiser epilogue needs the whole product)
            epi = true;
            x.adadelta = 1;
            x.p = e.p; x.sq = e.sq_avg; x.acc = e.acc_delta;
            x.lr = e.lr; x.rho = e.rho; x.eps = e.eps; x.wd = e.weight_decay; x.gs = e.grad_scale;
            const int nl = count + (e.g_extra != nullptr ? 1 : 0);
            for (int i = 0; i < nl; ++i) {
                const vgan_adadelta_layer& L = e.layer[i];
                VGAN_CHECK_ARG(L.w_packed && L.out > 0 && L.in > 0 && L.ldp >= L.in + 1 && L.off_w >= 0 && L.off_b >= 0);
                if (i < count) VGAN_CHECK_ARG(problems[i].m >= L.out && problems[i].n >= L.in + 1);
                x.layer[i] = L;
            }
            if (e.g_extra != nullptr) {
                VGAN_CHECK_ARG(e.ld_extra >= e.layer[count].in + 1);
                x.g_extra = e.g_extra;
                x.ld_extra = e.ld_extra;
                x.extra_blocks = extra_grid((long)e.layer[count].out * (e.layer[count].in + 1));
            }
        }
        if (e.next_noise != nullptr) {
            VGAN_CHECK_ARG(e.noise_rows > 0 && e.noise_cols > 0 && e.noise_ld >= e.noise_cols && e.noise_ones_col < e.noise_ld);
            x.z = e.next_noise;
            x.zrows = e.noise_rows; x.zcols = e.noise_cols; x.zld = e.noise_ld; x.zones = e.noise_ones_col;
            x.seed = (unsigned long long)e.seed;
            x.step_counter = reinterpret_cast<const unsigned long long*>(e.step_counter);
            x.noise_blocks = extra_grid(((long)e.noise_rows * e.noise_cols + 3) / 4);
        }
        if (e.fold != nullptr) {
            VGAN_CHECK_ARG(finalize_job_ok(*e.fold));
            x.fold = *e.fold;
            x.fold_blocks = 1;
        }
    }
    const int surplus = x.copy_blocks + x.extra_blocks + x.noise_blocks + x.fold_blocks;
    // all products long-K and few tiles: one 1024-thread launch with every problem on 32x32 tiles (no optimiser epilogue there)
    int kmin = problems[0].k, t32 = 0;
    for (int i = 0; i < count; ++i) {
        kmin = problems[i].k < kmin ? problems[i].k : kmin;
        t32 += ((problems[i].m + 31) / 32) * ((problems[i].n + 31) / 32);
    }
    if (vec && kmin >= 96 && t32 <= 256 && !epi && x.noise_blocks == 0 && !any_split) {  // (copy and fold jobs ride in this variant too)
        int acc = 0;
        for (int i = 0; i < count; ++i) {
            g.ks[i] = 1;
            g.tile_start[i] = acc;
            acc += ((problems[i].m + 31) / 32) * ((problems[i].n + 31) / 32);
        }
        for (int i = count; i <= VGAN_GEMM_MAX_GROUP; ++i) g.tile_start[i] = acc;
        hipLaunchKernelGGL(gemm_grouped_ks16_kernel, dim3(acc + surplus), dim3(1024), 0, (hipStream_t)stream, g, x);
        VGAN_CHECK_LAUNCH();
        return VGAN_OK;
    }
    const dim3 grid(tiles + surplus), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (vec && epi) hipLaunchKernelGGL((gemm_grouped_kernel<4, true>), grid, block, 0, st, g, x);
    else if (vec) hipLaunchKernelGGL((gemm_grouped_kernel<4, false>), grid, block, 0, st, g, x);
    else if (epi) hipLaunchKernelGGL((gemm_grouped_kernel<1, true>), grid, block, 0, st, g, x);
    else hipLaunchKernelGGL((gemm_grouped_kernel<1, false>), grid, block, 0, st, g, x);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_gemm_grouped(const vgan_gemm_problem* problems, int count, vgan_stream_t stream) {
    return vgan_gemm_grouped_ex(problems, count, nullptr, stream);
}
