// RBF + MMDLossConstrained on gfx950 without ever materialising the 2n x 2n kernel matrix.
// Reference ops replaced: src/models/Mmd_loss_constrained.py:16-26 (cdist**2, bandwidth, 5x exp, sum)
// and :42-50 (vstack, block means, penalty), plus their autograd (EuclideanDistBackward, exp, mean).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <algorithm>
#include <array>
#include <cstring>
#include <vector>

#include "gemm_core.hpp"
#include "mmd_common.hpp"

namespace vgan {

constexpr int GT = VGAN_TILE;  // 64 x 64 Gram tile: one 32x32 MFMA sub-tile per wave
constexpr int GBK = 32;  // 128-byte row segments per staging load (one full L2 line)

// ---- Gram tile + fused kernel epilogue ----------------------------------------------------
// CALIB: only sum of L (first-call bandwidth).  Otherwise: sum of K = t + t^2 + t^4 + t^8 + t^16 with
// t = exp(-L / (4 bw)), i.e. sum_k exp(-L / (bw m_k)), m = {4, 2, 1, .5, .25}, and (optionally) the
// gradient weights Wg = sgn * (2/n^2) * dK/dL with dK/dL = -(1/bw) (t/4 + t^2/2 + t^4 + 2 t^8 + 4 t^16).
// GEN: an RBF with other than the reference's default (n_kernels = 5, mul_factor = 2) -- RBF(n_kernels, mul_factor),
// Mmd_loss_constrained.py:7-13: one exp per kernel, scales bw * mult[k] rounded to float32 like the reference's
// `bandwidth * bandwidth_multipliers`.  The default RBF keeps the one-exp squaring chain.
struct RbfMults {
    int nk;
    float mult[VGAN_RBF_MAX_KERNELS];
};

template <int VEC, bool CALIB, int KW, bool GEN = false>
__global__ __launch_bounds__(kBlock * KW, 2) void mmd_gram_kernel(const float* __restrict__ Z, int ldz, const float* __restrict__ sq,
                                                            int n, int p, const float* __restrict__ bw_ptr,
                                                            const TileDesc* __restrict__ tiles, int ntiles, float* __restrict__ Wg,
                                                            int ldw, int wrow0, float* __restrict__ partial, ColmaxJob cj,
                                                            RbfMults rm = RbfMults{}) {
    using G = GemmTile<GT, GT, GBK * KW, KC, KC, VEC, 0, KW>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    __shared__ float red[16];
    if ((int)blockIdx.x >= ntiles) {  // block-uniform
        const int cb = blockIdx.x - ntiles;
        colmax_partial_body<4 * KW>(cj.S, cj.lds, cj.row_offset, cj.part, cj.n, cj.d, cj.from_softmax, cb % cj.nbx, cb / cj.nbx);
        return;
    }
    const TileDesc td = tiles[blockIdx.x];
    // the epilogue's operands (row norms, bandwidth) are requested BEFORE the main loop: issued after it they would add one
    // full memory latency (~1 us) to every tile (measured on the split-bf16 kernels: -2 us Gram, -6 us backward)
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, wid = threadIdx.x >> 6;
    const int r_lo = G::first_reg();
    const int j = td.c0 + G::sub_col(0);
    const bool jok = j < td.clim;
    const float sj = sq[min(j, td.clim - 1)];
    float si_pre[G::kNumRegs];
#pragma unroll
    for (int rr = 0; rr < G::kNumRegs; ++rr) si_pre[rr] = sq[min(td.r0 + G::sub_row(0, r_lo + rr), td.rlim - 1)];
    float bw_pre = 1.f;
    if constexpr (!CALIB) bw_pre = bw_ptr[0];
    f32x16 acc[1][1];
    zero_acc(acc);
    G::template run<false>(Z, ldz, Z, ldz, td.r0, td.c0, td.rlim, td.clim, p, lds, nullptr, acc);

    float ksum = 0.f, lsum = 0.f;

    float c2 = 0.f, wscale = 0.f;
    float ck[GEN ? VGAN_RBF_MAX_KERNELS : 1], ik[GEN ? VGAN_RBF_MAX_KERNELS : 1];  // GEN: exp2 factor and 1/scale per kernel
    if constexpr (!CALIB) {
        const float bw = bw_pre;
        c2 = -1.4426950408889634f / (4.f * bw);  // exp(-L/(4bw)) = exp2(L * c2)
        const float sgn = (td.flags & VGAN_TF_NEG) ? -1.f : 1.f;
        wscale = -sgn * 2.f / ((float)n * (float)n * bw);
        if constexpr (GEN) {
            wscale = -sgn * 2.f / ((float)n * (float)n);
#pragma unroll
            for (int k = 0; k < VGAN_RBF_MAX_KERNELS; ++k) {
                const float scale = bw * rm.mult[k < rm.nk ? k : 0];
                ik[k] = 1.f / scale;
                ck[k] = -1.4426950408889634f / scale;
            }
        }
    }
    const bool store = (!CALIB) && (td.flags & VGAN_TF_STORE) && Wg != nullptr;
    const bool mirror = store && (td.flags & VGAN_TF_MIRROR);
    float wv[16];
#pragma unroll
    for (int rr = 0; rr < G::kNumRegs; ++rr) {
        const int r = r_lo + rr;
        const int i = td.r0 + G::sub_row(0, r);
        const bool ok = jok && (i < td.rlim);
        const float si = si_pre[rr];
        const float L = fmaxf(si + sj - 2.f * acc[0][0][r], 0.f);
        if constexpr (CALIB) {
            lsum += ok ? L : 0.f;
        } else {
            float K, w;
            if constexpr (GEN) {
                float dk = 0.f;
                K = 0.f;
#pragma unroll
                for (int k = 0; k < VGAN_RBF_MAX_KERNELS; ++k)
                    if (k < rm.nk) {
                        const float e = __builtin_amdgcn_exp2f(L * ck[k]);
                        K += e;
                        dk = fmaf(e, ik[k], dk);
                    }
                w = wscale * dk;
            } else {
                const float t = __builtin_amdgcn_exp2f(L * c2);  // v_exp_f32
                const float t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
                K = ((t + t2) + (t4 + t8)) + t16;
                w = wscale * (((0.25f * t + 0.5f * t2) + (t4 + 2.f * t8)) + 4.f * t16);
            }
            ksum += ok ? K : 0.f;
            wv[rr] = w;
            if (store && ok) Wg[(long)(i - wrow0) * ldw + j] = w;
        }
    }
    if constexpr (!CALIB) {
        if (mirror && jok) {  // Wg[j - wrow0, i] = w: rows (r&3) are 4 consecutive i -> one 16-byte store when aligned
            const int ibase = td.r0 + (wave >> 1) * (GT / 2) + 4 * (lane >> 5);
            float* dst = Wg + (long)(j - wrow0) * ldw;
            const bool v4 = ((ldw & 3) == 0) && ((td.r0 & 3) == 0) && ((reinterpret_cast<uintptr_t>(Wg) & 15) == 0);
#pragma unroll
            for (int qq = 0; qq < G::kNumRegs / 4; ++qq) {
                const int i0 = ibase + 8 * (r_lo / 4 + qq);
                if (v4 && i0 + 3 < td.rlim) {
                    *reinterpret_cast<float4*>(dst + i0) = make_float4(wv[4 * qq], wv[4 * qq + 1], wv[4 * qq + 2], wv[4 * qq + 3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (i0 + e < td.rlim) dst[i0 + e] = wv[4 * qq + e];
                }
            }
        }
    }
    ksum = wave_sum(ksum);
    lsum = wave_sum(lsum);
    if (lane == 0) {
        red[wid] = ksum;
        red[8 + wid] = lsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 o;
        o.x = (red[0] + red[1]) + (red[2] + red[3]);
        o.y = (red[8] + red[9]) + (red[10] + red[11]);
        if (KW == 2) {
            o.x += (red[4] + red[5]) + (red[6] + red[7]);
            o.y += (red[12] + red[13]) + (red[14] + red[15]);
        }
        // placement census (diagnostic only; never read by the product): HW_ID and XCC_ID of the wave that closed the tile
        o.z = __uint_as_float(__builtin_amdgcn_s_getreg((31 << 11) | 4));
        o.w = __uint_as_float(__builtin_amdgcn_s_getreg((31 << 11) | 20));
        reinterpret_cast<float4*>(partial)[blockIdx.x] = o;
    }
}

// ---- deterministic reduction of the per-tile partials into the four block statistics ------
__global__ void mmd_reduce_kernel(const float* __restrict__ partial, const TileDesc* __restrict__ tiles, int ntiles,
                                  double* __restrict__ stats, int zero_first) {
    __shared__ double red[4][4];
    double s[4] = {0, 0, 0, 0};
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        const int fl = tiles[t].flags;
        const double w = (fl & VGAN_TF_TWICE) ? 2.0 : 1.0;
        const float4 pv = reinterpret_cast<const float4*>(partial)[t];
        s[fl & VGAN_TF_SLOT_MASK] += w * (double)pv.x;
        // sum of L over the FULL Z x Z matrix: the XY block is computed once but occurs twice (XY and YX)
        s[3] += (((fl & VGAN_TF_SLOT_MASK) == 1) ? 2.0 : w) * (double)pv.y;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        s[q] = wave_sum(s[q]);
        if (lane == 0) red[wave][q] = s[q];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        double v = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        stats[q] = zero_first ? v : stats[q] + v;
    }
}

__global__ void mmd_set_bandwidth_kernel(const double* __restrict__ stats, int n, float* __restrict__ bw) {
    const double N = 2.0 * n;
    bw[0] = (float)(stats[3] / (N * N - N));
}

__global__ void mmd_loss_kernel(const double* __restrict__ stats, const unsigned long long* __restrict__ colkey, int n, int d,
                                float weight, float* __restrict__ loss, float* __restrict__ loss_accum, float accum_scale,
                                unsigned long long* __restrict__ step_counter) {
    __shared__ double red[4];
    double pen = 0.0;
    if (colkey != nullptr)
        for (int j = threadIdx.x; j < d; j += blockDim.x) pen += 1.0 - (double)colkey_value(colkey[j]);
    pen = wave_sum(pen);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pen;
    __syncthreads();
    if (threadIdx.x == 0) {
        pen = (red[0] + red[1]) + (red[2] + red[3]);
        const double nn = (double)n * (double)n;
        const double v = (stats[0] - 2.0 * stats[1] + stats[2]) / nn + (colkey ? (double)weight * pen / (double)d : 0.0);
        loss[0] = (float)v;
        if (loss_accum) loss_accum[0] += (float)(v * (double)accum_scale);
        if (step_counter) step_counter[0] += 1ull;
    }
}

// ---- single-GPU step tail in ONE launch: per-tile partials -> block statistics, column chunk keys -> arg-max keys,
// then the loss (replaces mmd_reduce + colmax_final + mmd_loss, three latency-bound launches).
__global__ __launch_bounds__(1024) void mmd_finalize_kernel(vgan_finalize_job job) { finalize_body(job); }

// ---- backward: dZ_i = 2 (rowsum(Wg_i) z_i - (Wg . Z)_i), optionally times mul ---------------
// A = Wg [nr, ncols] (KC), B(j = feature, k = Z row) = Z[k*ldz + j] (MC).
template <int VEC, int KW>
__global__ __launch_bounds__(kBlock * KW, 2) void mmd_backward_kernel(const float* __restrict__ Wg, int ldw, const float* __restrict__ Z,
                                                                int ldz, int wrow0, int nr, int ncols, int p,
                                                                const float* __restrict__ mul, int ldmul,
                                                                const float* __restrict__ mul_shift, float* __restrict__ out,
                                                                int ldo, int kchunk, long slab_stride, vgan_finalize_job job) {
    using G = GemmTile<GT, GT, GBK * KW, KC, MC, VEC, 0, KW>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    __shared__ float rs[GT];
    // XCD-aware tile order (speed only): block b runs on XCD b % 8; XCD x walks a contiguous chunk of a
    // traversal that goes down 4 row panels, then to the next column panel, so the tiles resident on one
    // XCD share their Wg row panels and Z column panels in its private L2.
    const int gx = (p + GT - 1) / GT, gy = (nr + GT - 1) / GT, total = gx * gy;
    if ((int)blockIdx.x >= total) {  // the one surplus workgroup column of the launch: the step tail (see vgan_finalize_job)
        if (blockIdx.y == 0) finalize_body(job);
        return;
    }
    const int xcd = blockIdx.x % 8, kidx = blockIdx.x / 8;
    const int q = total / 8, r = total % 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kidx;
    const int band = t / (4 * gx), rem = t - band * 4 * gx;
    const int rows_in_band = min(4, gy - band * 4);
    const int by = band * 4 + rem % rows_in_band, bx = rem / rows_in_band;
    const int m0 = by * GT, n0 = bx * GT;
    // split-K over the Z rows (blockIdx.y): slice s contributes 2 (rowsum_s(Wg) z - (Wg_s . Z_s)) -- the result is
    // linear in the slice sums -- to slab s of `out`; the consumer (mask backward) adds the slabs.  With one
    // workgroup per CU the K = 2n loop runs at one wave per SIMD; slices raise that to 2-4.
    const int k0 = blockIdx.y * kchunk;
    const int klen = min(kchunk, ncols - k0);
    out += blockIdx.y * slab_stride;
    // epilogue operands requested before the main loop (see mmd_gram_kernel)
    const int col = n0 + G::sub_col(0), colc = min(col, p - 1);
    const int r_lo = G::first_reg();
    float z_pre[G::kNumRegs], m_pre[G::kNumRegs];
    const float mshift = mul_shift != nullptr ? mul_shift[colc] : 0.f;  // mul is stored centred (see vgan_mmd_backward)
#pragma unroll
    for (int rr = 0; rr < G::kNumRegs; ++rr) {
        const int rowc = min(m0 + G::sub_row(0, r_lo + rr), nr - 1);
        z_pre[rr] = Z[(long)(wrow0 + rowc) * ldz + colc];
        m_pre[rr] = mul != nullptr ? mul[(long)rowc * ldmul + colc] + mshift : 1.f;
    }
    f32x16 acc[1][1];
    zero_acc(acc);
    if (klen > 0) {
        G::template run<true>(Wg + k0, ldw, Z + (long)k0 * ldz, ldz, m0, n0, nr, p, klen, lds, rs, acc);
    } else {
        if (threadIdx.x < GT) rs[threadIdx.x] = 0.f;
        __syncthreads();
    }
    if (col >= p) return;
#pragma unroll
    for (int rr = 0; rr < G::kNumRegs; ++rr) {
        const int r = r_lo + rr;
        const int lrow = G::sub_row(0, r);
        const int row = m0 + lrow;
        if (row < nr) out[(long)row * ldo + col] = 2.f * (rs[lrow] * z_pre[rr] - acc[0][0][r]) * m_pre[rr];
    }
}

__global__ void row_sqnorm_kernel(const float* __restrict__ Z, int ldz, float* __restrict__ sq, int rows, int p) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* z = Z + (long)row * ldz;
    float s = 0.f;
    for (int j = threadIdx.x & 63; j < p; j += 64) s = fmaf(z[j], z[j], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sq[row] = s;
}

// per-feature mean of the resident data set (float64 accumulation, fixed order): the centre that the step engine subtracts
// from every row of the MMD operand.  Once per fit; one 64-column strip per workgroup, rows dealt round-robin to 16 waves.
__global__ __launch_bounds__(1024) void col_mean_kernel(const float* __restrict__ data, int ldd, int rows, int d, float* __restrict__ out) {
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (j < d)
        for (int r = wave; r < rows; r += 16) s += (double)data[(long)r * ldd + j];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && j < d) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        out[j] = (float)(t / (double)rows);
    }
}

// ---- error text (thread-local) ---------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace vgan

using namespace vgan;

// Workgroups are dealt round-robin over the 8 XCDs (block b runs on the XCD of b % 8), each with a private
// 4 MiB L2.  Tiles that share a row or column panel of Z should therefore sit on ONE XCD: the table is
// sorted along a Morton (Z-order) curve of the tile grid, cut into 8 contiguous chunks (compact, roughly
// square footprints) and chunk x is placed at the table positions congruent to x mod 8.  Speed only:
// any order gives the same sums (the reduction is by tile index, deterministic per table).
static uint64_t morton_key(uint32_t r, uint32_t c) {
    uint64_t k = 0;
    for (int b = 0; b < 16; ++b) k |= ((uint64_t)((r >> b) & 1) << (2 * b + 1)) | ((uint64_t)((c >> b) & 1) << (2 * b));
    return k;
}
static void order_tiles_for_xcds(int32_t* tiles, int count, int tile) {
    constexpr int NX = 8;
    const int T = tile == 256 ? 256 : tile, TCOL = tile == 256 ? 128 : tile;  // tile 256 = 256 rows x 128 columns
    std::vector<std::array<int32_t, VGAN_TILE_INTS>> v(count);
    for (int t = 0; t < count; ++t) std::memcpy(v[t].data(), tiles + (size_t)t * VGAN_TILE_INTS, sizeof(int32_t) * VGAN_TILE_INTS);
    std::stable_sort(v.begin(), v.end(), [T, TCOL](const auto& a, const auto& b) {
        return morton_key(a[0] / T, a[1] / TCOL) < morton_key(b[0] / T, b[1] / TCOL);
    });
    const int q = count / NX, r = count % NX;
    int src = 0;
    for (int x = 0; x < NX; ++x) {
        const int len = q + (x < r ? 1 : 0);
        for (int k = 0; k < len; ++k, ++src) std::memcpy(tiles + (size_t)(k * NX + x) * VGAN_TILE_INTS, v[src].data(), sizeof(int32_t) * VGAN_TILE_INTS);
    }
}

// XCD-aware order (see order_tiles_for_xcds) for a table the caller has filtered or concatenated itself
extern "C" int vgan_mmd_order_tiles(int32_t* tiles, int count, int tile) {
    if (tiles == nullptr || count < 0 || (tile != 64 && tile != 128 && tile != 256)) {
        set_error("vgan_mmd_order_tiles: bad argument");
        return VGAN_ERR_ARG;
    }
    if (count > 1) order_tiles_for_xcds(tiles, count, tile);
    return VGAN_OK;
}

extern "C" int vgan_abi_version(void) { return VGAN_ABI_VERSION; }
extern "C" const char* vgan_last_error(void) { return g_err; }

// Host-side tile table.  Single rank: symmetric blocks use the upper triangle only (TWICE + MIRROR);
// the XY block is laid out with rows in the Y half and columns in the X half so that Wg ([n,2n],
// wrow0 = n) needs no transposed store for it.  Row-sharded ranks cover (own rows) x (all columns).
extern "C" int vgan_mmd_build_tiles(int n, int grad_mode, int rank, int world, int tile, int32_t* out, int cap) {
    if (n <= 0 || grad_mode < 0 || grad_mode > 2 || world < 1 || rank < 0 || rank >= world || (tile != 64 && tile != 128 && tile != 256)) {
        set_error("vgan_mmd_build_tiles: bad argument");
        return -1;
    }
    // tile 64 / 128: square tiles; tile 256: 256 rows x 128 columns (GemmBF3Wide).  A symmetric block keeps the upper triangle:
    // tiles right of the T x T square on the diagonal are counted twice and mirrored, the square's own tiles (one for a square
    // tile on the diagonal, two side by side for 256 x 128) are computed in full -- a diagonal 64- or 128-wide tile is its own
    // mirror image, which is the same statement.
    const int T = tile == 256 ? 256 : tile, TC = tile == 256 ? 128 : tile;
    int count = 0;
    auto emit = [&](int r0, int c0, int rlim, int clim, int flags) {
        if (out != nullptr && count < cap) {
            int32_t* e = out + (size_t)count * VGAN_TILE_INTS;
            e[0] = r0; e[1] = c0; e[2] = rlim; e[3] = clim; e[4] = flags; e[5] = e[6] = e[7] = 0;
        }
        ++count;
    };
    if (world == 1) {
        // XX: upper triangle; gradient rows only when grad_mode == 2
        for (int r = 0; r < n; r += T)
            for (int c = r; c < n; c += TC) {
                const bool off = c >= r + T;
                int fl = 0 | (off ? VGAN_TF_TWICE : 0);
                if (grad_mode == 2) fl |= VGAN_TF_STORE | (off ? VGAN_TF_MIRROR : 0);
                emit(r, c, n, n, fl);
            }
        // XY: rows in Y, columns in X (full block)
        for (int r = 0; r < n; r += T)
            for (int c = 0; c < n; c += TC) {
                int fl = 1 | VGAN_TF_NEG;
                if (grad_mode >= 1) fl |= VGAN_TF_STORE;
                if (grad_mode == 2) fl |= VGAN_TF_MIRROR;
                emit(n + r, c, 2 * n, n, fl);
            }
        // YY: upper triangle
        for (int r = 0; r < n; r += T)
            for (int c = r; c < n; c += TC) {
                const bool off = c >= r + T;
                int fl = 2 | (off ? VGAN_TF_TWICE : 0);
                if (grad_mode >= 1) fl |= VGAN_TF_STORE | (off ? VGAN_TF_MIRROR : 0);
                emit(n + r, n + c, 2 * n, 2 * n, fl);
            }
    } else {
        // rank owns rows [lo, hi) of each half; Wg (if any) holds its own rows only:
        //   grad_mode 1: Wg [hi-lo, 2n], wrow0 = n + lo ; grad_mode 2 is not sharded (caller error).
        if (grad_mode == 2) {
            set_error("vgan_mmd_build_tiles: grad_mode 2 is not available row-sharded");
            return -1;
        }
        const int lo = (int)((long)n * rank / world), hi = (int)((long)n * (rank + 1) / world);
        // XX feeds only the block sum of the reported loss (no gradient weights, so no row ownership): the upper triangle of
        // the WHOLE block is dealt round-robin over the ranks -- n^2 / (2 world) pairs each instead of the n^2 / world of
        // "own rows x all columns".
        int k = 0;
        for (int r = 0; r < n; r += T)
            for (int c = r; c < n; c += TC, ++k)
                if (k % world == rank) emit(r, c, n, n, 0 | (c >= r + T ? VGAN_TF_TWICE : 0));
        for (int r = lo; r < hi; r += T)
            for (int c = 0; c < n; c += TC) emit(n + r, c, n + hi, n, 1 | VGAN_TF_NEG | (grad_mode ? VGAN_TF_STORE : 0));
        // YY: own rows x all columns; inside the rank's own diagonal block the upper triangle with mirrored stores does (both
        // halves of a mirrored pair are rows of this rank), when the block is a whole number of tiles on the tile grid
        const bool tri = lo % T == 0 && hi % T == 0;
        for (int r = lo; r < hi; r += T)
            for (int c = 0; c < n; c += TC) {
                const bool diag = tri && c >= lo && c < hi;
                if (diag && c < r) continue;
                int fl = 2 | (grad_mode ? VGAN_TF_STORE : 0);
                if (diag && c >= r + T) fl |= VGAN_TF_TWICE | (grad_mode ? VGAN_TF_MIRROR : 0);
                emit(n + r, n + c, n + hi, 2 * n, fl);
            }
    }
    if (out != nullptr && count > cap) return -1;
    if (out != nullptr) order_tiles_for_xcds(out, count, tile);
    return count;
}

// Workgroup shape per kernel.  Measured on MI355X at n = 1024, d = 784: the Gram (two tiles per CU) is faster with
// 256-thread workgroups (48.7 vs 57.4 us), the backward GEMM (208 tiles for 256 CUs: one per CU) with 512-thread K-split
// workgroups and no row slabs (48.9 vs 57.4 us; 50.4 with two slabs of 256-thread workgroups).
constexpr int kGramKW = 1, kBwdKW = 2;

template <int KW>
static void launch_gram_kw(dim3 grid, hipStream_t s, bool vec, int calibrate, const float* Z, int ldz, const float* sq, int n, int p,
                           const float* bw, const TileDesc* td, int ntiles, float* Wg, int ldw, int wrow0, float* partial,
                           const ColmaxJob& cj) {
    dim3 block(kBlock * KW);
    if (calibrate) {
        if (vec)
            hipLaunchKernelGGL((mmd_gram_kernel<4, true, KW>), grid, block, 0, s, Z, ldz, sq, n, p, bw, td, ntiles, Wg, ldw, wrow0, partial, cj, RbfMults{});
        else
            hipLaunchKernelGGL((mmd_gram_kernel<1, true, KW>), grid, block, 0, s, Z, ldz, sq, n, p, bw, td, ntiles, Wg, ldw, wrow0, partial, cj, RbfMults{});
    } else {
        if (vec)
            hipLaunchKernelGGL((mmd_gram_kernel<4, false, KW>), grid, block, 0, s, Z, ldz, sq, n, p, bw, td, ntiles, Wg, ldw, wrow0, partial, cj, RbfMults{});
        else
            hipLaunchKernelGGL((mmd_gram_kernel<1, false, KW>), grid, block, 0, s, Z, ldz, sq, n, p, bw, td, ntiles, Wg, ldw, wrow0, partial, cj, RbfMults{});
    }
}

static int launch_gram(const float* Z, int ldz, const float* sq, int n, int p, const float* bw, const int32_t* tiles, int ntiles,
                       int calibrate, float* Wg, int ldw, int wrow0, float* partial, const ColmaxJob& cj, int extra_blocks,
                       vgan_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const TileDesc* td = reinterpret_cast<const TileDesc*>(tiles);
    const bool vec = (p % 4 == 0) && (ldz % 4 == 0) && aligned16(Z);
    dim3 grid(ntiles + extra_blocks);
    launch_gram_kw<kGramKW>(grid, s, vec, calibrate, Z, ldz, sq, n, p, bw, td, ntiles, Wg, ldw, wrow0, partial, cj);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_gram_general(const float* Z, int ldz, const float* sq, int n, int p, const float* bw, const int32_t* tiles,
                                     int ntiles, const float* multipliers, int n_kernels, float* Wg, int ldw, int wrow0,
                                     float* partial, vgan_stream_t stream) {
    VGAN_CHECK_ARG(Z && sq && tiles && partial && bw && n > 0 && p > 0 && ntiles > 0 && ldz >= p);
    VGAN_CHECK_ARG(multipliers && n_kernels >= 1 && n_kernels <= VGAN_RBF_MAX_KERNELS && (reinterpret_cast<uintptr_t>(partial) & 15) == 0);
    RbfMults rm{};
    rm.nk = n_kernels;
    for (int k = 0; k < n_kernels; ++k) {
        VGAN_CHECK_ARG(multipliers[k] > 0.f);
        rm.mult[k] = multipliers[k];
    }
    const TileDesc* td = reinterpret_cast<const TileDesc*>(tiles);
    const bool vec = (p % 4 == 0) && (ldz % 4 == 0) && aligned16(Z);
    if (vec)
        hipLaunchKernelGGL((mmd_gram_kernel<4, false, 1, true>), dim3(ntiles), dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, sq, n, p, bw, td,
                           ntiles, Wg, ldw, wrow0, partial, ColmaxJob{}, rm);
    else
        hipLaunchKernelGGL((mmd_gram_kernel<1, false, 1, true>), dim3(ntiles), dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, sq, n, p, bw, td,
                           ntiles, Wg, ldw, wrow0, partial, ColmaxJob{}, rm);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_gram(const float* Z, int ldz, const float* sq, int n, int p, const float* bw, const int32_t* tiles,
                             int ntiles, int calibrate, float* Wg, int ldw, int wrow0, float* partial, vgan_stream_t stream) {
    VGAN_CHECK_ARG(Z && sq && tiles && partial && n > 0 && p > 0 && ntiles > 0 && ldz >= p);
    VGAN_CHECK_ARG(calibrate || bw);
    VGAN_CHECK_ARG((reinterpret_cast<uintptr_t>(partial) & 15) == 0);
    return launch_gram(Z, ldz, sq, n, p, bw, tiles, ntiles, calibrate, Wg, ldw, wrow0, partial, ColmaxJob{}, 0, stream);
}

extern "C" int vgan_mmd_gram_colmax(const float* Z, int ldz, const float* sq, int n, int p, const float* bw, const int32_t* tiles,
                                    int ntiles, float* Wg, int ldw, int wrow0, float* partial, const float* S, int lds,
                                    int from_softmax, int row_offset, uint64_t* colpart, int nrows, int d, vgan_stream_t stream) {
    VGAN_CHECK_ARG(Z && sq && tiles && partial && bw && n > 0 && p > 0 && ntiles > 0 && ldz >= p);
    VGAN_CHECK_ARG(S && colpart && nrows > 0 && d > 0 && lds >= d);
    VGAN_CHECK_ARG((reinterpret_cast<uintptr_t>(partial) & 15) == 0);
    ColmaxJob cj{S, reinterpret_cast<unsigned long long*>(colpart), lds, row_offset, nrows, d, from_softmax, (d + 63) / 64};
    const int extra = cj.nbx * ((nrows + kColChunkRows - 1) / kColChunkRows);
    return launch_gram(Z, ldz, sq, n, p, bw, tiles, ntiles, 0, Wg, ldw, wrow0, partial, cj, extra, stream);
}

extern "C" int vgan_mmd_reduce(const float* partial, const int32_t* tiles, int ntiles, double* stats, int zero_first,
                               vgan_stream_t stream) {
    VGAN_CHECK_ARG(partial && tiles && stats && ntiles > 0);
    hipLaunchKernelGGL(mmd_reduce_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, partial,
                       reinterpret_cast<const TileDesc*>(tiles), ntiles, stats, zero_first);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_set_bandwidth(const double* stats, int n, float* bw, vgan_stream_t stream) {
    VGAN_CHECK_ARG(stats && bw && n > 0);
    hipLaunchKernelGGL(mmd_set_bandwidth_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, stats, n, bw);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_loss(const double* stats, const uint64_t* colkey, int n, int d, float weight, float* loss,
                             float* loss_accum, float accum_scale, uint64_t* step_counter, vgan_stream_t stream) {
    VGAN_CHECK_ARG(stats && loss && n > 0 && d > 0);
    hipLaunchKernelGGL(mmd_loss_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, stats,
                       reinterpret_cast<const unsigned long long*>(colkey), n, d, weight, loss, loss_accum, accum_scale,
                       reinterpret_cast<unsigned long long*>(step_counter));
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_finalize(const float* partial, const int32_t* tiles, int ntiles, const uint64_t* colpart, int chunks,
                                 uint64_t* colkey, int n, int d, float weight, double* stats, float* loss, float* loss_accum,
                                 float accum_scale, uint64_t* step_counter, vgan_stream_t stream) {
    const vgan_finalize_job job{partial, tiles, colpart, colkey, stats, loss, loss_accum, step_counter, ntiles, chunks, n, d, weight,
                                accum_scale, 0, 0};
    VGAN_CHECK_ARG(finalize_job_ok(job));
    hipLaunchKernelGGL(mmd_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, job);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_backward(const float* Wg, int ldw, const float* Z, int ldz, int wrow0, int nr, int ncols, int p,
                                 const float* mul, int ldmul, const float* mul_shift, float* out, int ldo, int splits,
                                 int64_t slab_stride, const vgan_finalize_job* finalize, vgan_stream_t stream) {
    VGAN_CHECK_ARG(Wg && Z && out && nr > 0 && ncols > 0 && p > 0 && ldw >= ncols && ldz >= p && ldo >= p && wrow0 >= 0 &&
                   wrow0 + nr <= ncols);
    VGAN_CHECK_ARG(mul == nullptr || ldmul >= p);
    VGAN_CHECK_ARG(splits >= 1 && splits <= 64 && (splits == 1 || slab_stride >= (int64_t)nr * ldo));
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (ncols % 4 == 0) && (ldw % 4 == 0) && (p % 4 == 0) && (ldz % 4 == 0) && aligned16(Wg) && aligned16(Z);
    VGAN_CHECK_ARG(mul_shift == nullptr || mul != nullptr);
    constexpr int kw = kBwdKW;
    const int kt = GBK * kw;
    const int kchunk = ((ncols + splits - 1) / splits + kt - 1) / kt * kt;  // whole K tiles per slice (keeps 16-byte alignment)
    vgan_finalize_job job{};
    if (finalize != nullptr) {
        VGAN_CHECK_ARG(finalize_job_ok(*finalize));
        job = *finalize;
    }
    dim3 grid(((p + GT - 1) / GT) * ((nr + GT - 1) / GT) + (finalize != nullptr ? 1 : 0), splits), block(kBlock * kw);
#define VGAN_BWD(V) hipLaunchKernelGGL((mmd_backward_kernel<V, kBwdKW>), grid, block, 0, s, Wg, ldw, Z, ldz, wrow0, nr, ncols, p, mul, ldmul, mul_shift, out, ldo, kchunk, (long)slab_stride, job)
    if (vec) VGAN_BWD(4); else VGAN_BWD(1);
#undef VGAN_BWD
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_row_sqnorm(const float* Z, int ldz, float* sq, int rows, int p, vgan_stream_t stream) {
    VGAN_CHECK_ARG(Z && sq && rows > 0 && p > 0 && ldz >= p);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((rows + 3) / 4), dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, sq, rows, p);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_col_mean(const float* data, int ldd, int rows, int d, float* out, vgan_stream_t stream) {
    VGAN_CHECK_ARG(data && out && rows > 0 && d > 0 && ldd >= d);
    hipLaunchKernelGGL(col_mean_kernel, dim3((d + 63) / 64), dim3(1024), 0, (hipStream_t)stream, data, ldd, rows, d, out);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}
