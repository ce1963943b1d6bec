// nn.Linear forward/backward on the fp32 MFMA (Generator_big / Encoder / Decoder layers).
// Reference ops replaced: src/models/Generator.py:61-66, src/models/Detector.py:8-13,24-29
// (F.linear == addmm forward; autograd's two mm per layer backward).
#include "gemm_core.hpp"
#include "mmd_xx.hpp"

namespace vgan {

constexpr int LBM = 64, LBN = 64, LBK = 32;

// y = x . W^T + b           A = x (KC), B = W (KC)
template <int VEC, bool XSL>
__global__ __launch_bounds__(kBlock, 2) void linear_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                              int ldw, const float* __restrict__ b, float* __restrict__ y,
                                                              int ldy, int n, int in, int out, int x_nslabs, long x_slab_stride) {
    using G = GemmTile<LBM, LBN, LBK, KC, KC, VEC, XSL ? 1 : 0>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    const int m0 = blockIdx.y * LBM, n0 = blockIdx.x * LBN;
    const int col = n0 + G::sub_col(0);
    const float bias = (b != nullptr) ? b[min(col, out - 1)] : 0.f;  // requested before the main loop, not after it
    f32x16 acc[G::WM][G::WN];
    zero_acc(acc);
    G::template run<false>(x, ldx, W, ldw, m0, n0, n, out, in, lds, nullptr, acc, x_nslabs, x_slab_stride);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + G::sub_row(0, r);
        if (row < n && col < out) y[(long)row * ldy + col] = acc[0][0][r] + bias;
    }
}

// dx = dy . W               A = dy (KC, K = out), B(j=in, k=out) = W[k*ldw + j] (MC)
template <int VEC>
__global__ __launch_bounds__(kBlock, 2) void linear_bwd_input_kernel(const float* __restrict__ dy, int lddy,
                                                                    const float* __restrict__ W, int ldw, float* __restrict__ dx,
                                                                    int lddx, int n, int in, int out) {
    using G = GemmTile<LBM, LBN, LBK, KC, MC, VEC>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    const int m0 = blockIdx.y * LBM, n0 = blockIdx.x * LBN;
    f32x16 acc[G::WM][G::WN];
    zero_acc(acc);
    G::template run<false>(dy, lddy, W, ldw, m0, n0, n, in, out, lds, nullptr, acc);
    const int col = n0 + G::sub_col(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + G::sub_row(0, r);
        if (row < n && col < in) dx[(long)row * lddx + col] = acc[0][0][r];
    }
}

// dW = dy^T . x ; db = colsum(dy)    A(i=out, k=row) = dy[k*lddy + i] (MC), B(j=in, k=row) = x[k*ldx + j] (MC)
// Split-K over the batch rows (blockIdx.z): the contraction length is the batch (1024) while the output is
// small, so one launch needs the row range cut into `kchunk`-row slices to fill 256 CUs; slice s writes its
// partial result into slab s (dW + s*slab_stride, db + s*slab_stride), summed afterwards by vgan_reduce_slabs
// (fixed order: bitwise reproducible, no atomics).
template <int VEC, bool XSL>
__global__ __launch_bounds__(kBlock, 2) void linear_bwd_params_kernel(const float* __restrict__ dy, int lddy,
                                                                     const float* __restrict__ x, int ldx, float* __restrict__ dW,
                                                                     int lddw, float* __restrict__ db, int n, int in, int out,
                                                                     int kchunk, long slab_stride, int x_nslabs, long x_slab_stride) {
    using G = GemmTile<LBM, LBN, LBK, MC, MC, VEC, XSL ? 2 : 0>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    __shared__ float side[LBM];
    const int m0 = blockIdx.y * LBM, n0 = blockIdx.x * LBN;
    const int k0 = blockIdx.z * kchunk;
    const int klen = min(kchunk, n - k0);
    dy += (long)k0 * lddy;
    x += (long)k0 * ldx;
    dW += blockIdx.z * slab_stride;
    if (db != nullptr) db += blockIdx.z * slab_stride;
    n = klen;
    f32x16 acc[G::WM][G::WN];
    zero_acc(acc);
    const bool do_bias = (db != nullptr) && (blockIdx.x == 0);
    if (klen <= 0) {  // empty slice (more slabs than row slices): its slab is all zeros
        if (threadIdx.x < LBM) side[threadIdx.x] = 0.f;
        __syncthreads();
    } else if (do_bias)
        G::template run<true>(dy, lddy, x, ldx, m0, n0, out, in, n, lds, side, acc, x_nslabs, x_slab_stride);
    else
        G::template run<false>(dy, lddy, x, ldx, m0, n0, out, in, n, lds, side, acc, x_nslabs, x_slab_stride);
    const int col = n0 + G::sub_col(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + G::sub_row(0, r);
        if (row < out && col < in) dW[(long)row * lddw + col] = acc[0][0][r];
    }
    if (do_bias && threadIdx.x < LBM && m0 + threadIdx.x < out) db[m0 + threadIdx.x] = side[threadIdx.x];
}

// ---- tall-skinny variants (GemmTileKS): used when the 64x64 grid would be a handful of long-K tiles -----------------
constexpr int KSBK = 128;
// y = x . W^T + b for a narrow output (Encoder's collapsed chain: 2n x (d+1) -> L): 32x32 tiles, the K range of a tile split
// over the NW waves of its workgroup -- 4x the workgroups of the 64x64 kernel, each with a 1/NW-long dependent MFMA chain
template <int VEC, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void linear_fwd_ks_kernel(const float* __restrict__ x, int ldx,
                                                                               const float* __restrict__ W, int ldw,
                                                                               const float* __restrict__ b, float* __restrict__ y,
                                                                               int ldy, int n, int in, int out) {
    using G = GemmTileKS<KSBK, KC, KC, VEC, NW>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int col = n0 + G::col_of();
    const float bias = (b != nullptr) ? b[min(col, out - 1)] : 0.f;
    float o[G::NR];
    G::run(x, ldx, W, ldw, m0, n0, n, out, in, lds, o);
#pragma unroll
    for (int rr = 0; rr < G::NR; ++rr) {
        const int row = m0 + G::row_of(rr);
        if (row < n && col < out) y[(long)row * ldy + col] = o[rr] + bias;
    }
}
// `in_staged` >= in: the column range of W the staging may touch (a multiple of 4 on the vector path; the caller guarantees
// ldw >= in_staged); columns >= in are computed and not stored
template <int VEC>
__global__ __launch_bounds__(kBlock, 2) void linear_bwd_input_ks_kernel(const float* __restrict__ dy, int lddy,
                                                                       const float* __restrict__ W, int ldw, float* __restrict__ dx,
                                                                       int lddx, int n, int in, int out, int in_staged) {
    using G = GemmTileKS<KSBK, KC, MC, VEC>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    float o[G::NR];
    G::run(dy, lddy, W, ldw, m0, n0, n, in_staged, out, lds, o);
    const int col = n0 + G::col_of();
#pragma unroll
    for (int rr = 0; rr < G::NR; ++rr) {
        const int row = m0 + G::row_of(rr);
        if (row < n && col < in) dx[(long)row * lddx + col] = o[rr];
    }
}
template <int VEC, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void linear_bwd_params_ks_kernel(const float* __restrict__ dy, int lddy,
                                                                                      const float* __restrict__ x, int ldx,
                                                                                      float* __restrict__ dW, int lddw, int n, int in,
                                                                                      int out) {
    using G = GemmTileKS<KSBK, MC, MC, VEC, NW>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    float o[G::NR];
    G::run(dy, lddy, x, ldx, m0, n0, out, in, n, lds, o);
    const int col = n0 + G::col_of();
#pragma unroll
    for (int rr = 0; rr < G::NR; ++rr) {
        const int row = m0 + G::row_of(rr);
        if (row < out && col < in) dW[(long)row * lddw + col] = o[rr];
    }
}
// The same tall-skinny product with the X-X tiles of the Gram riding behind it (mmd_xx.hpp): the M_4 launch of the training
// step is long (8.5 us), occupies ~50 of the 256 CUs and is not L2-bound, and by then the X half of the split operand is
// L2 / Infinity-Cache warm -- the carrier the cold-operand experiments of mmd_xx.hpp were missing.  1-D grid: product tiles
// first, then one X-X tile per workgroup (its 12 surplus waves return at once).
__global__ __launch_bounds__(1024, 1) void linear_bwd_params_ks_xx_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                                         int ldx, float* __restrict__ dW, int lddw, int n, int in, int out,
                                                                         int gx, int ptiles, XXJob xx) {
    using G = GemmTileKS<KSBK, MC, MC, 4, 16>;
    constexpr int kBytes = G::kLdsFloats * 4 > GemmBF3<64>::kLdsBytes ? G::kLdsFloats * 4 : GemmBF3<64>::kLdsBytes;
    __shared__ __attribute__((aligned(16))) char raw[kBytes];
    __shared__ float xx_red[4];
    if ((int)blockIdx.x >= ptiles) {  // block-uniform
        xx_tile_body(xx, blockIdx.x - ptiles, raw, xx_red);
        return;
    }
    float* lds = reinterpret_cast<float*>(raw);
    const int m0 = (blockIdx.x / gx) * 32, n0 = (blockIdx.x % gx) * 32;
    float o[G::NR];
    G::run(dy, lddy, x, ldx, m0, n0, out, in, n, lds, o);
    const int col = n0 + G::col_of();
#pragma unroll
    for (int rr = 0; rr < G::NR; ++rr) {
        const int row = m0 + G::row_of(rr);
        if (row < out && col < in) dW[(long)row * lddw + col] = o[rr];
    }
}

// heuristic: few 64x64 tiles and a long contraction -> the K loop of a tile is the critical path
static inline bool use_ks(int rows, int cols, int k) {
    const long tiles64 = (long)((rows + 63) / 64) * ((cols + 63) / 64);
    return tiles64 <= 32 && k >= 128;
}

// the weight-gradient product dW = dy^T x has the batch as its contraction: from c4 sizes up a narrow dW (the collapsed
// generator's M_4: 2048 x 132 over 4096 rows, 4096 x 260 over 8192) is 100-320 64x64 tiles with a very long K loop each -- the
// 16-wave tall-skinny tile (4x the workgroups, K split over 16 waves) again, although the 64x64 grid is no longer "a handful":
// measured 130 -> 49 us at c4, 332 -> 244 us at c5
static inline bool use_ks_params(int rows, int cols, int k) {
    const long tiles64 = (long)((rows + 63) / 64) * ((cols + 63) / 64);
    return use_ks(rows, cols, k) || (tiles64 <= 512 && k >= 2048);
}

static inline dim3 grid_for(int rows, int cols) { return dim3((cols + LBN - 1) / LBN, (rows + LBM - 1) / LBM, 1); }

}  // namespace vgan

using namespace vgan;

extern "C" int vgan_linear_forward(const float* x, int ldx, int x_nslabs, int64_t x_slab_stride, const float* W, int ldw,
                                   const float* b, float* y, int ldy, int n, int in, int out, vgan_stream_t stream) {
    VGAN_CHECK_ARG(x && W && y && n > 0 && in > 0 && out > 0 && ldx >= in && ldw >= in && ldy >= out && x_nslabs >= 1);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (in % 4 == 0) && (ldx % 4 == 0) && (ldw % 4 == 0) && aligned16(x) && aligned16(W) && (x_slab_stride % 4 == 0);
    const long xs = (long)x_slab_stride;
    if (x_nslabs == 1 && use_ks(n, out, in)) {  // narrow output, long contraction
        dim3 g((out + 31) / 32, (n + 31) / 32);
        if (vec && in >= 512)
            hipLaunchKernelGGL((linear_fwd_ks_kernel<4, 16>), g, dim3(1024), 0, s, x, ldx, W, ldw, b, y, ldy, n, in, out);
        else if (vec)
            hipLaunchKernelGGL((linear_fwd_ks_kernel<4, 4>), g, dim3(kBlock), 0, s, x, ldx, W, ldw, b, y, ldy, n, in, out);
        else
            hipLaunchKernelGGL((linear_fwd_ks_kernel<1, 4>), g, dim3(kBlock), 0, s, x, ldx, W, ldw, b, y, ldy, n, in, out);
        VGAN_CHECK_LAUNCH();
        return VGAN_OK;
    }
#define VGAN_FWD(V, S) hipLaunchKernelGGL((linear_fwd_kernel<V, S>), grid_for(n, out), dim3(kBlock), 0, s, x, ldx, W, ldw, b, y, ldy, n, in, out, x_nslabs, xs)
    if (x_nslabs > 1) { if (vec) VGAN_FWD(4, true); else VGAN_FWD(1, true); }
    else { if (vec) VGAN_FWD(4, false); else VGAN_FWD(1, false); }
#undef VGAN_FWD
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_linear_backward_input(const float* dy, int lddy, const float* W, int ldw, float* dx, int lddx, int n, int in,
                                          int out, vgan_stream_t stream) {
    VGAN_CHECK_ARG(dy && W && dx && n > 0 && in > 0 && out > 0 && lddy >= out && ldw >= in && lddx >= in);
    hipStream_t s = (hipStream_t)stream;
    const bool vec_but_in = (out % 4 == 0) && (lddy % 4 == 0) && (ldw % 4 == 0) && aligned16(dy) && aligned16(W);
    const bool vec = vec_but_in && (in % 4 == 0);
    if (use_ks(n, in, out)) {
        dim3 g((in + 31) / 32, (n + 31) / 32);
        const int in4 = (in + 3) / 4 * 4;  // W's rows are ldw long: a ragged last group of 4 columns is still inside the row
        if (vec_but_in && ldw >= in4)
            hipLaunchKernelGGL(linear_bwd_input_ks_kernel<4>, g, dim3(kBlock), 0, s, dy, lddy, W, ldw, dx, lddx, n, in, out, in4);
        else
            hipLaunchKernelGGL(linear_bwd_input_ks_kernel<1>, g, dim3(kBlock), 0, s, dy, lddy, W, ldw, dx, lddx, n, in, out, in);
    } else if (vec)
        hipLaunchKernelGGL(linear_bwd_input_kernel<4>, grid_for(n, in), dim3(kBlock), 0, s, dy, lddy, W, ldw, dx, lddx, n, in, out);
    else
        hipLaunchKernelGGL(linear_bwd_input_kernel<1>, grid_for(n, in), dim3(kBlock), 0, s, dy, lddy, W, ldw, dx, lddx, n, in, out);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_linear_backward_params(const float* dy, int lddy, const float* x, int ldx, int x_nslabs, int64_t x_slab_stride,
                                           float* dW, int lddw, float* db, int n, int in, int out, int splits, int64_t slab_stride,
                                           vgan_stream_t stream) {
    VGAN_CHECK_ARG(dy && x && dW && n > 0 && in > 0 && out > 0 && lddy >= out && ldx >= in && lddw >= in && x_nslabs >= 1);
    VGAN_CHECK_ARG(splits >= 1 && splits <= 64 && (splits == 1 || slab_stride > 0));
    hipStream_t s = (hipStream_t)stream;
    // row slices are multiples of 4 rows so that every slice keeps the 16-byte alignment of the vector path
    int kchunk = (n + splits - 1) / splits;
    kchunk = (kchunk + 3) / 4 * 4;
    const int nz = (n + kchunk - 1) / kchunk;
    const bool vec = (out % 4 == 0) && (lddy % 4 == 0) && (in % 4 == 0) && (ldx % 4 == 0) && aligned16(dy) && aligned16(x) &&
                     (x_slab_stride % 4 == 0);
    dim3 grid = grid_for(out, in);
    grid.z = splits;  // slices beyond nz see klen <= 0 and write zeros, so the reducer may always sum `splits` slabs
    (void)nz;
    const long xs = (long)x_slab_stride;
    if (splits == 1 && x_nslabs == 1 && db == nullptr && use_ks_params(out, in, n)) {  // tall-skinny: no slabs needed at all
        dim3 g((in + 31) / 32, (out + 31) / 32);
        if (vec && n >= 256)  // long contraction: 16 waves per workgroup (see GemmTileKS)
            hipLaunchKernelGGL((linear_bwd_params_ks_kernel<4, 16>), g, dim3(1024), 0, s, dy, lddy, x, ldx, dW, lddw, n, in, out);
        else if (vec)
            hipLaunchKernelGGL((linear_bwd_params_ks_kernel<4, 4>), g, dim3(kBlock), 0, s, dy, lddy, x, ldx, dW, lddw, n, in, out);
        else
            hipLaunchKernelGGL((linear_bwd_params_ks_kernel<1, 4>), g, dim3(kBlock), 0, s, dy, lddy, x, ldx, dW, lddw, n, in, out);
        VGAN_CHECK_LAUNCH();
        return VGAN_OK;
    }
#define VGAN_BWP(V, S) hipLaunchKernelGGL((linear_bwd_params_kernel<V, S>), grid, dim3(kBlock), 0, s, dy, lddy, x, ldx, dW, lddw, db, n, in, out, kchunk, (long)slab_stride, x_nslabs, xs)
    if (x_nslabs > 1) { if (vec) VGAN_BWP(4, true); else VGAN_BWP(1, true); }
    else { if (vec) VGAN_BWP(4, false); else VGAN_BWP(1, false); }
#undef VGAN_BWP
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

// host-side query: does (n, in, out) meet the shape contract of vgan_linear_backward_params_xx (alignment aside)?
extern "C" int vgan_linear_backward_params_xx_supported(int n, int in, int out) {
    return (n >= 256 && in > 0 && out > 0 && in % 4 == 0 && out % 4 == 0 && use_ks(out, in, n)) ? 1 : 0;
}

extern "C" int vgan_linear_backward_params_xx(const float* dy, int lddy, const float* x, int ldx, float* dW, int lddw, int n, int in,
                                              int out, const vgan_xx_job* xxjob, vgan_stream_t stream) {
    VGAN_CHECK_ARG(dy && x && dW && xxjob && n > 0 && in > 0 && out > 0 && lddy >= out && ldx >= in && lddw >= in);
    const vgan_xx_job& j = *xxjob;
    VGAN_CHECK_ARG(j.Dh && j.Dl && j.dsq && j.tiles && j.bw && j.partial && j.ntiles > 0 && j.ldd % 64 == 0 && aligned16(j.Dh) && aligned16(j.Dl) &&
                   (reinterpret_cast<uintptr_t>(j.partial) & 15) == 0);
    // the shape contract of the 16-wave tall-skinny kernel (what vgan_linear_backward_params picks for the step's M_4 product)
    VGAN_CHECK_ARG((out % 4 == 0) && (lddy % 4 == 0) && (in % 4 == 0) && (ldx % 4 == 0) && aligned16(dy) && aligned16(x) && n >= 256 &&
                   use_ks(out, in, n));
    const int gx = (in + 31) / 32, gy = (out + 31) / 32;
    const XXJob xx{j.Dh, j.Dl, j.dsq, nullptr, nullptr, reinterpret_cast<const TileDesc*>(j.tiles), j.bw, j.partial, j.ldd, 1, 0, j.ntiles, 0};
    hipLaunchKernelGGL(linear_bwd_params_ks_xx_kernel, dim3(gx * gy + j.ntiles), dim3(1024), 0, (hipStream_t)stream, dy, lddy, x, ldx, dW, lddw, n,
                       in, out, gx, gx * gy, xx);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}
