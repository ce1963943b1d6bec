// Row-wise kernels of the V-GAN step: upper_softmax, the U * X projection, their backward, and the
// column arg-max of U used by the feature-count penalty.
// Reference ops replaced: src/models/Generator.py:18-22 (softmax, less/greater_equal, mul, add),
// src/vgan.py:616 (`fake_subspaces * batch`), src/models/Mmd_loss_constrained.py:50 (topk(U,1,0)),
// and the DataLoader batch gather (src/vgan.py:578-584, :599).
//
// All are HBM/L2-streaming kernels: one 64-lane wave per row, coalesced 256-byte wave accesses,
// reductions by wave shuffles only (no LDS).
#include "mmd_xx.hpp"
#include "vgan_common.hpp"

namespace vgan {

constexpr int kRowsPerBlock = kBlock / kWave;  // 4 rows per workgroup, one per wave

// S = softmax(logits row), U = S < 1/d ? S : 1, Y = U * X;  Z = [X ; Y], sq = row norms of Z.
// Batch row table: the batch of step t is rows[(t % row_batches) * row_stride + row_offset + i], with t read from
// the device-side step counter (so one captured HIP graph walks a whole epoch's shuffled batches).
struct RowSel {
    const int* rows;
    const unsigned long long* cursor;
    int row_batches, row_stride, row_offset;
    __device__ __forceinline__ long operator()(int i) const {
        if (rows == nullptr) return (long)(row_offset + i);
        const long b = cursor ? (long)(cursor[0] % (unsigned long long)row_batches) : 0l;
        return (long)rows[b * row_stride + row_offset + i];
    }
};

// y = fl(u * x) - c with the product ROUNDED first, as the reference forms `U * batch` (src/vgan.py:616) before any distance
// is taken (the file is compiled with -ffp-contract=fast, which would otherwise fuse the two into one fma)
__device__ __forceinline__ float project_centred(float u, float x, float c) {
#pragma clang fp contract(off)
    const float y = u * x;
    return y - c;
}

template <bool PROJECT>
__global__ __launch_bounds__(kBlock) void mask_forward_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ data,
                                                             int ldd, RowSel rows, float* __restrict__ S,
                                                             float* __restrict__ U, float* __restrict__ Zx, float* __restrict__ Zy,
                                                             int ldz, float* __restrict__ sqx, float* __restrict__ sqy, int n, int d,
                                                             const float* __restrict__ center, int norm_split) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* x = logits + (long)i * ldl;
    float m = -INFINITY;
    for (int j = lane; j < d; j += 64) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < d; j += 64) sum += expf(x[j] - m);
    sum = wave_sum(sum);
    const float tau = 1.0f / (float)d;  // float32(1/d), the threshold of Generator.py:20-21
    const float* xr = nullptr;
    float *zx = nullptr, *zy = nullptr;
    if constexpr (PROJECT) {
        const long src = rows(i);
        xr = data + src * ldd;
        zx = Zx ? Zx + (long)i * ldz : nullptr;
        zy = Zy + (long)i * ldz;
    }
    float nx = 0.f, ny = 0.f;
    for (int j = lane; j < d; j += 64) {
        const float s = expf(x[j] - m) / sum;
        const float u = s < tau ? s : 1.0f;
        S[(long)i * d + j] = s;
        if (U) U[(long)i * d + j] = u;
        if constexpr (PROJECT) {
            const float c = center ? center[j] : 0.f;
            const float xv = xr[j] - c;
            const float yv = project_centred(u, xr[j], c);
            if (zx) zx[j] = xv;
            zy[j] = yv;
            const float xn = norm_split ? split_value(xv) : xv, yn = norm_split ? split_value(yv) : yv;
            nx = fmaf(xn, xn, nx);
            ny = fmaf(yn, yn, ny);
        }
    }
    if constexpr (PROJECT) {
        nx = wave_sum(nx);
        ny = wave_sum(ny);
        if (lane == 0) {
            if (sqx) sqx[i] = nx;
            sqy[i] = ny;
        }
    }
}


// ---- logits inside the mask / projection launch (collapsed generator) ------------------------------------------------
// Generator_big without activations is ONE matrix (trainer.py): logits = [z|1] . At_4^T with At_4 [d, e0] (e0 = round4(L + 1)).
// With one wave per batch row and the row's logits living in registers anyway (NT float4 per lane), the product costs
// NT * 4 * e0 FMAs per lane (832 at d = 784).  MEASURED on MI355X at d = 784: not a win -- every workgroup stages all of At_4
// (163 KB) for its few rows, chunk after chunk behind barriers, and the launch grows by 12 us against the 5 us launch and
// 6 MB of traffic it replaces.  The step engine keeps it opt-in (VGAN_CHAIN_IN_MASK=1, v-gan_amd/trainer.py).
// At_4 is staged transposed through LDS in chunks of 16 k ([k][column], read by 16-byte LDS ops: the lanes' float4 columns);
// z_k is wave-uniform and comes through the scalar cache.
struct LogitsChain {
    const float* za;   // [n, ldza]: [z | 1 | 0-pad]
    const float* At4;  // [d, ldat], first e0 columns
    int ldza, ldat, e0;
};
constexpr int kChainK = 16;
__host__ __device__ constexpr int chain_pitch(int d) { return ((d + 3) / 4) * 4 + 4; }  // 4 * pitch = 16 (mod 32): 2-way stores at worst

template <int NT>
__device__ __forceinline__ void chain_logits(const LogitsChain& ch, float* __restrict__ lds_at, int d, int row, bool row_ok, float4 (&v)[NT]) {
    const int lane = threadIdx.x & 63, nq = d >> 2, pitch = chain_pitch(d);
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* zr = ch.za + (long)(row_ok ? row : 0) * ch.ldza;
    for (int kc = 0; kc < ch.e0; kc += kChainK) {
        __syncthreads();  // the previous chunk has been consumed
        for (int idx = threadIdx.x; idx < d * (kChainK / 4); idx += blockDim.x) {
            const int col = idx / (kChainK / 4), kq = idx % (kChainK / 4), k = kc + 4 * kq;
            const float4 a = k < ch.e0 ? *reinterpret_cast<const float4*>(ch.At4 + (long)col * ch.ldat + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            lds_at[(4 * kq + 0) * pitch + col] = a.x;
            lds_at[(4 * kq + 1) * pitch + col] = a.y;
            lds_at[(4 * kq + 2) * pitch + col] = a.z;
            lds_at[(4 * kq + 3) * pitch + col] = a.w;
        }
        __syncthreads();
        const int kw = min(kChainK, ch.e0 - kc);
        for (int k = 0; k < kw; ++k) {
            const float z = zr[kc + k];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int q = min(lane + 64 * t, nq - 1);
                const float4 a = *reinterpret_cast<const float4*>(lds_at + k * pitch + 4 * q);
                v[t].x = fmaf(z, a.x, v[t].x);
                v[t].y = fmaf(z, a.y, v[t].y);
                v[t].z = fmaf(z, a.z, v[t].z);
                v[t].w = fmaf(z, a.w, v[t].w);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (lane + 64 * t >= nq) v[t] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
}

// ---- fast paths: d % 4 == 0 and d <= 1024: the whole row lives in registers (NT float4 per lane), every global
// access is 16 bytes per lane, and each row is read exactly once.
template <int NT, bool CHAIN>
__global__ __launch_bounds__(kBlock) void mask_forward_vec_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ data,
                                                                 int ldd, RowSel rows, float* __restrict__ S, float* __restrict__ U,
                                                                 float* __restrict__ Zx, float* __restrict__ Zy, int ldz,
                                                                 float* __restrict__ sqx, float* __restrict__ sqy, int n, int d,
                                                                 const float* __restrict__ center, int norm_split, LogitsChain ch) {
    extern __shared__ __attribute__((aligned(16))) float chain_lds[];  // CHAIN: [kChainK][chain_pitch(d)]
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    const int nq = d >> 2;
    float4 v[NT], xv[NT], cv[NT];
    if constexpr (CHAIN) chain_logits<NT>(ch, chain_lds, d, i, i < n, v);  // all waves take part in the staging barriers
    if (i >= n) return;
    const float4* x4 = reinterpret_cast<const float4*>(logits + (long)i * ldl);
    const float4* xr4 = reinterpret_cast<const float4*>(data + rows(i) * ldd);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = lane + 64 * t;
        const bool ok = q < nq;
        if constexpr (!CHAIN) v[t] = ok ? x4[min(q, nq - 1)] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        xv[t] = xr4[min(q, nq - 1)];
        cv[t] = center ? reinterpret_cast<const float4*>(center)[min(q, nq - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) m = fmaxf(m, fmaxf(fmaxf(v[t].x, v[t].y), fmaxf(v[t].z, v[t].w)));
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        v[t].x = expf(v[t].x - m); v[t].y = expf(v[t].y - m); v[t].z = expf(v[t].z - m); v[t].w = expf(v[t].w - m);
        sum += (v[t].x + v[t].y) + (v[t].z + v[t].w);  // padded lanes hold exp(-inf) = 0
    }
    sum = wave_sum(sum);
    const float tau = 1.0f / (float)d;
    float nx = 0.f, ny = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = lane + 64 * t;
        if (q < nq) {
            float4 s4 = make_float4(v[t].x / sum, v[t].y / sum, v[t].z / sum, v[t].w / sum);
            float4 u4 = make_float4(s4.x < tau ? s4.x : 1.f, s4.y < tau ? s4.y : 1.f, s4.z < tau ? s4.z : 1.f, s4.w < tau ? s4.w : 1.f);
            // the MMD operand is CENTRED (a per-feature constant is subtracted from every row of [X ; U*X]: the distances
            // are translation invariant) so that a common offset of a feature does not eat the operand's mantissa
            const float4 x4c = make_float4(xv[t].x - cv[t].x, xv[t].y - cv[t].y, xv[t].z - cv[t].z, xv[t].w - cv[t].w);
            const float4 y4 = make_float4(project_centred(u4.x, xv[t].x, cv[t].x), project_centred(u4.y, xv[t].y, cv[t].y), project_centred(u4.z, xv[t].z, cv[t].z),
                                              project_centred(u4.w, xv[t].w, cv[t].w));
            reinterpret_cast<float4*>(S + (long)i * d)[q] = s4;
            if (U) reinterpret_cast<float4*>(U + (long)i * d)[q] = u4;
            if (Zx) reinterpret_cast<float4*>(Zx + (long)i * ldz)[q] = x4c;
            reinterpret_cast<float4*>(Zy + (long)i * ldz)[q] = y4;
            const float xs[4] = {x4c.x, x4c.y, x4c.z, x4c.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
            float xn[4], yn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xn[e] = norm_split ? split_value(xs[e]) : xs[e];
                yn[e] = norm_split ? split_value(ys[e]) : ys[e];
            }
            nx += (xn[0] * xn[0] + xn[1] * xn[1]) + (xn[2] * xn[2] + xn[3] * xn[3]);
            ny += (yn[0] * yn[0] + yn[1] * yn[1]) + (yn[2] * yn[2] + yn[3] * yn[3]);
        }
    }
    nx = wave_sum(nx);
    ny = wave_sum(ny);
    if (lane == 0) {
        if (sqx) sqx[i] = nx;
        sqy[i] = ny;
    }
}

// ---- the same forward fused with the operand preparation of the bf16x3 MMD kernels (vgan_mmd_bf3_prepare) -----------
// A workgroup owns 8 consecutive batch rows (two per wave) and emits, besides everything mask_forward_vec_kernel emits,
// the split images of both the X row and the Y = U*X row: row-major (Zh, Zl) straight from registers, transposed
// (ZTh, ZTl: [feature][row]) through an LDS tile, so that 8 rows leave as one 16-byte store per feature and image.  This
// removes the separate preparation launch (~6 us of a ~120 us step at d = 784) and its 13 MB re-read of Z.
// XX: workgroups past the row groups run the X-X tiles of this step's Gram (mmd_xx.hpp), one tile each.
template <int NT, bool XX, bool CHAIN>
__global__ __launch_bounds__(512) void mask_forward_bf3_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ data,
                                                                 int ldd, RowSel rows, float* __restrict__ S, float* __restrict__ Z,
                                                                 int ldz, float* __restrict__ sq, unsigned short* __restrict__ Zh,
                                                                 unsigned short* __restrict__ Zl, int kp, unsigned short* __restrict__ ZTh,
                                                                 unsigned short* __restrict__ ZTl, int kn, int n, int d,
                                                                 const float* __restrict__ center, int write_x, int mask_blocks,
                                                                 XXJob xx, LogitsChain ch) {
    constexpr int R = 8;  // rows per workgroup = waves per workgroup (512 threads; 4 rows in 256 threads measured the same: 11.7 us)
    if constexpr (XX) {
        __shared__ __attribute__((aligned(16))) char xx_lds[GemmBF3<64>::kLdsBytes];
        __shared__ float xx_red[4];
        if ((int)blockIdx.x >= mask_blocks) {  // block-uniform
            xx_tile_body(xx, blockIdx.x - mask_blocks, xx_lds, xx_red);
            return;
        }
    }
    extern __shared__ __attribute__((aligned(16))) unsigned short tile[];  // [4 images: Xh, Xl, Yh, Yl][R][ldt]
    const int ldt = 4 * (d >> 2) + 8;                     // bf16 elements per tile row (8-byte stores stay aligned)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = d >> 2;
    // Workgroup -> row group: block b runs on XCD b % 8 (private L2 each).  The eight row groups that share the 128-byte
    // lines of the transposed images (8 rows = 16 bytes each) must meet in ONE L2, or every line is written back partially
    // by up to eight of them: XCD x takes the contiguous range of groups [x C, (x + 1) C).
    const int groups = n / R, C = (groups + 7) / 8;
    const int grp = (blockIdx.x % 8) * C + blockIdx.x / 8;
    if (grp >= groups) return;
    const int i0 = grp * R;
    const float tau = 1.0f / (float)d;
    {
        const int lr = wave, i = i0 + lr;
        float4 v[NT];
        // CHAIN (needs ZTh == NULL: the dynamic LDS region holds the At_4 chunk instead of the transposed tile)
        if constexpr (CHAIN) chain_logits<NT>(ch, reinterpret_cast<float*>(tile), d, i, i < n, v);
        if (i < n) {
            const float4* x4 = reinterpret_cast<const float4*>(logits + (long)i * ldl);
            const float4* xr4 = reinterpret_cast<const float4*>(data + rows(i) * ldd);
            float4 xv[NT], cv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int q = lane + 64 * t;
                const bool ok = q < nq;
                if constexpr (!CHAIN) v[t] = ok ? x4[min(q, nq - 1)] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                xv[t] = xr4[min(q, nq - 1)];
                cv[t] = center ? reinterpret_cast<const float4*>(center)[min(q, nq - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t) m = fmaxf(m, fmaxf(fmaxf(v[t].x, v[t].y), fmaxf(v[t].z, v[t].w)));
            m = wave_max(m);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                v[t].x = expf(v[t].x - m); v[t].y = expf(v[t].y - m); v[t].z = expf(v[t].z - m); v[t].w = expf(v[t].w - m);
                sum += (v[t].x + v[t].y) + (v[t].z + v[t].w);
            }
            sum = wave_sum(sum);
            float nx = 0.f, ny = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int q = lane + 64 * t;
                if (q < nq) {
                    const float4 s4 = make_float4(v[t].x / sum, v[t].y / sum, v[t].z / sum, v[t].w / sum);
                    const float4 u4 = make_float4(s4.x < tau ? s4.x : 1.f, s4.y < tau ? s4.y : 1.f, s4.z < tau ? s4.z : 1.f, s4.w < tau ? s4.w : 1.f);
                    // centred operand (see mask_forward_vec_kernel); the row norms are those of the SPLIT values hi + lo, the
                    // numbers the Gram kernel actually multiplies, so that L = s_i + s_j - 2 g is |zhat_i - zhat_j|^2 exactly
                    const float4 x4c = make_float4(xv[t].x - cv[t].x, xv[t].y - cv[t].y, xv[t].z - cv[t].z, xv[t].w - cv[t].w);
                    const float4 y4 = make_float4(project_centred(u4.x, xv[t].x, cv[t].x), project_centred(u4.y, xv[t].y, cv[t].y), project_centred(u4.z, xv[t].z, cv[t].z),
                                              project_centred(u4.w, xv[t].w, cv[t].w));
                    reinterpret_cast<float4*>(S + (long)i * d)[q] = s4;
                    if (write_x) reinterpret_cast<float4*>(Z + (long)i * ldz)[q] = x4c;
                    reinterpret_cast<float4*>(Z + (long)(n + i) * ldz)[q] = y4;
                    const float xs[4] = {x4c.x, x4c.y, x4c.z, x4c.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
                    unsigned short h[2][4], l[2][4];
                    float xn[4], yn[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        split_bf16(xs[e], h[0][e], l[0][e]);
                        split_bf16(ys[e], h[1][e], l[1][e]);
                        xn[e] = bf16_val(h[0][e]) + bf16_val(l[0][e]);
                        yn[e] = bf16_val(h[1][e]) + bf16_val(l[1][e]);
                    }
                    nx += (xn[0] * xn[0] + xn[1] * xn[1]) + (xn[2] * xn[2] + xn[3] * xn[3]);
                    ny += (yn[0] * yn[0] + yn[1] * yn[1]) + (yn[2] * yn[2] + yn[3] * yn[3]);
#pragma unroll
                    for (int im = 0; im < 2; ++im) {  // 0: X row i, 1: Y row n + i
                        const uint2 ph = make_uint2((unsigned)h[im][0] | ((unsigned)h[im][1] << 16), (unsigned)h[im][2] | ((unsigned)h[im][3] << 16));
                        const uint2 pl = make_uint2((unsigned)l[im][0] | ((unsigned)l[im][1] << 16), (unsigned)l[im][2] | ((unsigned)l[im][3] << 16));
                        const long grow = (long)(im * n + i) * kp + 4 * q;
                        if (im == 1 || write_x) {  // (the X half may already be in place: vgan_gather_rows_split ran ahead)
                            *reinterpret_cast<uint2*>(Zh + grow) = ph;
                            *reinterpret_cast<uint2*>(Zl + grow) = pl;
                        }
                        if (ZTh != nullptr) {
                            *reinterpret_cast<uint2*>(tile + ((2 * im) * R + lr) * ldt + 4 * q) = ph;
                            *reinterpret_cast<uint2*>(tile + ((2 * im + 1) * R + lr) * ldt + 4 * q) = pl;
                        }
                    }
                }
            }
            nx = wave_sum(nx);
            ny = wave_sum(ny);
            if (lane == 0) {
                if (write_x) sq[i] = nx;
                sq[n + i] = ny;
            }
        } else if (ZTh != nullptr) {  // rows past the batch: their tile rows are read by the transposed store below (full 16-byte pieces)
            for (int c = lane; c < ldt; c += 64)
#pragma unroll
                for (int im = 0; im < 4; ++im) tile[(im * R + lr) * ldt + c] = 0;
        }
    }
    if (ZTh == nullptr) return;  // (uniform) no transposed images wanted: the backward product reads the row-major ones
    __syncthreads();
    // transposed images: features (j, j+1), rows i0 .. i0+7 of image X (columns i0..) and Y (columns n + i0 ..): per image
    // eight 4-byte LDS reads (two features at once) and one 16-byte store per feature
    const int npairs = d >> 1;
    for (int item = threadIdx.x; item < 4 * npairs; item += 512) {
        const int im = item / npairs, j = 2 * (item - im * npairs);
        unsigned w0[4], w1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned a = *reinterpret_cast<const unsigned*>(tile + (im * R + 2 * e) * ldt + j);
            const unsigned c = *reinterpret_cast<const unsigned*>(tile + (im * R + 2 * e + 1) * ldt + j);
            w0[e] = (a & 0xFFFFu) | (c << 16);
            w1[e] = (a >> 16) | (c & 0xFFFF0000u);
        }
        unsigned short* dst = ((im & 1) ? ZTl : ZTh) + (long)j * kn + (im >> 1) * n + i0;
        *reinterpret_cast<uint4*>(dst) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
        *reinterpret_cast<uint4*>(dst + kn) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
    }
}

template <int NT>
__global__ __launch_bounds__(kBlock) void mask_backward_vec_kernel(const float* __restrict__ gU, int ldg, const float* __restrict__ S,
                                                                  int lds, const unsigned long long* __restrict__ colkey,
                                                                  float pen_weight, int row_offset, float* __restrict__ dlogits,
                                                                  int ldo, int n, int d, int nslabs, long slab_stride) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const int nq = d >> 2;
    const float tau = 1.0f / (float)d;
    const float pg = -pen_weight / (float)d;
    const unsigned me = (unsigned)(row_offset + i);
    float4 sv[NT], gs[NT];
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = min(lane + 64 * t, nq - 1);
        const bool ok = lane + 64 * t < nq;
        sv[t] = reinterpret_cast<const float4*>(S + (long)i * lds)[q];
        float4 g = reinterpret_cast<const float4*>(gU + (long)i * ldg)[q];
        for (int sl = 1; sl < nslabs; ++sl) {
            const float4 b = reinterpret_cast<const float4*>(gU + sl * slab_stride + (long)i * ldg)[q];
            g.x += b.x; g.y += b.y; g.z += b.z; g.w += b.w;
        }
        if (colkey != nullptr) {
            const ulonglong2 k0 = reinterpret_cast<const ulonglong2*>(colkey)[2 * q], k1 = reinterpret_cast<const ulonglong2*>(colkey)[2 * q + 1];
            if (colkey_row(k0.x) == me) g.x += pg;
            if (colkey_row(k0.y) == me) g.y += pg;
            if (colkey_row(k1.x) == me) g.z += pg;
            if (colkey_row(k1.y) == me) g.w += pg;
        }
        gs[t] = make_float4(ok && sv[t].x < tau ? g.x : 0.f, ok && sv[t].y < tau ? g.y : 0.f, ok && sv[t].z < tau ? g.z : 0.f,
                            ok && sv[t].w < tau ? g.w : 0.f);
        dot += (gs[t].x * sv[t].x + gs[t].y * sv[t].y) + (gs[t].z * sv[t].z + gs[t].w * sv[t].w);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = lane + 64 * t;
        if (q < nq)
            reinterpret_cast<float4*>(dlogits + (long)i * ldo)[q] =
                make_float4(sv[t].x * (gs[t].x - dot), sv[t].y * (gs[t].y - dot), sv[t].z * (gs[t].z - dot), sv[t].w * (gs[t].w - dot));
    }
}

// dlogits = S * (g_s - sum_j g_s S),  g_s = [S < 1/d] * (gU + penalty gradient at the column arg-max row)
__global__ __launch_bounds__(kBlock) void mask_backward_kernel(const float* __restrict__ gU, int ldg, const float* __restrict__ S,
                                                              int lds, const unsigned long long* __restrict__ colkey,
                                                              float pen_weight, int row_offset, float* __restrict__ dlogits,
                                                              int ldo, int n, int d, int nslabs, long slab_stride) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float tau = 1.0f / (float)d;
    const float pg = -pen_weight / (float)d;
    const float* g = gU + (long)i * ldg;
    const float* s = S + (long)i * lds;
    const unsigned me = (unsigned)(row_offset + i);
    // gU may arrive as split-K slabs of the MMD backward GEMM: the sum over slabs is taken here (fixed order)
    auto grad = [&](int j) {
        float gv = g[j];
        for (int q = 1; q < nslabs; ++q) gv += g[q * slab_stride + j];
        if (colkey != nullptr && colkey_row(colkey[j]) == me) gv += pg;
        return gv;
    };
    float dot = 0.f;
    for (int j = lane; j < d; j += 64) {
        const float sv = s[j];
        dot = fmaf(sv < tau ? grad(j) : 0.f, sv, dot);
    }
    dot = wave_sum(dot);
    for (int j = lane; j < d; j += 64) {
        const float sv = s[j];
        dlogits[(long)i * ldo + j] = sv * ((sv < tau ? grad(j) : 0.f) - dot);
    }
}

// column arg-max of U = (S < tau ? S : 1) over a chunk of rows: grid (ceil(d/64), chunks); body in vgan_common.hpp
__global__ __launch_bounds__(kBlock) void colmax_partial_kernel(const float* __restrict__ S, int lds, int row_offset,
                                                               unsigned long long* __restrict__ part, int n, int d, int from_softmax) {
    colmax_partial_body<4>(S, lds, row_offset, part, n, d, from_softmax, blockIdx.x, blockIdx.y);
}
__global__ void colmax_final_kernel(const unsigned long long* __restrict__ part, int chunks, unsigned long long* __restrict__ colkey,
                                    int d) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    unsigned long long b = 0ull;
    for (int c = 0; c < chunks; ++c) {
        const unsigned long long k = part[(long)c * d + j];
        b = k > b ? k : b;
    }
    colkey[j] = b;
}

// out[i] = data[rows[i]] with squared norms: the batch gather for rows whose mask another rank owns
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ data, int ldd, RowSel rows,
                                                            float* __restrict__ out, int ldo, float* __restrict__ sq, int n, int d) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* x = data + rows(i) * ldd;
    float* o = out + (long)i * ldo;
    float nx = 0.f;
    for (int j = lane; j < d; j += 64) {
        const float v = x[j];
        o[j] = v;
        nx = fmaf(v, v, nx);
    }
    nx = wave_sum(nx);
    if (lane == 0 && sq) sq[i] = nx;
}

// The X half of the MMD operand alone, for a batch whose mask is not known yet: Zx[i] = data[rows[i]] - c with its norm and
// (optionally) its split images.  The data-parallel step runs this for the NEXT batch while the gradient all-reduce is in
// flight, so that the X-X tiles of the next Gram (which need nothing else) can run behind the collective too.
__global__ __launch_bounds__(kBlock) void gather_center_split_kernel(const float* __restrict__ data, int ldd, RowSel rows,
                                                                    const float* __restrict__ center, float* __restrict__ out, int ldo,
                                                                    float* __restrict__ sq, int norm_split,
                                                                    unsigned short* __restrict__ Zh, unsigned short* __restrict__ Zl,
                                                                    int kp, int n, int d) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* x = data + rows(i) * ldd;
    float nx = 0.f;
    for (int j = lane; j < d; j += 64) {
        const float v = x[j] - (center ? center[j] : 0.f);
        if (out) out[(long)i * ldo + j] = v;
        unsigned short hi, lo;
        split_bf16(v, hi, lo);
        if (Zh) {
            Zh[(long)i * kp + j] = hi;
            Zl[(long)i * kp + j] = lo;
        }
        const float vn = norm_split ? bf16_val(hi) + bf16_val(lo) : v;
        nx = fmaf(vn, vn, nx);
    }
    nx = wave_sum(nx);
    if (lane == 0 && sq) sq[i] = nx;
}

// 16 bytes per lane (d % 4 == 0, aligned bases and leading dimensions): the form the step engine's shapes take
__global__ __launch_bounds__(kBlock) void gather_center_split_vec_kernel(const float* __restrict__ data, int ldd, RowSel rows,
                                                                        const float* __restrict__ center, float* __restrict__ out, int ldo,
                                                                        float* __restrict__ sq, int norm_split,
                                                                        unsigned short* __restrict__ Zh, unsigned short* __restrict__ Zl,
                                                                        int kp, int n, int d) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float4* x4 = reinterpret_cast<const float4*>(data + rows(i) * ldd);
    const int nq = d >> 2;
    float nx = 0.f;
    for (int q = lane; q < nq; q += 64) {
        float4 v = x4[q];
        if (center) {
            const float4 c = reinterpret_cast<const float4*>(center)[q];
            v.x -= c.x; v.y -= c.y; v.z -= c.z; v.w -= c.w;
        }
        if (out) reinterpret_cast<float4*>(out + (long)i * ldo)[q] = v;
        const float vs[4] = {v.x, v.y, v.z, v.w};
        unsigned short h[4], l[4];
        float vn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            split_bf16(vs[e], h[e], l[e]);
            vn[e] = norm_split ? bf16_val(h[e]) + bf16_val(l[e]) : vs[e];
        }
        if (Zh) {
            *reinterpret_cast<uint2*>(Zh + (long)i * kp + 4 * q) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
            *reinterpret_cast<uint2*>(Zl + (long)i * kp + 4 * q) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
        }
        nx += (vn[0] * vn[0] + vn[1] * vn[1]) + (vn[2] * vn[2] + vn[3] * vn[3]);
    }
    nx = wave_sum(nx);
    if (lane == 0 && sq) sq[i] = nx;
}

__global__ void mask_from_softmax_kernel(const float* __restrict__ S, int lds, float* __restrict__ U, int ldu, int n, int d) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * d) return;
    const int i = (int)(idx / d), j = (int)(idx % d);
    const float sv = S[(long)i * lds + j];
    U[(long)i * ldu + j] = sv < 1.0f / (float)d ? sv : 1.0f;
}

}  // namespace vgan

using namespace vgan;

extern "C" int vgan_mask_project_forward(const float* logits, int ldl, const float* data, int ldd, const int32_t* rows,
                                         const uint64_t* row_cursor, int row_batches, int row_stride, int row_offset, float* S,
                                         float* U, float* Zx, float* Zy, int ldz, float* sqx, float* sqy, int n, int d,
                                         const float* center, int norm_split, const vgan_logits_chain* chain, vgan_stream_t stream) {
    VGAN_CHECK_ARG((logits || chain) && data && S && Zy && sqy && n > 0 && d > 0 && (logits == nullptr || ldl >= d) && ldd >= d && ldz >= d);
    VGAN_CHECK_ARG(row_batches >= 1 && row_stride >= 0 && row_offset >= 0);
    LogitsChain ch{};
    if (chain != nullptr) {
        VGAN_CHECK_ARG(chain->za && chain->At4 && chain->e0 > 0 && chain->e0 % 4 == 0 && chain->ldza >= chain->e0 && chain->ldat >= chain->e0 &&
                       chain->ldat % 4 == 0 && aligned16(chain->At4));
        ch = LogitsChain{chain->za, chain->At4, chain->ldza, chain->ldat, chain->e0};
        if (logits == nullptr) { logits = chain->za; ldl = 4; }  // never dereferenced: keeps the alignment checks below meaningful
    }
    const RowSel sel{rows, reinterpret_cast<const unsigned long long*>(row_cursor), row_batches, row_stride, row_offset};
    const dim3 grid((n + kRowsPerBlock - 1) / kRowsPerBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    // row-in-registers path: one wave per row, d / 256 float4 per lane and array -- up to d = 4096 (16 per lane); the in-launch
    // logits (chain) stage At_4 through LDS and stop at d = 1024
    const bool vec = (d % 4 == 0) && (d <= (chain != nullptr ? 1024 : 4096)) && (ldl % 4 == 0) && (ldd % 4 == 0) && (ldz % 4 == 0) && aligned16(logits) &&
                     aligned16(data) && aligned16(S) && (U == nullptr || aligned16(U)) && (Zx == nullptr || aligned16(Zx)) && aligned16(Zy) && (center == nullptr || aligned16(center));
    if (vec) {
        const int nt = (d / 4 + 63) / 64;
        const size_t chain_bytes = (size_t)kChainK * chain_pitch(d) * sizeof(float);
#define VGAN_LAUNCH_FWD(NT)                                                                                                               \
    do {                                                                                                                                  \
        if (chain != nullptr) {                                                                                                           \
            if (chain_bytes > 64 * 1024)                                                                                                  \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_forward_vec_kernel<NT, true>),                                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_bytes);                                  \
            hipLaunchKernelGGL((mask_forward_vec_kernel<NT, true>), grid, block, chain_bytes, st, logits, ldl, data, ldd, sel, S, U, Zx, Zy, ldz, \
                               sqx, sqy, n, d, center, norm_split, ch);                                                                   \
        } else                                                                                                                            \
            hipLaunchKernelGGL((mask_forward_vec_kernel<NT, false>), grid, block, 0, st, logits, ldl, data, ldd, sel, S, U, Zx, Zy, ldz, sqx, \
                               sqy, n, d, center, norm_split, ch);                                                                        \
    } while (0)
        if (nt == 1) VGAN_LAUNCH_FWD(1); else if (nt == 2) VGAN_LAUNCH_FWD(2); else if (nt == 3) VGAN_LAUNCH_FWD(3); else if (nt == 4) VGAN_LAUNCH_FWD(4);
        else if (nt <= 8)
            hipLaunchKernelGGL((mask_forward_vec_kernel<8, false>), grid, block, 0, st, logits, ldl, data, ldd, sel, S, U, Zx, Zy, ldz, sqx, sqy, n, d,
                               center, norm_split, ch);
        else
            hipLaunchKernelGGL((mask_forward_vec_kernel<16, false>), grid, block, 0, st, logits, ldl, data, ldd, sel, S, U, Zx, Zy, ldz, sqx, sqy, n, d,
                               center, norm_split, ch);
#undef VGAN_LAUNCH_FWD
    } else {
        VGAN_CHECK_ARG(chain == nullptr);  // the in-launch logits need the row-in-registers path (d % 4 == 0, d <= 1024, aligned)
        hipLaunchKernelGGL(mask_forward_kernel<true>, grid, block, 0, st, logits, ldl, data, ldd, sel, S, U, Zx, Zy, ldz, sqx, sqy, n, d, center,
                           norm_split);
    }
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mask_project_forward_bf3(const float* logits, int ldl, const float* data, int ldd, const int32_t* rows,
                                             const uint64_t* row_cursor, int row_batches, int row_stride, float* S, float* Z, int ldz,
                                             float* sq, uint16_t* Zh, uint16_t* Zl, int kp, uint16_t* ZTh, uint16_t* ZTl, int kn,
                                             int n, int d, const float* center, int write_x, const vgan_xx_job* xxjob,
                                             const vgan_logits_chain* chain, vgan_stream_t stream) {
    VGAN_CHECK_ARG((logits || chain) && data && S && Z && sq && Zh && Zl && n > 0 && d > 0 && (logits == nullptr || ldl >= d) && ldd >= d &&
                   ldz >= d);
    LogitsChain ch{};
    if (chain != nullptr) {
        VGAN_CHECK_ARG(ZTh == nullptr && chain->za && chain->At4 && chain->e0 > 0 && chain->e0 % 4 == 0 && chain->ldza >= chain->e0 &&
                       chain->ldat >= chain->e0 && chain->ldat % 4 == 0 && aligned16(chain->At4));
        ch = LogitsChain{chain->za, chain->At4, chain->ldza, chain->ldat, chain->e0};
        if (logits == nullptr) { logits = chain->za; ldl = 4; }
    }
    VGAN_CHECK_ARG((ZTh == nullptr) == (ZTl == nullptr));
    VGAN_CHECK_ARG(row_batches >= 1 && row_stride >= 0 && kp >= d && kp % 64 == 0 && (ZTh == nullptr || (kn >= 2 * n && kn % 64 == 0)));
    VGAN_CHECK_ARG(write_x || ZTh == nullptr);  // the transposed images are always written whole
    // shape contract of the fused path (callers fall back to vgan_mask_project_forward + vgan_mmd_bf3_prepare otherwise)
    VGAN_CHECK_ARG(d % 4 == 0 && d <= 1024 && n % 8 == 0 && ldl % 4 == 0 && ldd % 4 == 0 && ldz % 4 == 0);
    VGAN_CHECK_ARG(aligned16(logits) && aligned16(data) && aligned16(S) && aligned16(Z) && aligned16(Zh) && aligned16(Zl) &&
                   (ZTh == nullptr || (aligned16(ZTh) && aligned16(ZTl))) && (center == nullptr || aligned16(center)));
    const RowSel sel{rows, reinterpret_cast<const unsigned long long*>(row_cursor), row_batches, row_stride, 0};
    const int mask_blocks = 8 * ((n / 8 + 7) / 8);
    XXJob xx{};
    int xx_blocks = 0;
    if (xxjob != nullptr) {
        const vgan_xx_job& j = *xxjob;
        VGAN_CHECK_ARG(ZTh == nullptr && j.Dh && j.Dl && j.dsq && j.tiles && j.bw && j.partial && j.ntiles > 0 && j.ldd >= d && j.ldd % 64 == 0);
        VGAN_CHECK_ARG(aligned16(j.Dh) && aligned16(j.Dl) && (reinterpret_cast<uintptr_t>(j.partial) & 15) == 0);
        xx = XXJob{j.Dh, j.Dl, j.dsq, rows, reinterpret_cast<const unsigned long long*>(row_cursor), reinterpret_cast<const TileDesc*>(j.tiles),
                   j.bw, j.partial, j.ldd, row_batches, row_stride, j.ntiles, n};
        xx_blocks = j.ntiles;
    }
    const dim3 grid(mask_blocks + xx_blocks), block(512);
    const size_t shmem = ZTh != nullptr ? (size_t)4 * 8 * (d + 8) * sizeof(unsigned short)
                                        : (chain != nullptr ? (size_t)kChainK * chain_pitch(d) * sizeof(float) : 0);
    hipStream_t st = (hipStream_t)stream;
    const int nt = (d / 4 + 63) / 64;
    // (dynamic LDS above 64 KB -- d close to 1024 -- needs the opt-in; setting it is idempotent and cheap)
#define VGAN_LAUNCH_K3(NT, XXF, CHF)                                                                                                \
    do {                                                                                                                            \
        if (shmem > 64 * 1024)                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_forward_bf3_kernel<NT, XXF, CHF>),                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                                      \
        hipLaunchKernelGGL((mask_forward_bf3_kernel<NT, XXF, CHF>), grid, block, shmem, st, logits, ldl, data, ldd, sel, S, Z, ldz, sq, \
                           Zh, Zl, kp, ZTh, ZTl, kn, n, d, center, write_x, mask_blocks, xx, ch);                                    \
    } while (0)
#define VGAN_LAUNCH_FWD3(NT)                                                                                                        \
    do {                                                                                                                            \
        if (xx_blocks > 0 && chain != nullptr) VGAN_LAUNCH_K3(NT, true, true);                                                      \
        else if (xx_blocks > 0) VGAN_LAUNCH_K3(NT, true, false);                                                                    \
        else if (chain != nullptr) VGAN_LAUNCH_K3(NT, false, true);                                                                 \
        else VGAN_LAUNCH_K3(NT, false, false);                                                                                      \
    } while (0)
    if (nt == 1) VGAN_LAUNCH_FWD3(1); else if (nt == 2) VGAN_LAUNCH_FWD3(2); else if (nt == 3) VGAN_LAUNCH_FWD3(3); else VGAN_LAUNCH_FWD3(4);
#undef VGAN_LAUNCH_FWD3
#undef VGAN_LAUNCH_K3
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_upper_softmax_forward(const float* logits, int ldl, float* S, float* U, int n, int d, vgan_stream_t stream) {
    VGAN_CHECK_ARG(logits && S && n > 0 && d > 0 && ldl >= d);
    hipLaunchKernelGGL(mask_forward_kernel<false>, dim3((n + kRowsPerBlock - 1) / kRowsPerBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, logits, ldl, nullptr, 0, RowSel{nullptr, nullptr, 1, 0, 0}, S, U, nullptr, nullptr, 0, nullptr, nullptr, n, d, nullptr, 0);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mask_backward(const float* gU, int ldg, int nslabs, int64_t slab_stride, const float* S, int lds,
                                  const uint64_t* colkey, float pen_weight, int row_offset, float* dlogits, int ldo, int n, int d,
                                  vgan_stream_t stream) {
    VGAN_CHECK_ARG(gU && S && dlogits && n > 0 && d > 0 && ldg >= d && lds >= d && ldo >= d && nslabs >= 1);
    const dim3 grid((n + kRowsPerBlock - 1) / kRowsPerBlock), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long* ck = reinterpret_cast<const unsigned long long*>(colkey);
    const bool vec = (d % 4 == 0) && (d <= 4096) && (ldg % 4 == 0) && (lds % 4 == 0) && (ldo % 4 == 0) && (slab_stride % 4 == 0) &&
                     aligned16(gU) && aligned16(S) && aligned16(dlogits) && (colkey == nullptr || aligned16(colkey));
    if (vec) {
        const int nt = (d / 4 + 63) / 64;
#define VGAN_LAUNCH_BWD(NT) hipLaunchKernelGGL(mask_backward_vec_kernel<NT>, grid, block, 0, st, gU, ldg, S, lds, ck, pen_weight, row_offset, dlogits, ldo, n, d, nslabs, (long)slab_stride)
        if (nt == 1) VGAN_LAUNCH_BWD(1); else if (nt == 2) VGAN_LAUNCH_BWD(2); else if (nt == 3) VGAN_LAUNCH_BWD(3); else if (nt == 4) VGAN_LAUNCH_BWD(4);
        else if (nt <= 8) VGAN_LAUNCH_BWD(8); else VGAN_LAUNCH_BWD(16);
#undef VGAN_LAUNCH_BWD
    } else
        hipLaunchKernelGGL(mask_backward_kernel, grid, block, 0, st, gU, ldg, S, lds, ck, pen_weight, row_offset, dlogits, ldo, n, d, nslabs,
                           (long)slab_stride);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_gather_rows(const float* data, int ldd, const int32_t* rows, const uint64_t* row_cursor, int row_batches,
                                int row_stride, int row_offset, float* out, int ldo, float* sq, int n, int d, vgan_stream_t stream) {
    VGAN_CHECK_ARG(data && out && n > 0 && d > 0 && ldd >= d && ldo >= d && row_batches >= 1 && row_stride >= 0 && row_offset >= 0);
    const RowSel sel{rows, reinterpret_cast<const unsigned long long*>(row_cursor), row_batches, row_stride, row_offset};
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + kRowsPerBlock - 1) / kRowsPerBlock), dim3(kBlock), 0, (hipStream_t)stream, data,
                       ldd, sel, out, ldo, sq, n, d);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_gather_rows_split(const float* data, int ldd, const int32_t* rows, const uint64_t* row_cursor, int row_batches,
                                      int row_stride, int row_offset, const float* center, float* out, int ldo, float* sq,
                                      int norm_split, uint16_t* Zh, uint16_t* Zl, int kp, int n, int d, vgan_stream_t stream) {
    VGAN_CHECK_ARG(data && n > 0 && d > 0 && ldd >= d && row_batches >= 1 && row_stride >= 0 && row_offset >= 0);
    VGAN_CHECK_ARG((out == nullptr || ldo >= d) && (out || sq || Zh) && (Zh == nullptr) == (Zl == nullptr) && (Zh == nullptr || kp >= d));
    const RowSel sel{rows, reinterpret_cast<const unsigned long long*>(row_cursor), row_batches, row_stride, row_offset};
    const bool vec = (d % 4 == 0) && (ldd % 4 == 0) && aligned16(data) && (center == nullptr || aligned16(center)) &&
                     (out == nullptr || (ldo % 4 == 0 && aligned16(out))) && (Zh == nullptr || (kp % 4 == 0 && aligned16(Zh) && aligned16(Zl)));
    const dim3 grid((n + kRowsPerBlock - 1) / kRowsPerBlock), block(kBlock);
    if (vec)
        hipLaunchKernelGGL(gather_center_split_vec_kernel, grid, block, 0, (hipStream_t)stream, data, ldd, sel, center, out, ldo, sq, norm_split,
                           Zh, Zl, kp, n, d);
    else
        hipLaunchKernelGGL(gather_center_split_kernel, grid, block, 0, (hipStream_t)stream, data, ldd, sel, center, out, ldo, sq, norm_split,
                           Zh, Zl, kp, n, d);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_colmax_chunks(int n) { return n > 0 ? (n + kColChunkRows - 1) / kColChunkRows : 0; }

extern "C" int vgan_colmax_partial(const float* S, int lds, int from_softmax, int row_offset, uint64_t* part, int n, int d,
                                   vgan_stream_t stream) {
    VGAN_CHECK_ARG(S && part && n > 0 && d > 0 && lds >= d);
    hipLaunchKernelGGL(colmax_partial_kernel, dim3((d + 63) / 64, vgan_colmax_chunks(n)), dim3(kBlock), 0, (hipStream_t)stream, S,
                       lds, row_offset, reinterpret_cast<unsigned long long*>(part), n, d, from_softmax);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_colmax(const float* S, int lds, int from_softmax, int row_offset, uint64_t* part, uint64_t* colkey, int n,
                           int d, vgan_stream_t stream) {
    VGAN_CHECK_ARG(S && part && colkey && n > 0 && d > 0 && lds >= d);
    const int chunks = vgan_colmax_chunks(n);
    hipLaunchKernelGGL(colmax_partial_kernel, dim3((d + 63) / 64, chunks), dim3(kBlock), 0, (hipStream_t)stream, S, lds, row_offset,
                       reinterpret_cast<unsigned long long*>(part), n, d, from_softmax);
    VGAN_CHECK_LAUNCH();
    hipLaunchKernelGGL(colmax_final_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(part), chunks, reinterpret_cast<unsigned long long*>(colkey), d);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mask_from_softmax(const float* S, int lds, float* U, int ldu, int n, int d, vgan_stream_t stream) {
    VGAN_CHECK_ARG(S && U && n > 0 && d > 0 && lds >= d && ldu >= d);
    const long total = (long)n * d;
    hipLaunchKernelGGL(mask_from_softmax_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, lds, U,
                       ldu, n, d);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}
