// Split-bf16 ("bf16x3") variants of the two dense contractions of the MMD, for large problems.
//
// gfx950 has no TF32; its fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s), the bf16 MFMA 16x faster.  Writing
// every operand as z = hi + lo with hi = bf16(z), lo = bf16(z - hi) (16 significant bits together) and computing
//     g = hi.hi' + hi.lo' + lo.hi'                         (three v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// drops only the lo.lo' term (2^-18 relative per product, random sign) and the rounding of lo (2^-17): the Gram entry
// keeps ~3e-7 relative accuracy at K = 784 -- the level of an fp32 fma chain of that length -- at 3/16 of the fp32
// MFMA time.  The C/D fragment layout of the MFMA does not depend on the input type, so the fused epilogues (distance,
// exp, squaring chain, block sums, gradient weights) are the fp32 kernels' code, character for character.
//
// Operands are prepared once per step by vgan_mmd_bf3_prepare: Z -> (Zh, Zl) row-major for the Gram and (ZTh, ZTl),
// the transposed copy, for the backward product, whose B fragment needs 8 consecutive k (= Z rows) per lane.
#include <stdlib.h>

#include "gemm_core.hpp"
#include "mmd_common.hpp"

namespace vgan {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // plain vector type: assignable in any address space
typedef __attribute__((address_space(3))) u32x4 lds_u4;
typedef __attribute__((address_space(3))) unsigned short lds_u16;

__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float bf16_val(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ void split_bf16(float v, unsigned short& hi, unsigned short& lo) {
    hi = bf16_bits(v);
    lo = bf16_bits(v - bf16_val(hi));
}

// ---- operand preparation: 64x64 tiles of Z -> row-major and transposed hi/lo images ---------------------------
// Zh/Zl [rows_pad, kp]  (kp = features padded to 64, zero filled);  ZTh/ZTl [kp, kn]  (kn = rows padded to 64)
__global__ __launch_bounds__(kBlock) void bf3_prepare_kernel(const float* __restrict__ Z, int ldz, int rows, int p,
                                                            unsigned short* __restrict__ Zh, unsigned short* __restrict__ Zl, int kp,
                                                            unsigned short* __restrict__ ZTh, unsigned short* __restrict__ ZTl, int kn) {
    __shared__ unsigned short th[64][66], tl[64][66];  // [row k][feature j], padded
    const int j0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // tx: feature within tile, ty: row group
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, j = j0 + tx;
        const float v = (k < rows && j < p) ? Z[(long)k * ldz + j] : 0.f;
        unsigned short hi, lo;
        split_bf16(v, hi, lo);
        th[r][tx] = hi;
        tl[r][tx] = lo;
        if (k < rows) {  // row-major image (rows beyond `rows` do not exist in Zh/Zl)
            Zh[(long)k * kp + j] = hi;
            Zl[(long)k * kp + j] = lo;
        }
    }
    __syncthreads();
    if (ZTh != nullptr) {
#pragma unroll 4
        for (int r = ty; r < 64; r += 4) {  // r: feature within tile, tx: row k within tile
            const long o = (long)(j0 + r) * kn + k0 + tx;
            ZTh[o] = th[tx][r];
            ZTl[o] = tl[tx][r];
        }
    }
}

// ---- the split-bf16 tile main loop: 64x64 output, 256 threads (2x2 waves), K tile of 64 -----------------------
struct GemmBF3 {
    static constexpr int BK = 64;                  // bf16 elements of K per tile
    static constexpr int ROWB = (BK + 8) * 2;      // bytes per LDS row: 128 + 16 pad (stride = 36 dwords = 4 * odd)
    static constexpr int PART = 64 * ROWB;         // one operand part (64 rows)
    static constexpr int BUF = 4 * PART;           // Ah | Al | Bh | Bl
    static constexpr int kLdsBytes = 2 * BUF;      // double buffered: 73,728 B

    struct Stage {
        u32x4 v[4][2];  // [part][r]
        const char* src[4][2];
        int lofs[2];
        __device__ __forceinline__ void init(const unsigned short* Ah, const unsigned short* Al, long lda, int m0, int M,
                                             const unsigned short* Bh, const unsigned short* Bl, long ldb, int n0, int N, int tid) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = tid + kBlock * r, row = f >> 3, q = f & 7;
                const long ra = (long)min(m0 + row, M - 1) * lda + 8 * q, rb = (long)min(n0 + row, N - 1) * ldb + 8 * q;
                src[0][r] = reinterpret_cast<const char*>(Ah + ra);
                src[1][r] = reinterpret_cast<const char*>(Al + ra);
                src[2][r] = reinterpret_cast<const char*>(Bh + rb);
                src[3][r] = reinterpret_cast<const char*>(Bl + rb);
                lofs[r] = row * ROWB + q * 16;
            }
        }
        __device__ __forceinline__ void load(int k0) {  // K is a multiple of 64 by construction: no k guard
#pragma unroll
            for (int part = 0; part < 4; ++part)
#pragma unroll
                for (int r = 0; r < 2; ++r) v[part][r] = *reinterpret_cast<const u32x4*>(src[part][r] + 2 * (long)k0);
        }
        __device__ __forceinline__ void store(char __attribute__((address_space(3)))* buf) const {
#pragma unroll
            for (int part = 0; part < 4; ++part)
#pragma unroll
                for (int r = 0; r < 2; ++r) *(lds_u4*)(buf + part * PART + lofs[r]) = v[part][r];
        }
        // sum over k of the staged A values (hi + lo) of staged row r
        __device__ __forceinline__ float a_rowpart(int r) const {
            float s = 0.f;
            const unsigned* h = reinterpret_cast<const unsigned*>(&v[0][r]);
            const unsigned* l = reinterpret_cast<const unsigned*>(&v[1][r]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s += __uint_as_float(h[e] << 16) + __uint_as_float(h[e] & 0xFFFF0000u);
                s += __uint_as_float(l[e] << 16) + __uint_as_float(l[e] & 0xFFFF0000u);
            }
            return s;
        }
    };

    // acc (+)= A[m0.., :] . B[n0.., :]^T over K (multiple of 64).  SIDE_A: rs_lds[64] = sum_k A[m0 + m, k].
    template <bool SIDE_A>
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K,
                                               char* lds_generic, float* rs_generic, f32x16& acc) {
        typedef char __attribute__((address_space(3))) lds_c;
        lds_c* lds = (lds_c*)lds_generic;
        lds_f* rs_lds = (lds_f*)rs_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        const int fi = lane & 31, fh = lane >> 5;
        Stage st;
        st.init(Ah, Al, lda, m0, M, Bh, Bl, ldb, n0, N, tid);
        float rsum[2] = {0.f, 0.f};
        const int nk = K / BK;
        st.load(0);
        st.store(lds);
        if constexpr (SIDE_A) { rsum[0] += st.a_rowpart(0); rsum[1] += st.a_rowpart(1); }
        if (nk > 1) st.load(BK);
        __syncthreads();
        auto body = [&](int kt, auto store_next, auto load_next2) {
            const lds_c* buf = lds + (kt & 1) * BUF;
            const lds_c* pa = buf + (wm0 + fi) * ROWB + fh * 16;
            const lds_c* pb = buf + 2 * PART + (wn0 + fi) * ROWB + fh * 16;
            u32x4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {  // k16 step s: this lane's 8 consecutive k = 16 s + 8 fh ..
                ah[s] = *(const lds_u4*)(pa + s * 32);
                al[s] = *(const lds_u4*)(pa + PART + s * 32);
                bh[s] = *(const lds_u4*)(pb + s * 32);
                bl[s] = *(const lds_u4*)(pb + PART + s * 32);
            }
            if constexpr (decltype(store_next)::value) {
                st.store(lds + ((kt & 1) ^ 1) * BUF);
                if constexpr (SIDE_A) { rsum[0] += st.a_rowpart(0); rsum[1] += st.a_rowpart(1); }
            }
            if constexpr (decltype(load_next2)::value) st.load((kt + 2) * BK);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[s]), xl = __builtin_bit_cast(bf16x8, al[s]);
                const bf16x8 yh = __builtin_bit_cast(bf16x8, bh[s]), yl = __builtin_bit_cast(bf16x8, bl[s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc, 0, 0, 0);  // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
            }
            __builtin_amdgcn_iglp_opt(0);
            __syncthreads();
        };
        using T = std::true_type;
        using F = std::false_type;
        int kt = 0;
        for (; kt + 2 < nk; ++kt) body(kt, T{}, T{});
        if (kt + 1 < nk) {
            body(kt, T{}, F{});
            ++kt;
        }
        body(kt, F{}, F{});
        if constexpr (SIDE_A) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float s = rsum[r];
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                const int f = tid + kBlock * r;
                if ((f & 7) == 0) rs_lds[f >> 3] = s;
            }
            __syncthreads();
        }
    }
    __device__ static __forceinline__ int sub_row(int r) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int sub_col() {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave & 1) * 32 + (lane & 31);
    }
};

// ---- Gram tile + fused epilogue (see mmd.hip's mmd_gram_kernel; Wg leaves as a hi/lo bf16 pair) ----------------
__global__ __launch_bounds__(kBlock, 2) void mmd_gram_bf3_kernel(const unsigned short* __restrict__ Zh, const unsigned short* __restrict__ Zl,
                                                                int kp, const float* __restrict__ sq, int n,
                                                                const float* __restrict__ bw_ptr, const TileDesc* __restrict__ tiles,
                                                                int ntiles, unsigned short* __restrict__ Wh,
                                                                unsigned short* __restrict__ Wl, int ldw, int wrow0,
                                                                float* __restrict__ partial, ColmaxJob cj) {
    __shared__ __attribute__((aligned(16))) char lds[GemmBF3::kLdsBytes];
    __shared__ float red[8];
    if ((int)blockIdx.x >= ntiles) {
        const int cb = blockIdx.x - ntiles;
        colmax_partial_body<4>(cj.S, cj.lds, cj.row_offset, cj.part, cj.n, cj.d, cj.from_softmax, cb % cj.nbx, cb / cj.nbx);
        return;
    }
    const TileDesc td = tiles[blockIdx.x];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    GemmBF3::run<false>(Zh, Zl, kp, Zh, Zl, kp, td.r0, td.c0, td.rlim, td.clim, kp, lds, nullptr, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = td.c0 + GemmBF3::sub_col();
    const bool jok = j < td.clim;
    const float sj = sq[min(j, td.clim - 1)];
    const float bw = bw_ptr[0];
    const float c2 = -1.4426950408889634f / (4.f * bw);
    const float wscale = -((td.flags & VGAN_TF_NEG) ? -1.f : 1.f) * 2.f / ((float)n * (float)n * bw);
    const bool store = (td.flags & VGAN_TF_STORE) && Wh != nullptr;
    const bool mirror = store && (td.flags & VGAN_TF_MIRROR);
    float ksum = 0.f;
    unsigned short wh[16], wl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = td.r0 + GemmBF3::sub_row(r);
        const bool ok = jok && (i < td.rlim);
        const float si = sq[min(i, td.rlim - 1)];
        const float L = fmaxf(si + sj - 2.f * acc[r], 0.f);
        const float t = __builtin_amdgcn_exp2f(L * c2);
        const float t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
        ksum += ok ? ((t + t2) + (t4 + t8)) + t16 : 0.f;
        const float w = wscale * (((0.25f * t + 0.5f * t2) + (t4 + 2.f * t8)) + 4.f * t16);
        split_bf16(w, wh[r], wl[r]);
        if (store && ok) {
            const long o = (long)(i - wrow0) * ldw + j;
            Wh[o] = wh[r];
            Wl[o] = wl[r];
        }
    }
    if (mirror && jok) {  // W[j - wrow0, i]: registers 4q..4q+3 are 4 consecutive i -> one 8-byte store per image
        const int ibase = td.r0 + (wave >> 1) * 32 + 4 * (lane >> 5);
        const long rowo = (long)(j - wrow0) * ldw;
        const bool v4 = ((ldw & 3) == 0) && ((td.r0 & 3) == 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i0 = ibase + 8 * q;
            if (v4 && i0 + 3 < td.rlim) {
                *reinterpret_cast<uint2*>(Wh + rowo + i0) =
                    make_uint2((unsigned)wh[4 * q] | ((unsigned)wh[4 * q + 1] << 16), (unsigned)wh[4 * q + 2] | ((unsigned)wh[4 * q + 3] << 16));
                *reinterpret_cast<uint2*>(Wl + rowo + i0) =
                    make_uint2((unsigned)wl[4 * q] | ((unsigned)wl[4 * q + 1] << 16), (unsigned)wl[4 * q + 2] | ((unsigned)wl[4 * q + 3] << 16));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i0 + e < td.rlim) {
                        Wh[rowo + i0 + e] = wh[4 * q + e];
                        Wl[rowo + i0 + e] = wl[4 * q + e];
                    }
            }
        }
    }
    ksum = wave_sum(ksum);
    if (lane == 0) red[wave] = ksum;
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<float4*>(partial)[blockIdx.x] = make_float4((red[0] + red[1]) + (red[2] + red[3]), 0.f, 0.f, 0.f);
}

// ---- backward: out = 2 (rowsum(W) z - W . Z) * mul, W = Wh + Wl [nr, kn], Z^T = ZTh + ZTl [kp, kn] -------------
__global__ __launch_bounds__(kBlock, 2) void mmd_backward_bf3_kernel(const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl,
                                                                    int ldw, const unsigned short* __restrict__ ZTh,
                                                                    const unsigned short* __restrict__ ZTl, int kn,
                                                                    const float* __restrict__ Z, int ldz, int wrow0, int nr, int p,
                                                                    int ptiles, const float* __restrict__ mul, int ldmul,
                                                                    float* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) char lds[GemmBF3::kLdsBytes];
    __shared__ float rs[64];
    // XCD-aware order as in mmd_backward_kernel: down 4 row panels, then the next feature panel
    const int gx = ptiles, gy = (nr + 63) / 64, total = gx * gy;
    const int xcd = blockIdx.x % 8, kidx = blockIdx.x / 8;
    const int q = total / 8, r8 = total % 8;
    const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + kidx;
    const int band = t / (4 * gx), rem = t - band * 4 * gx;
    const int rows_in_band = min(4, gy - band * 4);
    const int m0 = (band * 4 + rem % rows_in_band) * 64, n0 = (rem / rows_in_band) * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    GemmBF3::run<true>(Wh, Wl, ldw, ZTh, ZTl, kn, m0, n0, nr, gx * 64, kn, lds, rs, acc);
    const int col = n0 + GemmBF3::sub_col();
    if (col >= p) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lrow = GemmBF3::sub_row(r), row = m0 + lrow;
        if (row < nr) {
            float v = 2.f * (rs[lrow] * Z[(long)(wrow0 + row) * ldz + col] - acc[r]);
            if (mul != nullptr) v *= mul[(long)row * ldmul + col];
            out[(long)row * ldo + col] = v;
        }
    }
}

}  // namespace vgan

using namespace vgan;

extern "C" int vgan_mmd_bf3_prepare(const float* Z, int ldz, int rows, int p, uint16_t* Zh, uint16_t* Zl, int kp, uint16_t* ZTh,
                                    uint16_t* ZTl, int kn, vgan_stream_t stream) {
    VGAN_CHECK_ARG(Z && Zh && Zl && rows > 0 && p > 0 && ldz >= p && kp >= p && kp % 64 == 0);
    VGAN_CHECK_ARG((ZTh == nullptr) == (ZTl == nullptr) && (ZTh == nullptr || (kn >= rows && kn % 64 == 0)));
    dim3 grid(kp / 64, (rows + 63) / 64);
    if (ZTh != nullptr) grid.y = kn / 64;
    hipLaunchKernelGGL(bf3_prepare_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, rows, p, Zh, Zl, kp, ZTh, ZTl, kn);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_gram_bf3(const uint16_t* Zh, const uint16_t* Zl, int kp, const float* sq, int n, const float* bw,
                                 const int32_t* tiles, int ntiles, uint16_t* Wh, uint16_t* Wl, int ldw, int wrow0, float* partial,
                                 const float* S, int lds, int from_softmax, int row_offset, uint64_t* colpart, int nrows, int d,
                                 vgan_stream_t stream) {
    VGAN_CHECK_ARG(Zh && Zl && sq && bw && tiles && partial && n > 0 && ntiles > 0 && kp > 0 && kp % 64 == 0);
    VGAN_CHECK_ARG((Wh == nullptr) == (Wl == nullptr) && (reinterpret_cast<uintptr_t>(partial) & 15) == 0);
    VGAN_CHECK_ARG(aligned16(Zh) && aligned16(Zl) && (Wh == nullptr || (aligned16(Wh) && aligned16(Wl))));
    ColmaxJob cj{};
    int extra = 0;
    if (S != nullptr) {
        VGAN_CHECK_ARG(colpart && nrows > 0 && d > 0 && lds >= d);
        cj = ColmaxJob{S, reinterpret_cast<unsigned long long*>(colpart), lds, row_offset, nrows, d, from_softmax, (d + 63) / 64};
        extra = cj.nbx * ((nrows + kColChunkRows - 1) / kColChunkRows);
    }
    hipLaunchKernelGGL(mmd_gram_bf3_kernel, dim3(ntiles + extra), dim3(kBlock), 0, (hipStream_t)stream, Zh, Zl, kp, sq, n, bw,
                       reinterpret_cast<const TileDesc*>(tiles), ntiles, Wh, Wl, ldw, wrow0, partial, cj);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_backward_bf3(const uint16_t* Wh, const uint16_t* Wl, int ldw, const uint16_t* ZTh, const uint16_t* ZTl, int kn,
                                     int kp, const float* Z, int ldz, int wrow0, int nr, int p, const float* mul, int ldmul,
                                     float* out, int ldo, vgan_stream_t stream) {
    VGAN_CHECK_ARG(Wh && Wl && ZTh && ZTl && Z && out && nr > 0 && p > 0 && kn > 0 && kn % 64 == 0 && kp >= p && kp % 64 == 0);
    VGAN_CHECK_ARG(ldw >= kn && ldz >= p && ldo >= p && (mul == nullptr || ldmul >= p) && wrow0 >= 0);
    VGAN_CHECK_ARG(aligned16(Wh) && aligned16(Wl) && aligned16(ZTh) && aligned16(ZTl) && ldw % 8 == 0);
    const int ptiles = (p + 63) / 64;
    dim3 grid(ptiles * ((nr + 63) / 64));
    hipLaunchKernelGGL(mmd_backward_bf3_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, Wh, Wl, ldw, ZTh, ZTl, kn, Z, ldz, wrow0, nr, p,
                       ptiles, mul, ldmul, out, ldo);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}
