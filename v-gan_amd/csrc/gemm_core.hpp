// fp32-MFMA tile engine shared by every dense contraction on the V-GAN hot path
// (Linear fwd/bwd, the Gram tile of the MMD, and the W.Z product of its backward).
//
// One workgroup = 256 threads = 4 waves laid out 2 (M) x 2 (N); each wave owns WM x WN
// sub-tiles of 32x32 computed with v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 64 cycles per
// issue per SIMD -- the fp32 matrix rate equals the fp32 vector peak on gfx950, so the MFMA
// pipe, not LDS or L2, is the bound; staging is therefore kept simple: register-staged,
// double-buffered LDS, one barrier per K tile, loads for tile t+1 issued before the MFMAs of t).
//
// LDS image of both operands is [k][mn] (mn contiguous): the A/B fragment of the 32x32x2 MFMA is
// lane l -> (mn = l & 31, k = l >> 5), so a fragment read is one conflict-free ds_read_b32.
#pragma once
#include "vgan_common.hpp"

namespace vgan {

enum : int { KC = 0, MC = 1 };  // KC: elem(mn,k) = p[mn*ld + k] ; MC: elem(mn,k) = p[k*ld + mn]

constexpr int kPad = 4;  // row pad (floats): keeps ds_write_b128 aligned, transposing writes <= 2-way

// ---- global -> register -> LDS staging of one operand tile [BMN x BK] ----------------------
template <int BMN, int BK, int LAYOUT, int VEC>
struct Stager;

template <int BMN, int BK>
struct Stager<BMN, BK, KC, 4> {
    static constexpr int NV = BMN * BK / 4 / kBlock;
    static_assert(NV >= 1, "tile too small");
    float4 v[NV];
    __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int mn0, int k0, int MN, int K, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int m = f / (BK / 4), kq = f % (BK / 4);
            bool ok = (mn0 + m < MN) && (k0 + 4 * kq < K);
            v[r] = ok ? *reinterpret_cast<const float4*>(p + (long)(mn0 + m) * ld + k0 + 4 * kq) : make_float4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int m = f / (BK / 4), kq = f % (BK / 4);
            float* q = lds + (4 * kq) * (BMN + kPad) + m;
            q[0] = v[r].x;
            q[BMN + kPad] = v[r].y;
            q[2 * (BMN + kPad)] = v[r].z;
            q[3 * (BMN + kPad)] = v[r].w;
        }
    }
    // side product: sum over k of the staged operand, per mn (rowsum of A / column sums for db)
    typedef float4 Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) { s[r].x += v[r].x; s[r].y += v[r].y; s[r].z += v[r].z; s[r].w += v[r].w; }
    }
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], float* /*scratch*/, float* out, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            float t = (s[r].x + s[r].y) + (s[r].z + s[r].w);
#pragma unroll
            for (int o = 1; o < BK / 4; o <<= 1) t += __shfl_xor(t, o, 64);
            int f = tid + kBlock * r;
            if (f % (BK / 4) == 0) out[f / (BK / 4)] = t;
        }
    }
};

template <int BMN, int BK>
struct Stager<BMN, BK, KC, 1> {
    static constexpr int NV = BMN * BK / kBlock;
    float v[NV];
    __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int mn0, int k0, int MN, int K, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int m = f / BK, k = f % BK;
            bool ok = (mn0 + m < MN) && (k0 + k < K);
            v[r] = ok ? p[(long)(mn0 + m) * ld + k0 + k] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int m = f / BK, k = f % BK;
            lds[k * (BMN + kPad) + m] = v[r];
        }
    }
    typedef float Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) s[r] += v[r];
    }
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], float* /*scratch*/, float* out, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            float t = s[r];
#pragma unroll
            for (int o = 1; o < BK; o <<= 1) t += __shfl_xor(t, o, 64);
            int f = tid + kBlock * r;
            if (f % BK == 0) out[f / BK] = t;
        }
    }
};

template <int BMN, int BK>
struct Stager<BMN, BK, MC, 4> {
    static constexpr int NV = BMN * BK / 4 / kBlock;
    static_assert(NV >= 1, "tile too small");
    float4 v[NV];
    __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int mn0, int k0, int MN, int K, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int k = f / (BMN / 4), mq = f % (BMN / 4);
            bool ok = (k0 + k < K) && (mn0 + 4 * mq < MN);  // MN % 4 == 0 on this path
            v[r] = ok ? *reinterpret_cast<const float4*>(p + (long)(k0 + k) * ld + mn0 + 4 * mq) : make_float4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int k = f / (BMN / 4), mq = f % (BMN / 4);
            *reinterpret_cast<float4*>(lds + k * (BMN + kPad) + 4 * mq) = v[r];
        }
    }
    typedef float4 Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) { s[r].x += v[r].x; s[r].y += v[r].y; s[r].z += v[r].z; s[r].w += v[r].w; }
    }
    // every staged float4 of a thread has the same mq (kBlock % (BMN/4) == 0): fold r, then across threads via LDS
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], float* scratch, float* out, int tid) {
        static_assert(kBlock % (BMN / 4) == 0, "mq must be fixed per thread");
        constexpr int ROWS = kBlock / (BMN / 4);
        float4 t = s[0];
#pragma unroll
        for (int r = 1; r < NV; ++r) { t.x += s[r].x; t.y += s[r].y; t.z += s[r].z; t.w += s[r].w; }
        *reinterpret_cast<float4*>(scratch + (tid / (BMN / 4)) * BMN + 4 * (tid % (BMN / 4))) = t;
        __syncthreads();
        if (tid < BMN) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < ROWS; ++q) a += scratch[q * BMN + tid];
            out[tid] = a;
        }
    }
};

template <int BMN, int BK>
struct Stager<BMN, BK, MC, 1> {
    static constexpr int NV = BMN * BK / kBlock;
    float v[NV];
    __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int mn0, int k0, int MN, int K, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int k = f / BMN, m = f % BMN;
            bool ok = (k0 + k < K) && (mn0 + m < MN);
            v[r] = ok ? p[(long)(k0 + k) * ld + mn0 + m] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int k = f / BMN, m = f % BMN;
            lds[k * (BMN + kPad) + m] = v[r];
        }
    }
    typedef float Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) s[r] += v[r];
    }
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], float* scratch, float* out, int tid) {
        static_assert(kBlock % BMN == 0, "m must be fixed per thread");
        constexpr int ROWS = kBlock / BMN;
        float t = s[0];
#pragma unroll
        for (int r = 1; r < NV; ++r) t += s[r];
        scratch[(tid / BMN) * BMN + (tid % BMN)] = t;
        __syncthreads();
        if (tid < BMN) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < ROWS; ++q) a += scratch[q * BMN + tid];
            out[tid] = a;
        }
    }
};

// ---- the tile main loop --------------------------------------------------------------------
template <int BM, int BN, int BK, int LA, int LB, int VEC>
struct GemmTile {
    static constexpr int SA = BM + kPad, SB = BN + kPad;
    static constexpr int WM = BM / 64, WN = BN / 64;
    static constexpr int kLdsFloats = 2 * BK * (SA + SB);
    static_assert(BM % 64 == 0 && BN % 64 == 0 && BK % 4 == 0, "tile shape");

    // acc[wm][wn] (+)= A[m0.., :] . B[n0.., :]^T over k in [0,K).  Rows >= M / cols >= N / k >= K read as 0.
    // If SIDE_A: side_lds[BM] receives sum_k A[m0+m, k] (valid after the call's final barrier); the operand
    // LDS buffers are reused as scratch for that reduction.
    template <bool SIDE_A>
    __device__ static __forceinline__ void run(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                               int m0, int n0, int M, int N, int K, float* lds, float* side_lds,
                                               f32x16 (&acc)[WM][WN]) {
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
        const int fi = lane & 31, fh = lane >> 5;
        float* sA[2] = {lds, lds + BK * SA};
        float* sB[2] = {lds + 2 * BK * SA, lds + 2 * BK * SA + BK * SB};

        Stager<BM, BK, LA, VEC> ga;
        Stager<BN, BK, LB, VEC> gb;
        using SG = Stager<BM, BK, LA, VEC>;
        typename SG::Side side[SG::NV];
        if constexpr (SIDE_A) {
#pragma unroll
            for (int r = 0; r < SG::NV; ++r) side[r] = typename SG::Side{};
        }

        const int nk = (K + BK - 1) / BK;
        ga.load(A, lda, m0, 0, M, K, tid);
        gb.load(B, ldb, n0, 0, N, K, tid);
        ga.store(sA[0], tid);
        gb.store(sB[0], tid);
        if constexpr (SIDE_A) ga.side_add(side);
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nk;
            if (more) {  // issue next tile's global loads before this tile's MFMAs (latency hides under them)
                ga.load(A, lda, m0, (kt + 1) * BK, M, K, tid);
                gb.load(B, ldb, n0, (kt + 1) * BK, N, K, tid);
            }
            const float* a_base = sA[cur] + fh * SA + wm0 + fi;
            const float* b_base = sB[cur] + fh * SB + wn0 + fi;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float a[WM], b[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) a[i] = a_base[kk * SA + i * 32];
#pragma unroll
                for (int j = 0; j < WN; ++j) b[j] = b_base[kk * SB + j * 32];
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (more) {
                ga.store(sA[cur ^ 1], tid);
                gb.store(sB[cur ^ 1], tid);
                if constexpr (SIDE_A) ga.side_add(side);
            }
            __syncthreads();
        }
        if constexpr (SIDE_A) {
            SG::side_reduce(side, lds, side_lds, tid);
            __syncthreads();
        }
    }

    // C/D lane map of the 32x32 MFMA: reg r of lane l is (row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31).
    __device__ static __forceinline__ int sub_row(int wm, int r) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave >> 1) * (BM / 2) + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int sub_col(int wn) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave & 1) * (BN / 2) + wn * 32 + (lane & 31);
    }
};

template <int WM, int WN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[WM][WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

}  // namespace vgan
