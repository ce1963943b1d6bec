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

// LDS pointers carry their address space explicitly: through a generic `float*` (e.g. a runtime-selected
// double-buffer pointer) hipcc emits flat_load/flat_store, which are slower than ds_read/ds_write AND count
// on vmcnt, so every fragment read would also wait for the global prefetch in flight.
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) f32x4 lds_f4;  // ext-vector: plain assignment works in any address space

// ---- global -> register -> LDS staging of one operand tile [BMN x BK] ----------------------
template <int BMN, int BK, int LAYOUT, int VEC>
struct Stager;

template <int BMN, int BK>
struct Stager<BMN, BK, KC, 4> {
    static constexpr int NV = BMN * BK / 4 / kBlock;
    static_assert(NV >= 1, "tile too small");
    float4 v[NV];
    bool ok[NV];
    // Loads are UNCONDITIONAL from a clamped (always valid) address and the out-of-range zeroing happens in
    // store(): a guarded load compiles to a branch whose merge forces s_waitcnt vmcnt(0) right after issue,
    // which would serialise the prefetch with the MFMAs it is meant to hide under.
    __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int mn0, int k0, int MN, int K, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int m = mn0 + f / (BK / 4), k = k0 + 4 * (f % (BK / 4));
            ok[r] = (m < MN) && (k < K);
            v[r] = *reinterpret_cast<const float4*>(p + (long)min(m, MN - 1) * ld + min(k, K - 4));
        }
    }
    __device__ __forceinline__ void store(lds_f* lds, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int m = f / (BK / 4), kq = f % (BK / 4);
            if (!ok[r]) v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            lds_f* q = lds + (4 * kq) * (BMN + kPad) + m;
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
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], lds_f* /*scratch*/, lds_f* out, int tid) {
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
    bool ok[NV];
    __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int mn0, int k0, int MN, int K, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int m = mn0 + f / BK, k = k0 + f % BK;
            ok[r] = (m < MN) && (k < K);
            v[r] = p[(long)min(m, MN - 1) * ld + min(k, K - 1)];
        }
    }
    __device__ __forceinline__ void store(lds_f* lds, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int m = f / BK, k = f % BK;
            if (!ok[r]) v[r] = 0.f;
            lds[k * (BMN + kPad) + m] = v[r];
        }
    }
    typedef float Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) s[r] += v[r];
    }
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], lds_f* /*scratch*/, lds_f* out, int tid) {
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
    bool ok[NV];
    __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int mn0, int k0, int MN, int K, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int k = k0 + f / (BMN / 4), m = mn0 + 4 * (f % (BMN / 4));
            ok[r] = (k < K) && (m < MN);  // MN % 4 == 0 on this path
            v[r] = *reinterpret_cast<const float4*>(p + (long)min(k, K - 1) * ld + min(m, MN - 4));
        }
    }
    __device__ __forceinline__ void store(lds_f* lds, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int k = f / (BMN / 4), mq = f % (BMN / 4);
            if (!ok[r]) v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            *(lds_f4*)(lds + k * (BMN + kPad) + 4 * mq) = f32x4{v[r].x, v[r].y, v[r].z, v[r].w};
        }
    }
    typedef float4 Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) { s[r].x += v[r].x; s[r].y += v[r].y; s[r].z += v[r].z; s[r].w += v[r].w; }
    }
    // every staged float4 of a thread has the same mq (kBlock % (BMN/4) == 0): fold r, then across threads via LDS
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], lds_f* scratch, lds_f* out, int tid) {
        static_assert(kBlock % (BMN / 4) == 0, "mq must be fixed per thread");
        constexpr int ROWS = kBlock / (BMN / 4);
        float4 t = s[0];
#pragma unroll
        for (int r = 1; r < NV; ++r) { t.x += s[r].x; t.y += s[r].y; t.z += s[r].z; t.w += s[r].w; }
        *(lds_f4*)(scratch + (tid / (BMN / 4)) * BMN + 4 * (tid % (BMN / 4))) = f32x4{t.x, t.y, t.z, t.w};
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
    bool ok[NV];
    __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int mn0, int k0, int MN, int K, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int k = k0 + f / BMN, m = mn0 + f % BMN;
            ok[r] = (k < K) && (m < MN);
            v[r] = p[(long)min(k, K - 1) * ld + min(m, MN - 1)];
        }
    }
    __device__ __forceinline__ void store(lds_f* lds, int tid) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int f = tid + kBlock * r;
            int k = f / BMN, m = f % BMN;
            if (!ok[r]) v[r] = 0.f;
            lds[k * (BMN + kPad) + m] = v[r];
        }
    }
    typedef float Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) s[r] += v[r];
    }
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], lds_f* scratch, lds_f* out, int tid) {
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
                                               int m0, int n0, int M, int N, int K, float* lds_generic, float* side_generic,
                                               f32x16 (&acc)[WM][WN]) {
        lds_f* lds = (lds_f*)lds_generic;
        lds_f* side_lds = (lds_f*)side_generic;
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
        const int fi = lane & 31, fh = lane >> 5;
        lds_f* const sA0 = lds;                // buffers: A0 | A1 | B0 | B1
        lds_f* const sB0 = lds + 2 * BK * SA;

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
        ga.store(sA0, tid);  // store() also zeroes out-of-range values, so side_add after it sees masked data
        gb.store(sB0, tid);
        if constexpr (SIDE_A) ga.side_add(side);
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nk;
            if (more) {  // issue next tile's global loads before this tile's MFMAs (latency hides under them)
                ga.load(A, lda, m0, (kt + 1) * BK, M, K, tid);
                gb.load(B, ldb, n0, (kt + 1) * BK, N, K, tid);
            }
            const lds_f* a_base = sA0 + cur * (BK * SA) + fh * SA + wm0 + fi;
            const lds_f* b_base = sB0 + cur * (BK * SB) + fh * SB + wn0 + fi;
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
                ga.store(sA0 + (cur ^ 1) * (BK * SA), tid);
                gb.store(sB0 + (cur ^ 1) * (BK * SB), tid);
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
