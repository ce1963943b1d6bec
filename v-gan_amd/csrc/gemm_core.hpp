// fp32-MFMA tile engine shared by every dense contraction on the V-GAN hot path
// (Linear fwd/bwd, the Gram tile of the MMD, and the W.Z product of its backward).
//
// One workgroup = 256 threads = 4 waves laid out 2 (M) x 2 (N); each wave owns WM x WN
// sub-tiles of 32x32 computed with v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 64 cycles per
// issue per SIMD -- the fp32 matrix rate equals the fp32 vector peak on gfx950, so the MFMA
// pipe, not LDS or L2, is the bound).  Staging is register-staged and double-buffered in LDS with
// a three-stage pipeline (global loads two K tiles ahead, LDS fill one tile ahead), one barrier
// per K tile.
//
// LDS images never transpose on the way in (a transposing scalar store is a 4-way bank conflict that makes
// the LDS, not the MFMA, the bound):
//   KC operand (k contiguous in memory) -> image [mn][BK+4], filled by ds_write_b128, read by ds_read_b128;
//   MC operand (mn contiguous in memory) -> image [BK][mn+4], filled by ds_write_b128, read by ds_read_b32.
// The 32x32x2 MFMA takes lane l -> (mn = l & 31, k-slot = l >> 5).  Which k a slot carries is free as long as
// A and B agree, so within each group of 8 k the step s (0..3) of half h = l >> 5 uses k = 8c + 4h + s: a KC
// lane then needs exactly the 4 consecutive floats [8c + 4h, +4) of its row -- one 16-byte read per 4 MFMAs.
// Row pads of 4 floats make every one of these accesses conflict-free (row stride = 4 * odd).
#pragma once
#include <type_traits>

#include "vgan_common.hpp"

namespace vgan {

enum : int { KC = 0, MC = 1 };  // KC: elem(mn,k) = p[mn*ld + k] ; MC: elem(mn,k) = p[k*ld + mn]

constexpr int kPad = 4;  // row pad (floats)

// LDS pointers carry their address space explicitly: through a generic `float*` (e.g. a runtime-selected
// double-buffer pointer) hipcc emits flat_load/flat_store, which are slower than ds_read/ds_write AND count
// on vmcnt, so every fragment read would also wait for the global prefetch in flight.
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) f32x4 lds_f4;  // ext-vector: plain assignment works in any address space

// ---- global -> register -> LDS staging of one operand tile [BMN x BK] ----------------------
// init() hoists everything that does not depend on the K tile (row pointers, row validity, LDS offsets);
// load(k0) issues UNCONDITIONAL loads from clamped, always-valid addresses and store() zeroes what was out
// of range: a guarded load compiles to a branch whose merge forces s_waitcnt vmcnt(0) right after issue,
// which would serialise the prefetch with the MFMAs it is meant to hide under.
// SL: the operand is given as `nslabs` partial slabs `slab_stride` elements apart (split-K output of a previous
// product); load() sums them in ascending order, which removes a separate reduction launch.
template <int BMN, int BK, int LAYOUT, int VEC, bool SL = false, int NTH = kBlock>
struct Stager;

template <int BMN, int BK, bool SL, int NTH>
struct Stager<BMN, BK, KC, 4, SL, NTH> {
    int nslabs = 1;
    long slab_stride = 0;
    static constexpr int NV = BMN * BK / 4 / NTH;
    static_assert(NV >= 1, "tile too small");
    float4 v[NV];
    const float* rowp[NV];
    int lofs[NV];
    bool rowok[NV], ok[NV];
    int kq4, K;
    __device__ __forceinline__ void init(const float* __restrict__ p, long ld, int mn0, int MN, int K_, int tid) {
        K = K_;
        kq4 = 4 * (tid % (BK / 4));  // NTH % (BK/4) == 0: the same k offset for every r
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int m = (tid + NTH * r) / (BK / 4);
            rowok[r] = mn0 + m < MN;
            rowp[r] = p + (long)min(mn0 + m, MN - 1) * ld;
            lofs[r] = m * (BK + kPad) + kq4;
        }
    }
    __device__ __forceinline__ void load(int k0) {
        const int k = k0 + kq4;
        const int kc = min(k, K - 4);
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            ok[r] = rowok[r] && (k < K);
            v[r] = *reinterpret_cast<const float4*>(rowp[r] + kc);
            if constexpr (SL)
                for (int sl = 1; sl < nslabs; ++sl) {
                    const float4 t = *reinterpret_cast<const float4*>(rowp[r] + sl * slab_stride + kc);
                    v[r].x += t.x; v[r].y += t.y; v[r].z += t.z; v[r].w += t.w;
                }
        }
    }
    __device__ __forceinline__ void store(lds_f* lds) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            if (!ok[r]) v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            *(lds_f4*)(lds + lofs[r]) = f32x4{v[r].x, v[r].y, v[r].z, v[r].w};
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
            int f = tid + NTH * r;
            if (f % (BK / 4) == 0) out[f / (BK / 4)] = t;
        }
    }
};

template <int BMN, int BK, bool SL, int NTH>
struct Stager<BMN, BK, KC, 1, SL, NTH> {
    int nslabs = 1;
    long slab_stride = 0;
    static constexpr int NV = BMN * BK / NTH;
    float v[NV];
    const float* rowp[NV];
    int lofs[NV];
    bool rowok[NV], ok[NV];
    int kk, K;
    __device__ __forceinline__ void init(const float* __restrict__ p, long ld, int mn0, int MN, int K_, int tid) {
        K = K_;
        kk = tid % BK;  // NTH % BK == 0
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int m = (tid + NTH * r) / BK;
            rowok[r] = mn0 + m < MN;
            rowp[r] = p + (long)min(mn0 + m, MN - 1) * ld;
            lofs[r] = m * (BK + kPad) + kk;
        }
    }
    __device__ __forceinline__ void load(int k0) {
        const int k = k0 + kk;
        const int kc = min(k, K - 1);
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            ok[r] = rowok[r] && (k < K);
            v[r] = rowp[r][kc];
            if constexpr (SL)
                for (int sl = 1; sl < nslabs; ++sl) v[r] += rowp[r][sl * slab_stride + kc];
        }
    }
    __device__ __forceinline__ void store(lds_f* lds) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            if (!ok[r]) v[r] = 0.f;
            lds[lofs[r]] = v[r];
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
            int f = tid + NTH * r;
            if (f % BK == 0) out[f / BK] = t;
        }
    }
};

template <int BMN, int BK, bool SL, int NTH>
struct Stager<BMN, BK, MC, 4, SL, NTH> {
    int nslabs = 1;
    long slab_stride = 0;
    static constexpr int NV = BMN * BK / 4 / NTH;
    static_assert(NV >= 1, "tile too small");
    static_assert(NTH % (BMN / 4) == 0, "mq must be fixed per thread");
    float4 v[NV];
    const float* colp;  // p + clamped column of this thread
    long ld;
    int krow[NV], lofs[NV];
    bool colok, ok[NV];
    int K;
    __device__ __forceinline__ void init(const float* __restrict__ p, long ld_, int mn0, int MN, int K_, int tid) {
        K = K_;
        ld = ld_;
        const int m = mn0 + 4 * (tid % (BMN / 4));  // MN % 4 == 0 on this path
        colok = m < MN;
        colp = p + min(m, MN - 4);
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            krow[r] = (tid + NTH * r) / (BMN / 4);
            lofs[r] = krow[r] * (BMN + kPad) + 4 * (tid % (BMN / 4));
        }
    }
    __device__ __forceinline__ void load(int k0) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int k = k0 + krow[r];
            ok[r] = colok && (k < K);
            v[r] = *reinterpret_cast<const float4*>(colp + (long)min(k, K - 1) * ld);
            if constexpr (SL)
                for (int sl = 1; sl < nslabs; ++sl) {
                    const float4 t = *reinterpret_cast<const float4*>(colp + sl * slab_stride + (long)min(k, K - 1) * ld);
                    v[r].x += t.x; v[r].y += t.y; v[r].z += t.z; v[r].w += t.w;
                }
        }
    }
    __device__ __forceinline__ void store(lds_f* lds) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            if (!ok[r]) v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            *(lds_f4*)(lds + lofs[r]) = f32x4{v[r].x, v[r].y, v[r].z, v[r].w};
        }
    }
    typedef float4 Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) { s[r].x += v[r].x; s[r].y += v[r].y; s[r].z += v[r].z; s[r].w += v[r].w; }
    }
    // every staged float4 of a thread has the same mq: fold r, then across threads via LDS
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], lds_f* scratch, lds_f* out, int tid) {
        constexpr int ROWS = NTH / (BMN / 4);
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

template <int BMN, int BK, bool SL, int NTH>
struct Stager<BMN, BK, MC, 1, SL, NTH> {
    int nslabs = 1;
    long slab_stride = 0;
    static constexpr int NV = BMN * BK / NTH;
    static_assert(NTH % BMN == 0, "m must be fixed per thread");
    float v[NV];
    const float* colp;
    long ld;
    int krow[NV], lofs[NV];
    bool colok, ok[NV];
    int K;
    __device__ __forceinline__ void init(const float* __restrict__ p, long ld_, int mn0, int MN, int K_, int tid) {
        K = K_;
        ld = ld_;
        const int m = mn0 + tid % BMN;
        colok = m < MN;
        colp = p + min(m, MN - 1);
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            krow[r] = (tid + NTH * r) / BMN;
            lofs[r] = krow[r] * (BMN + kPad) + tid % BMN;
        }
    }
    __device__ __forceinline__ void load(int k0) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int k = k0 + krow[r];
            ok[r] = colok && (k < K);
            v[r] = colp[(long)min(k, K - 1) * ld];
            if constexpr (SL)
                for (int sl = 1; sl < nslabs; ++sl) v[r] += colp[sl * slab_stride + (long)min(k, K - 1) * ld];
        }
    }
    __device__ __forceinline__ void store(lds_f* lds) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            if (!ok[r]) v[r] = 0.f;
            lds[lofs[r]] = v[r];
        }
    }
    typedef float Side;
    __device__ __forceinline__ void side_add(Side (&s)[NV]) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) s[r] += v[r];
    }
    __device__ static __forceinline__ void side_reduce(const Side (&s)[NV], lds_f* scratch, lds_f* out, int tid) {
        constexpr int ROWS = NTH / BMN;
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
// SLABS bit 0: the A operand arrives as slabs, bit 1: the B operand does (see Stager).
// KW = 2: 512-thread workgroups; waves 4..7 mirror waves 0..3 on the second half of every K tile (k-groups split between
// the two wave sets) and the partial accumulators are exchanged through LDS at the end, each wave keeping 8 of its 16
// result registers (first_reg()/num_regs()).  Same MFMA count per CU, but four waves per SIMD from two resident
// workgroups instead of two: the staging/barrier phases of one wave hide under the MFMAs of the others.
template <int BM, int BN, int BK, int LA, int LB, int VEC, int SLABS = 0, int KW = 1>
struct GemmTile {
    static constexpr int NTH = kBlock * KW;
    static constexpr int WM = BM / 64, WN = BN / 64;
    static_assert(KW == 1 || (KW == 2 && BM == 64 && BN == 64 && (BK / 8) % 2 == 0), "K-split needs one sub-tile per wave");
    static constexpr int kImgA = (LA == KC) ? BM * (BK + kPad) : BK * (BM + kPad);  // floats per buffer
    static constexpr int kImgB = (LB == KC) ? BN * (BK + kPad) : BK * (BN + kPad);
    static constexpr int kLdsFloats = 2 * (kImgA + kImgB);
    static_assert(BM % 64 == 0 && BN % 64 == 0 && BK % 8 == 0, "tile shape");

    // the 4 fragment values (steps s = 0..3) of k-group c for the 32 rows starting at `mn` of an operand image
    template <int LAYOUT, int BMN>
    __device__ static __forceinline__ f32x4 frag(const lds_f* img, int mn, int c, int fh) {
        if constexpr (LAYOUT == KC) {
            return *(const lds_f4*)(img + mn * (BK + kPad) + 8 * c + 4 * fh);
        } else {
            const lds_f* q = img + (8 * c + 4 * fh) * (BMN + kPad) + mn;
            return f32x4{q[0], q[BMN + kPad], q[2 * (BMN + kPad)], q[3 * (BMN + kPad)]};
        }
    }

    // acc[wm][wn] (+)= A[m0.., :] . B[n0.., :]^T over k in [0,K), K >= 1.  Rows >= M / cols >= N / k >= K read as 0.
    // If SIDE_A: side_lds[BM] receives sum_k A[m0+m, k] (valid after the call's final barrier); the operand
    // LDS buffers are reused as scratch for that reduction.
    template <bool SIDE_A>
    __device__ static __forceinline__ void run(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                               int m0, int n0, int M, int N, int K, float* lds_generic, float* side_generic,
                                               f32x16 (&acc)[WM][WN], int nslabs = 1, long slab_stride = 0) {
        lds_f* lds = (lds_f*)lds_generic;
        lds_f* side_lds = (lds_f*)side_generic;
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = (tid >> 6) & 3, kp = tid >> 8;  // kp: which share of each K tile (KW = 2)
        const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
        const int fi = lane & 31, fh = lane >> 5;
        lds_f* const sA0 = lds;  // buffers: A0 | A1 | B0 | B1
        lds_f* const sB0 = lds + 2 * kImgA;
        constexpr int GC = BK / 8 / KW;  // k-groups of 8 per wave per K tile

        using SG = Stager<BM, BK, LA, VEC, (SLABS & 1) != 0, NTH>;
        SG ga;
        Stager<BN, BK, LB, VEC, (SLABS & 2) != 0, NTH> gb;
        ga.init(A, lda, m0, M, K, tid);
        gb.init(B, ldb, n0, N, K, tid);
        if constexpr (SLABS & 1) { ga.nslabs = nslabs; ga.slab_stride = slab_stride; }
        if constexpr (SLABS & 2) { gb.nslabs = nslabs; gb.slab_stride = slab_stride; }
        typename SG::Side side[SG::NV];
        if constexpr (SIDE_A) {
#pragma unroll
            for (int r = 0; r < SG::NV; ++r) side[r] = typename SG::Side{};
        }

        const int nk = (K + BK - 1) / BK;
        // Three-stage pipeline over K tiles: while the MFMAs of tile t run out of LDS buffer t&1, tile t+1 (loaded
        // from global one whole iteration earlier, so its vmcnt wait is short) is written into buffer (t+1)&1 and
        // the global loads of tile t+2 are issued.  One barrier per tile: it orders both the reads of buffer t&1
        // against its refill in iteration t+1 and the writes of buffer (t+1)&1 against their reads.
        ga.load(0);
        gb.load(0);
        ga.store(sA0);  // store() also zeroes out-of-range values, so side_add after it sees masked data
        gb.store(sB0);
        if constexpr (SIDE_A) ga.side_add(side);
        if (nk > 1) {
            ga.load(BK);
            gb.load(BK);
        }
        __syncthreads();

        // The steady-state body is branch-free (the last two tiles are peeled) so that it is ONE scheduling region:
        // a wave issues in order, so the staging stores, the next global loads and their address arithmetic should
        // sit BETWEEN the MFMAs (64 cycles of pipe time each) rather than before/after a solid MFMA block.
        auto body = [&](int kt, auto store_next, auto load_next2) {
            const int cur = kt & 1;
            const lds_f* imgA = sA0 + cur * kImgA;
            const lds_f* imgB = sB0 + cur * kImgB;
            f32x4 a[GC][WM], b[GC][WN];
#pragma unroll
            for (int c = 0; c < GC; ++c) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[c][i] = frag<LA, BM>(imgA, wm0 + i * 32 + fi, kp * GC + c, fh);
#pragma unroll
                for (int j = 0; j < WN; ++j) b[c][j] = frag<LB, BN>(imgB, wn0 + j * 32 + fi, kp * GC + c, fh);
            }
#ifndef VGAN_ABLATE_NO_LDS_STORE
            if constexpr (decltype(store_next)::value) {
                ga.store(sA0 + (cur ^ 1) * kImgA);
                gb.store(sB0 + (cur ^ 1) * kImgB);
                if constexpr (SIDE_A) ga.side_add(side);
            }
#endif
#ifndef VGAN_ABLATE_NO_GLOBAL
            if constexpr (decltype(load_next2)::value) {
                ga.load((kt + 2) * BK);
                gb.load((kt + 2) * BK);
            }
#endif
#pragma unroll
            for (int c = 0; c < GC; ++c)
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
#ifndef VGAN_ABLATE_NO_MFMA
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i][st], b[c][j][st], acc[i][j], 0, 0, 0);
#else
                            acc[i][j][st] += a[c][i][st] * b[c][j][st];
#endif
#if !defined(VGAN_SCHED_NONE)
            // let the scheduler interleave the LDS traffic with the MFMA chain inside this single-block body
            // (measured on MI355X, 64x64x32 tiles: none 74.6 / hand-written group barriers 79.6 / iglp_opt(0) 87.5 TFLOP/s
            // at one wave per SIMD)
            __builtin_amdgcn_iglp_opt(0);
#endif
#ifndef VGAN_ABLATE_NO_BARRIER
            __syncthreads();
#endif
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
            SG::side_reduce(side, lds, side_lds, tid);
            __syncthreads();
        }
        if constexpr (KW == 2) {
            // exchange halves: wave set kp keeps registers [8 kp, 8 kp + 8) and receives the other set's share of them
            lds_f* xch = lds;  // [set][sub-tile][8 regs][64 lanes], staging buffers are free after the loop's last barrier
#pragma unroll
            for (int r = 0; r < 8; ++r) xch[((kp * 4 + wave) * 8 + r) * 64 + lane] = acc[0][0][8 * (1 - kp) + r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[0][0][8 * kp + r] += xch[(((1 - kp) * 4 + wave) * 8 + r) * 64 + lane];
            __syncthreads();  // the epilogue may reuse lds
        }
    }
    // result registers this wave owns after run(): all 16 (KW = 1) or 8 of them (KW = 2)
    __device__ static __forceinline__ int first_reg() { return KW == 2 ? 8 * (threadIdx.x >> 8) : 0; }
    static constexpr int kNumRegs = 16 / KW;

    // C/D lane map of the 32x32 MFMA: reg r of lane l is (row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31).
    __device__ static __forceinline__ int sub_row(int wm, int r) {
        const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
        return (wave >> 1) * (BM / 2) + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int sub_col(int wn) {
        const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
        return (wave & 1) * (BN / 2) + wn * 32 + (lane & 31);
    }
};


// ---- tall-skinny / tiny products: 32x32 tile per workgroup, the four waves split every K tile ---------------
// A chain of small products (the collapsed generator: outputs of at most 788 x 52, contractions of 100..1024) is
// bound by the LENGTH of the K loop of its few tiles, not by throughput.  Here the output tile is 32x32, so there are
// 4-8x more workgroups than with 64x64 tiles, and wave w of a workgroup takes the k-groups [w*BK/32, (w+1)*BK/32) of
// each BK-deep K tile (BK = 128: 16 MFMAs per wave per K tile, as in the big engine), so the loop is BK/32 = 4x shorter.
// The four partial accumulators meet in LDS once at the end; wave w then owns result registers 4w..4w+3.
// NW waves per workgroup (4 or 16): the K loop of these products is bound by the 64-cycle fp32 MFMAs each wave issues
// back to back (BK = 128, NW = 4: 16 per wave and K tile = 0.43 us), so for long contractions 16 waves (one k-group each)
// shorten it (measured: M_4 product 10.0 -> 8.9 us, the K = 788 group 10.0 -> 8.2 us; keeping four K tiles of loads in
// flight on top of that was slower again, 9.9 / 9.6 us, and so was a 256-deep K tile for the 16-wave variant, 9.2 / 7.2 us);
// wave w then owns result registers [w*16/NW, (w+1)*16/NW).
template <int BK, int LA, int LB, int VEC, int NW = 4>
struct GemmTileKS {
    static constexpr int T = 32;
    static constexpr int NTH = 64 * NW;
    static constexpr int NR = 16 / NW;  // result registers per wave
    static constexpr int kImgA = (LA == KC) ? T * (BK + kPad) : BK * (T + kPad);
    static constexpr int kImgB = (LB == KC) ? T * (BK + kPad) : BK * (T + kPad);
    static constexpr int kLdsFloats = 2 * (kImgA + kImgB);
    static constexpr int GW = BK / (8 * NW);  // k-groups (of 8) per wave per K tile
    static_assert(GW >= 1 && BK % (8 * NW) == 0 && kLdsFloats >= NW * 16 * 64, "K tile / reduction scratch");

    template <int LAYOUT>
    __device__ static __forceinline__ f32x4 frag(const lds_f* img, int mn, int c, int fh) {
        if constexpr (LAYOUT == KC) {
            return *(const lds_f4*)(img + mn * (BK + kPad) + 8 * c + 4 * fh);
        } else {
            const lds_f* q = img + (8 * c + 4 * fh) * (T + kPad) + mn;
            return f32x4{q[0], q[T + kPad], q[2 * (T + kPad)], q[3 * (T + kPad)]};
        }
    }

    // out[rr] (rr = 0..3) = full sum for result register 4*wave + rr of this lane (see row_of / col_of).
    __device__ static __forceinline__ void run(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb, int m0,
                                               int n0, int M, int N, int K, float* lds_generic, float (&out)[NR]) {
        lds_f* lds = (lds_f*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int fi = lane & 31, fh = lane >> 5;
        lds_f* const sA0 = lds;
        lds_f* const sB0 = lds + 2 * kImgA;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int nk = (K + BK - 1) / BK;
        Stager<T, BK, LA, VEC, false, NTH> ga;
        Stager<T, BK, LB, VEC, false, NTH> gb;
        ga.init(A, lda, m0, M, K, tid);
        gb.init(B, ldb, n0, N, K, tid);
        ga.load(0);
        gb.load(0);
        ga.store(sA0);
        gb.store(sB0);
        if (nk > 1) {
            ga.load(BK);
            gb.load(BK);
        }
        __syncthreads();
        auto body = [&](int kt, auto store_next, auto load_next2) {
            const int cur = kt & 1;
            const lds_f* imgA = sA0 + cur * kImgA;
            const lds_f* imgB = sB0 + cur * kImgB;
            f32x4 a[GW], b[GW];
#pragma unroll
            for (int c = 0; c < GW; ++c) {
                a[c] = frag<LA>(imgA, fi, wave * GW + c, fh);
                b[c] = frag<LB>(imgB, fi, wave * GW + c, fh);
            }
            if constexpr (decltype(store_next)::value) {
                ga.store(sA0 + (cur ^ 1) * kImgA);
                gb.store(sB0 + (cur ^ 1) * kImgB);
            }
            if constexpr (decltype(load_next2)::value) {
                ga.load((kt + 2) * BK);
                gb.load((kt + 2) * BK);
            }
#pragma unroll
            for (int c = 0; c < GW; ++c)
#pragma unroll
                for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][st], b[c][st], acc, 0, 0, 0);
            __builtin_amdgcn_iglp_opt(0);
            __syncthreads();
        };
        using Tt = std::true_type;
        using Ff = std::false_type;
        int kt = 0;
        for (; kt + 2 < nk; ++kt) body(kt, Tt{}, Tt{});
        if (kt + 1 < nk) {
            body(kt, Tt{}, Ff{});
            ++kt;
        }
        body(kt, Ff{}, Ff{});
        // cross-wave sum: red[w][r][lane]
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[(wave * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int r = NR * wave + rr;
            if constexpr (NW == 4) {
                out[rr] = (lds[(0 * 16 + r) * 64 + lane] + lds[(1 * 16 + r) * 64 + lane]) +
                          (lds[(2 * 16 + r) * 64 + lane] + lds[(3 * 16 + r) * 64 + lane]);
            } else {  // fixed order: quads of waves, then the quads
                float q[NW / 4];
#pragma unroll
                for (int w4 = 0; w4 < NW / 4; ++w4)
                    q[w4] = (lds[((4 * w4 + 0) * 16 + r) * 64 + lane] + lds[((4 * w4 + 1) * 16 + r) * 64 + lane]) +
                            (lds[((4 * w4 + 2) * 16 + r) * 64 + lane] + lds[((4 * w4 + 3) * 16 + r) * 64 + lane]);
                float t = q[0];
#pragma unroll
                for (int w4 = 1; w4 < NW / 4; ++w4) t += q[w4];
                out[rr] = t;
            }
        }
    }
    __device__ static __forceinline__ int row_of(int rr) {  // row inside the 32x32 tile of out[rr]
        const int lane = threadIdx.x & 63, r = NR * (threadIdx.x >> 6) + rr;
        return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int col_of() { return threadIdx.x & 31; }
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
