// Split-bf16 tile main loop shared by the Gram and backward kernels of mmd_bf16.hip (and by tools/ablate_bf3.hip).
#pragma once
#include <type_traits>

#include "gemm_core.hpp"

namespace vgan {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // plain vector type: assignable in any address space
typedef __attribute__((address_space(3))) u32x4 lds_u4;
typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s4;

// 8 consecutive k of one column of a ROW-MAJOR [k][n] bf16 LDS image, for this lane's MFMA B fragment: two
// ds_read_b64_tr_b16 (gfx950).  Per group of 16 lanes the instruction reads a block of 4 rows x 16 columns and hands lane i
// column i of the 4 rows; lane 4q + p of the group supplies the address of row q, columns 4p .. 4p+3 (probed on MI355X:
// tools/trtest.hip).  `p0` is this lane's address for rows R .. R+3, `rows4` the byte distance to rows R+4 .. R+7.
// EXEC must be all ones (no divergence around the call).
__device__ __forceinline__ bf16x8 tr_read_k8(const char __attribute__((address_space(3)))* p0, int rows4) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p0 + rows4));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}


// ---- the split-bf16 tile main loop: 64x64 output, 256 threads (2x2 waves), K tile of 64 -----------------------
// BK = 64: 73,728 B of LDS -> two workgroups per CU (the default).  BK = 32: 40,960 B -> three per CU, so that all 528
// Gram tiles of the metric's configuration are resident at once -- measured SLOWER (28.3 vs 25.5 us; the CUs that are
// dealt three tiles are throughput-bound, not waiting for a second round), kept behind VGAN_BF3_BK=32 for measurement.
//
// Variants measured and dropped (profiles/README.md): K tile 32 with three workgroups per CU (slower, bank-conflicted), the
// four waves splitting K instead of the tile (half the fragment reads but longer MFMA chains and a combine: slower).
template <int BK_>
struct GemmBF3 {
    static constexpr int BK = BK_;                 // bf16 elements of K per tile
    static constexpr int NR = BK / 32;             // 16-byte pieces per thread and operand part
    static constexpr int QPR = BK / 8;             // 16-byte pieces per row
    static constexpr int KS = BK / 16;             // MFMA k steps per tile
    static constexpr int ROWB = (BK + 8) * 2;      // bytes per LDS row: +16 pad (stride = 36 or 20 dwords = 4 * odd)
    static constexpr int PART = 64 * ROWB;         // one operand part (64 rows)
    static constexpr int BUF = 4 * PART;           // Ah | Al | Bh | Bl
    static constexpr int kLdsBytes = 2 * BUF;      // double buffered

    struct Stage {
        u32x4 v[4][NR];  // [part][r]
        const char* src[4][NR];
        int lofs[NR];
        // amap / bmap (optional): operand row g is stored at image row map[g] (a batch gathered by index from a resident
        // image: one dependent lookup per staged row, at tile start only)
        __device__ __forceinline__ void init(const unsigned short* Ah, const unsigned short* Al, long lda, int m0, int M,
                                             const unsigned short* Bh, const unsigned short* Bl, long ldb, int n0, int N, int tid,
                                             const int* amap = nullptr, const int* bmap = nullptr) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int f = tid + kBlock * r, row = f / QPR, q = f % QPR;
                const int ga = min(m0 + row, M - 1), gb = min(n0 + row, N - 1);
                const long ra = (long)(amap ? amap[ga] : ga) * lda + 8 * q, rb = (long)(bmap ? bmap[gb] : gb) * ldb + 8 * q;
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
                for (int r = 0; r < NR; ++r) v[part][r] = *reinterpret_cast<const u32x4*>(src[part][r] + 2 * (long)k0);
        }
        __device__ __forceinline__ void store(char __attribute__((address_space(3)))* buf) const {
#pragma unroll
            for (int part = 0; part < 4; ++part)
#pragma unroll
                for (int r = 0; r < NR; ++r) *(lds_u4*)(buf + part * PART + lofs[r]) = v[part][r];
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
    // (thread ids are taken modulo 256: a 512-thread workgroup may run TWO tiles side by side, waves 0-3 and 4-7 each on
    //  their own LDS region -- both halves execute the same number of barriers, K being the same)
    template <bool SIDE_A>
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K,
                                               char* lds_generic, float* rs_generic, f32x16& acc, const int* amap = nullptr,
                                               const int* bmap = nullptr) {
        typedef char __attribute__((address_space(3))) lds_c;
        lds_c* lds = (lds_c*)lds_generic;
        lds_f* rs_lds = (lds_f*)rs_generic;
        const int tid = threadIdx.x & (kBlock - 1), lane = tid & 63, wave = tid >> 6;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        const int fi = lane & 31, fh = lane >> 5;
        Stage st;
        st.init(Ah, Al, lda, m0, M, Bh, Bl, ldb, n0, N, tid, amap, bmap);
        float rsum[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) rsum[r] = 0.f;
        auto side = [&]() {
#pragma unroll
            for (int r = 0; r < NR; ++r) rsum[r] += st.a_rowpart(r);
        };
        const int nk = K / BK;
        st.load(0);
        st.store(lds);
        if constexpr (SIDE_A) side();
        if (nk > 1) st.load(BK);
        __syncthreads();
        auto body = [&](int kt, auto store_next, auto load_next2) {
            const lds_c* buf = lds + (kt & 1) * BUF;
            const lds_c* pa = buf + (wm0 + fi) * ROWB + fh * 16;
            const lds_c* pb = buf + 2 * PART + (wn0 + fi) * ROWB + fh * 16;
            u32x4 ah[KS], al[KS], bh[KS], bl[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {  // k16 step s: this lane's 8 consecutive k = 16 s + 8 fh ..
                ah[s] = *(const lds_u4*)(pa + s * 32);
                al[s] = *(const lds_u4*)(pa + PART + s * 32);
                bh[s] = *(const lds_u4*)(pb + s * 32);
                bl[s] = *(const lds_u4*)(pb + PART + s * 32);
            }
            if constexpr (decltype(store_next)::value) {
#ifndef VGAN_ABLATE_NO_LDS_STORE
                st.store(lds + ((kt & 1) ^ 1) * BUF);
#endif
                if constexpr (SIDE_A) side();
            }
#ifndef VGAN_ABLATE_NO_GLOBAL
            if constexpr (decltype(load_next2)::value) st.load((kt + 2) * BK);
#endif
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[s]), xl = __builtin_bit_cast(bf16x8, al[s]);
                const bf16x8 yh = __builtin_bit_cast(bf16x8, bh[s]), yl = __builtin_bit_cast(bf16x8, bl[s]);
#ifdef VGAN_ABLATE_NO_MFMA
                acc[s] += __builtin_bit_cast(f32x4, ah[s] ^ al[s] ^ bh[s] ^ bl[s])[0];
#else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc, 0, 0, 0);  // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc, 0, 0, 0);
#ifndef VGAN_ABLATE_ONE_PRODUCT
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
#endif
#endif
            }
            __builtin_amdgcn_iglp_opt(0);
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
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float s = rsum[r];
#pragma unroll
                for (int o = 1; o < QPR; o <<= 1) s += __shfl_xor(s, o, 64);
                const int f = tid + kBlock * r;
                if (f % QPR == 0) rs_lds[f / QPR] = s;
            }
            __syncthreads();
        }
    }
    __device__ static __forceinline__ int sub_row(int r) {
        const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
        return (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int sub_col() {
        const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
        return (wave & 1) * 32 + (lane & 31);
    }

    // ---- the same product with B given ROW-MAJOR over the contraction index: acc (+)= A[m0.., k] . B[k, n0..] ------------
    // (the backward product W . Z reads Z's split images exactly as the Gram does: no transposed copy of Z exists any more.)
    // LDS image of a B part: [64 k][64 n] bf16, 128-byte rows, the 16-byte chunk c of row r at chunk c ^ (((r >> 1) & 1) << 2):
    // a 32-lane half of a transposed read touches 4 rows x 64 bytes, and the swizzle spreads the four rows over the four
    // 16-bank quarters (rows r and r+2 would otherwise share banks: pitch 32 dwords) -- conflict-free, no padding.
    static_assert(BK == 64, "row-major B path: K tile of 64");
    static constexpr int PARTB = 64 * 128;
    static constexpr int BUFT = 2 * PART + 2 * PARTB;   // Ah | Al | Bh | Bl
    static constexpr int kLdsBytesT = 2 * BUFT;

    struct StageT {
        u32x4 v[4][NR];
        const char* srcA[2][NR];
        const unsigned short *Bh, *Bl;
        long ldb;
        int lofsA[NR], lofsB[NR], rowB[NR], colB[NR], zrows;
        __device__ __forceinline__ void init(const unsigned short* Ah, const unsigned short* Al, long lda, int m0, int M,
                                             const unsigned short* Bh_, const unsigned short* Bl_, long ldb_, int n0, int zrows_, int tid) {
            Bh = Bh_; Bl = Bl_; ldb = ldb_; zrows = zrows_;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int f = tid + kBlock * r, row = f / QPR, q = f % QPR;
                const long ra = (long)min(m0 + row, M - 1) * lda + 8 * q;
                srcA[0][r] = reinterpret_cast<const char*>(Ah + ra);
                srcA[1][r] = reinterpret_cast<const char*>(Al + ra);
                lofsA[r] = row * ROWB + q * 16;
                rowB[r] = row;                       // k row of the tile
                colB[r] = n0 + 8 * q;                // first of 8 consecutive n
                lofsB[r] = row * 128 + ((q ^ (((row >> 1) & 1) << 2)) << 4);
            }
        }
        __device__ __forceinline__ void load(int k0) {  // rows past the image are clamped: their A columns are zero by contract
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                v[0][r] = *reinterpret_cast<const u32x4*>(srcA[0][r] + 2 * (long)k0);
                v[1][r] = *reinterpret_cast<const u32x4*>(srcA[1][r] + 2 * (long)k0);
                const long ob = (long)min(k0 + rowB[r], zrows - 1) * ldb + colB[r];
                v[2][r] = *reinterpret_cast<const u32x4*>(Bh + ob);
                v[3][r] = *reinterpret_cast<const u32x4*>(Bl + ob);
            }
        }
        __device__ __forceinline__ void store(char __attribute__((address_space(3)))* buf) const {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                *(lds_u4*)(buf + lofsA[r]) = v[0][r];
                *(lds_u4*)(buf + PART + lofsA[r]) = v[1][r];
                *(lds_u4*)(buf + 2 * PART + lofsB[r]) = v[2][r];
                *(lds_u4*)(buf + 2 * PART + PARTB + lofsB[r]) = v[3][r];
            }
        }
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

    // Bh/Bl: [zrows, ldb] row-major (zrows = rows that exist from the tile's first k on); K a multiple of 64.
    template <bool SIDE_A>
    __device__ static __forceinline__ void run_bt(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                                  const unsigned short* Bl, long ldb, int zrows, int m0, int n0, int M, int K,
                                                  char* lds_generic, float* rs_generic, f32x16& acc) {
        typedef char __attribute__((address_space(3))) lds_c;
        lds_c* lds = (lds_c*)lds_generic;
        lds_f* rs_lds = (lds_f*)rs_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        const int fi = lane & 31, fh = lane >> 5;
        // transposed-read geometry of this lane (see tr_read_k8): group g of 16 lanes, row q and column piece p inside it
        const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
        const int chunk = (wn0 >> 3) + 2 * (g & 1) + (pp >> 1);
        const int offB0 = (8 * (g >> 1) + qq) * 128 + ((chunk ^ (((qq >> 1) & 1) << 2)) << 4) + (pp & 1) * 8;
        StageT st;
        st.init(Ah, Al, lda, m0, M, Bh, Bl, ldb, n0, zrows, tid);
        float rsum[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) rsum[r] = 0.f;
        auto side = [&]() {
#pragma unroll
            for (int r = 0; r < NR; ++r) rsum[r] += st.a_rowpart(r);
        };
        const int nk = K / BK;
        st.load(0);
        st.store(lds);
        if constexpr (SIDE_A) side();
        if (nk > 1) st.load(BK);
        __syncthreads();
        auto body = [&](int kt, auto store_next, auto load_next2) {
            const lds_c* buf = lds + (kt & 1) * BUFT;
            const lds_c* pa = buf + (wm0 + fi) * ROWB + fh * 16;
            const lds_c* pb = buf + 2 * PART + offB0;
            u32x4 ah[KS], al[KS];
            bf16x8 yh[KS], yl[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ah[s] = *(const lds_u4*)(pa + s * 32);
                al[s] = *(const lds_u4*)(pa + PART + s * 32);
                yh[s] = tr_read_k8(pb + s * 2048, 512);
                yl[s] = tr_read_k8(pb + PARTB + s * 2048, 512);
            }
            if constexpr (decltype(store_next)::value) {
                st.store(lds + ((kt & 1) ^ 1) * BUFT);
                if constexpr (SIDE_A) side();
            }
            if constexpr (decltype(load_next2)::value) st.load((kt + 2) * BK);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[s]), xl = __builtin_bit_cast(bf16x8, al[s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh[s], acc, 0, 0, 0);  // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh[s], acc, 0, 0, 0);
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
            for (int r = 0; r < NR; ++r) {
                float s = rsum[r];
#pragma unroll
                for (int o = 1; o < QPR; o <<= 1) s += __shfl_xor(s, o, 64);
                const int f = tid + kBlock * r;
                if (f % QPR == 0) rs_lds[f / QPR] = s;
            }
            __syncthreads();
        }
    }
};


// ---- 128x128 output tile, 512 threads: 8 waves as 2 (rows) x 4 (columns), each wave two stacked 32x32 sub-tiles ----
// The 64x64 kernel moves (64 + 64) rows x 64 k x 2 images x 2 B = 32 KB from L2 into LDS per K tile and tile; at the
// metric's size that is 220 MB per launch, and the measured main loop time (10 us) is exactly that traffic at ~22 TB/s --
// the aggregate L2 -> CU rate.  Doubling the tile edge halves the bytes per flop (113 MB) and the LDS fill per flop; the
// 136 tiles of that size still fill more than half of the CUs, and the loop becomes MFMA-bound (128 x 128 x K x 6 flop per
// CU at 4 x 1017 flop/cycle: 8.4 us).  Chosen by the caller (tile argument of vgan_mmd_build_tiles / vgan_mmd_gram_bf3).
struct GemmBF3Big {
    static constexpr int BM = 128, BN = 128, BK = 64, NTH = 512;
    static constexpr int ROWB = (BK + 8) * 2;      // 144 B per LDS row (36 dwords = 4 * odd)
    static constexpr int PART = BM * ROWB;         // one operand part (128 rows)
    static constexpr int BUF = 4 * PART;           // Ah | Al | Bh | Bl
    static constexpr int kLdsBytes = 2 * BUF;      // double buffered: 147,456 B -> one workgroup per CU

    struct Stage {
        u32x4 v[4][2];
        const char* src[4][2];
        int lofs[2];
        __device__ __forceinline__ void init(const unsigned short* Ah, const unsigned short* Al, long lda, int m0, int M,
                                             const unsigned short* Bh, const unsigned short* Bl, long ldb, int n0, int N, int tid) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = tid + NTH * r, row = f >> 3, q = f & 7;
                const long ra = (long)min(m0 + row, M - 1) * lda + 8 * q, rb = (long)min(n0 + row, N - 1) * ldb + 8 * q;
                src[0][r] = reinterpret_cast<const char*>(Ah + ra);
                src[1][r] = reinterpret_cast<const char*>(Al + ra);
                src[2][r] = reinterpret_cast<const char*>(Bh + rb);
                src[3][r] = reinterpret_cast<const char*>(Bl + rb);
                lofs[r] = row * ROWB + q * 16;
            }
        }
        __device__ __forceinline__ void load(int k0) {
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
        // sum over k of the staged A values (hi + lo) of staged piece r
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

    // acc[i] (+)= A[m0 + 64 wr + 32 i .., :] . B[n0 + 32 wc .., :]^T over K (a multiple of 64).
    // SIDE_A: rs_generic[128] = sum_k A[m0 + m, k] (the row sums the backward product needs).
    template <bool SIDE_A = false>
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K,
                                               char* lds_generic, f32x16 (&acc)[2], float* rs_generic = nullptr) {
        typedef char __attribute__((address_space(3))) lds_c;
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wr = wave >> 2, wc = wave & 3;
        const int fi = lane & 31, fh = lane >> 5;
        Stage st;
        st.init(Ah, Al, lda, m0, M, Bh, Bl, ldb, n0, N, tid);
        const int nk = K / BK;
        float rsum[2] = {0.f, 0.f};
        auto side = [&]() {
            rsum[0] += st.a_rowpart(0);
            rsum[1] += st.a_rowpart(1);
        };
        st.load(0);
        st.store(lds);
        if constexpr (SIDE_A) side();
        if (nk > 1) st.load(BK);
        __syncthreads();
        auto body = [&](int kt, auto store_next, auto load_next2) {
            const lds_c* buf = lds + (kt & 1) * BUF;
            const lds_c* pa = buf + (wr * 64 + fi) * ROWB + fh * 16;
            const lds_c* pb = buf + 2 * PART + (wc * 32 + fi) * ROWB + fh * 16;
            u32x4 ah[2][4], al[2][4], bh[4], bl[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i][s] = *(const lds_u4*)(pa + i * 32 * ROWB + s * 32);
                    al[i][s] = *(const lds_u4*)(pa + PART + i * 32 * ROWB + s * 32);
                }
                bh[s] = *(const lds_u4*)(pb + s * 32);
                bl[s] = *(const lds_u4*)(pb + PART + s * 32);
            }
            if constexpr (decltype(store_next)::value) {
                st.store(lds + ((kt & 1) ^ 1) * BUF);
                if constexpr (SIDE_A) side();
            }
            if constexpr (decltype(load_next2)::value) st.load((kt + 2) * BK);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 yh = __builtin_bit_cast(bf16x8, bh[s]), yl = __builtin_bit_cast(bf16x8, bl[s]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[i][s]), xl = __builtin_bit_cast(bf16x8, al[i][s]);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc[i], 0, 0, 0);  // small terms first
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc[i], 0, 0, 0);
                }
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
            lds_f* rs_lds = (lds_f*)rs_generic;
#pragma unroll
            for (int r = 0; r < 2; ++r) {  // the eight threads q = 0..7 of a staged row hold its 64 k of a tile
                float s = rsum[r];
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                const int f = tid + NTH * r;
                if ((f & 7) == 0) rs_lds[f >> 3] = s;
            }
            __syncthreads();
        }
    }
    // ---- B given ROW-MAJOR over the contraction index (see GemmBF3::run_bt): B tile [64 k][128 n] bf16, 256-byte rows, the
    // 16-byte chunk c of row r at chunk c ^ ((r & 3) << 2): the four rows of a transposed read land in four different 16-bank
    // quarters whatever the wave's column span (conflict-free, no padding).
    static constexpr int PARTB = 64 * 256;
    static constexpr int BUFT = 2 * PART + 2 * PARTB;
    static constexpr int kLdsBytesT = 2 * BUFT;  // 139,264 B: one workgroup per CU

    struct StageT {
        u32x4 v[4][2];
        const char* srcA[2][2];
        const unsigned short *Bh, *Bl;
        long ldb;
        int lofsA[2], lofsB[2], rowB[2], colB[2], zrows;
        __device__ __forceinline__ void init(const unsigned short* Ah, const unsigned short* Al, long lda, int m0, int M,
                                             const unsigned short* Bh_, const unsigned short* Bl_, long ldb_, int n0, int ncols, int zrows_,
                                             int tid) {
            Bh = Bh_; Bl = Bl_; ldb = ldb_; zrows = zrows_;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int f = tid + NTH * r;
                const int rowa = f >> 3, qa = f & 7;            // A: 128 rows x 8 pieces
                const long ra = (long)min(m0 + rowa, M - 1) * lda + 8 * qa;
                srcA[0][r] = reinterpret_cast<const char*>(Ah + ra);
                srcA[1][r] = reinterpret_cast<const char*>(Al + ra);
                lofsA[r] = rowa * ROWB + qa * 16;
                const int rowb = f >> 4, qb = f & 15;           // B: 64 k rows x 16 pieces
                rowB[r] = rowb;
                colB[r] = min(n0 + 8 * qb, ncols - 8);          // columns past the image are clamped (their outputs are discarded)
                lofsB[r] = rowb * 256 + ((qb ^ ((rowb & 3) << 2)) << 4);
            }
        }
        __device__ __forceinline__ void load(int k0) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                v[0][r] = *reinterpret_cast<const u32x4*>(srcA[0][r] + 2 * (long)k0);
                v[1][r] = *reinterpret_cast<const u32x4*>(srcA[1][r] + 2 * (long)k0);
                const long ob = (long)min(k0 + rowB[r], zrows - 1) * ldb + colB[r];
                v[2][r] = *reinterpret_cast<const u32x4*>(Bh + ob);
                v[3][r] = *reinterpret_cast<const u32x4*>(Bl + ob);
            }
        }
        __device__ __forceinline__ void store(char __attribute__((address_space(3)))* buf) const {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                *(lds_u4*)(buf + lofsA[r]) = v[0][r];
                *(lds_u4*)(buf + PART + lofsA[r]) = v[1][r];
                *(lds_u4*)(buf + 2 * PART + lofsB[r]) = v[2][r];
                *(lds_u4*)(buf + 2 * PART + PARTB + lofsB[r]) = v[3][r];
            }
        }
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

    // Bh/Bl: [zrows, ldb] row-major with `ncols` (a multiple of 8) valid columns; K a multiple of 64.
    template <bool SIDE_A = false>
    __device__ static __forceinline__ void run_bt(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                                  const unsigned short* Bl, long ldb, int ncols, int zrows, int m0, int n0, int M, int K,
                                                  char* lds_generic, f32x16 (&acc)[2], float* rs_generic = nullptr) {
        typedef char __attribute__((address_space(3))) lds_c;
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wr = wave >> 2, wc = wave & 3;
        const int fi = lane & 31, fh = lane >> 5;
        const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
        const int chunk = 4 * wc + 2 * (g & 1) + (pp >> 1);
        const int offB0 = (8 * (g >> 1) + qq) * 256 + ((chunk ^ (qq << 2)) << 4) + (pp & 1) * 8;
        StageT st;
        st.init(Ah, Al, lda, m0, M, Bh, Bl, ldb, n0, ncols, zrows, tid);
        const int nk = K / BK;
        float rsum[2] = {0.f, 0.f};
        auto side = [&]() {
            rsum[0] += st.a_rowpart(0);
            rsum[1] += st.a_rowpart(1);
        };
        st.load(0);
        st.store(lds);
        if constexpr (SIDE_A) side();
        if (nk > 1) st.load(BK);
        __syncthreads();
        auto body = [&](int kt, auto store_next, auto load_next2) {
            const lds_c* buf = lds + (kt & 1) * BUFT;
            const lds_c* pa = buf + (wr * 64 + fi) * ROWB + fh * 16;
            const lds_c* pb = buf + 2 * PART + offB0;
            u32x4 ah[2][4], al[2][4];
            bf16x8 yh[4], yl[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i][s] = *(const lds_u4*)(pa + i * 32 * ROWB + s * 32);
                    al[i][s] = *(const lds_u4*)(pa + PART + i * 32 * ROWB + s * 32);
                }
                yh[s] = tr_read_k8(pb + s * 4096, 1024);
                yl[s] = tr_read_k8(pb + PARTB + s * 4096, 1024);
            }
            if constexpr (decltype(store_next)::value) {
                st.store(lds + ((kt & 1) ^ 1) * BUFT);
                if constexpr (SIDE_A) side();
            }
            if constexpr (decltype(load_next2)::value) st.load((kt + 2) * BK);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[i][s]), xl = __builtin_bit_cast(bf16x8, al[i][s]);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh[s], acc[i], 0, 0, 0);  // small terms first
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl[s], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh[s], acc[i], 0, 0, 0);
                }
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
            lds_f* rs_lds = (lds_f*)rs_generic;
#pragma unroll
            for (int r = 0; r < 2; ++r) {  // the eight threads q = 0..7 of a staged A row hold its 64 k of a tile
                float s = rsum[r];
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                const int f = tid + NTH * r;
                if ((f & 7) == 0) rs_lds[f >> 3] = s;
            }
            __syncthreads();
        }
    }

    __device__ static __forceinline__ int sub_row(int i, int r) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave >> 2) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int sub_col() {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave & 3) * 32 + (lane & 31);
    }
};

}  // namespace vgan
