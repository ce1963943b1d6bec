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


// ---- the split-bf16 tile main loop on 128x128 tiles: 512 threads, one workgroup per CU, K tile of 64 ------------------
// Staging is DIRECT-TO-LDS (`global_load_lds_dwordx4`): no staging registers and no `ds_write` pass.  An LDS-DMA writes
// `wave-uniform base + 16 * lane`: the images are lane-linear, unpadded, and the bank swizzle sits in the SOURCE address.
//   A-type part (Ah, Al, and Bh, Bl of `run`): 128 rows x 128 B; the 16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
//     (a `ds_read_b128` is served in groups of 16 lanes = 16 rows of one chunk column: 8 even rows on banks 0-31 and 8 odd
//     rows on banks 32-63, and (r >> 1) & 7 takes eight distinct values over each -- conflict-free).
//   row-major B part (Bh, Bl of `run_bt`): 64 k rows x 256 B, chunk c of row r at c ^ ((r & 3) << 2) (as in GemmBF3::run_bt).
// Wave layout: two GROUPS of 2 x 2 waves; wave (grp, R, C) multiplies the 64 x 64 quadrant (R, C) of the tile over the k16
// steps 2 grp, 2 grp + 1 of every K tile (8 fragment reads for 12 MFMAs; eight 64 x 32 waves over all four steps read 6 for
// 6).  The groups run HALF AN ITERATION APART, a raw barrier per half: while one group is on the MFMA pipe the other issues
// its fragment reads and its share of the next tile's fill, and every SIMD hosts one wave of each.  After the loop the groups
// exchange halves through LDS: wave (grp, R, C) ends with rows 64 R .., columns 64 C + 32 grp .. (sub_row / sub_col).
// Measured (4 096 tiles, K = 4 096, operands L2 / Infinity-Cache resident, ten warm-up launches; A/B binaries alternated on
// one box): 1.275 ms = 0.517 of the nominal 2.5 PFLOP/s executed, against 1.32 ms = 0.50 for the loop this replaces (eight
// 64 x 32 waves, register staging, one barrier per K tile) -- +3.5 %.  Ceilings on the same data (tools/ablate_big_glds.hip):
// the MFMAs alone (fragments held, a barrier per 24) 0.68-0.70 -- under non-constant operands the shader clock drops to ~78 %
// of what a data-movement-only loop runs at -- and the fill alone 0.67-0.68 (64 KB per K tile and CU at the ~69 GB/s a CU
// takes in).  The first launches after an idle period run ~20 % slower (clock ramp): unwarmed timings of this loop mislead.
// At the 64x64 tile's two workgroups per CU direct-to-LDS staging measured nothing (tools/ablate_bf3_glds.hip: 14.0 vs 14.0
// us at 392 tiles, 15.0 vs 15.2 at 512), so GemmBF3 keeps its register staging.
struct GemmBF3Big {
    static constexpr int BM = 128, BN = 128, BK = 64, NTH = 512;
    static constexpr int PART = 128 * 128;         // one A-type operand part
    static constexpr int PARTB = 64 * 256;         // one row-major B part (same size)
    static constexpr int BUF = 4 * PART;           // Ah | Al | Bh | Bl
    static constexpr int BUFT = 2 * PART + 2 * PARTB;
    static constexpr int kLdsBytes = 2 * BUF;      // double buffered: 131,072 B -> one workgroup per CU
    static constexpr int kLdsBytesT = 2 * BUFT;
    typedef char __attribute__((address_space(3))) lds_c;

    struct Quad {
        f32x16 a[2][2];
        __device__ __forceinline__ void zero() {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[i][j][r] = 0.f;
        }
        // three products per (i, j), small terms first; the four accumulators alternate between dependent MFMAs
        __device__ __forceinline__ void mac(const bf16x8 (&xh)[2], const bf16x8 (&xl)[2], const bf16x8 (&yh)[2], const bf16x8 (&yl)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) a[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[j], a[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) a[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[j], a[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) a[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[j], a[i][j], 0, 0, 0);
        }
        // acc[i] += own half + the partner group's half (the main loop's LDS traffic is behind a barrier already)
        __device__ __forceinline__ void exchange(lds_c* lds, f32x16 (&acc)[2]) const {
            lds_f* xch = (lds_f*)lds;  // 8 waves x 2 tiles x 16 registers x 64 lanes x 4 B = 64 KB
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((wave * 2 + i) * 16 + r) * 64 + lane] = grp ? a[i][0][r] : a[i][1][r];
            __syncthreads();
            const int partner = wave ^ 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] += (grp ? a[i][1][r] : a[i][0][r]) + xch[((partner * 2 + i) * 16 + r) * 64 + lane];
            __syncthreads();  // the caller's epilogue reuses the LDS
        }
    };

    // one wave's share of a K tile's fill: eight 1-KB pieces.
    // FillRows: A-type part `part` (0..3 of Ah, Al, Bh, Bl), rows half * 64 + 8 e + (lane >> 3) of the tile's 128
    struct FillRows {
        const char* src[8];
        int dst0;
        __device__ __forceinline__ void init(const unsigned short* base, long ld, int r0, int lim, int part, int half, int lane) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = half * 64 + 8 * e + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                src[e] = reinterpret_cast<const char*>(base + (long)min(r0 + row, lim - 1) * ld) + 16 * c;
            }
            dst0 = part * PART + half * 8192;
        }
        __device__ __forceinline__ void issue(lds_c* buf, int kt) const {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[e] + 2 * (long)kt * BK),
                                                 (void __attribute__((address_space(3)))*)(buf + dst0 + e * 1024), 16, 0, 0);
        }
    };
    // FillRM: row-major B part (0: Bh, 1: Bl), image rows half * 32 + 4 e + (lane >> 4) = k index inside the tile, columns
    // n0 + 8 c (columns past the image are clamped: their outputs are discarded; rows past `zrows` are clamped per tile)
    struct FillRM {
        const char* src[8];
        long ldb2;
        int krow0, zrows, dst0;
        __device__ __forceinline__ void init(const unsigned short* base, long ld, int n0, int ncols, int zrows_, int bpart, int half, int lane) {
            ldb2 = 2 * ld;
            zrows = zrows_;
            krow0 = half * 32 + (lane >> 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = krow0 + 4 * e;  // (row & 3) == (lane >> 4) & 3 for every e: one column per lane
                const int c = (lane & 15) ^ ((row & 3) << 2);
                src[e] = reinterpret_cast<const char*>(base + min(n0 + 8 * c, ncols - 8));
            }
            dst0 = 2 * PART + bpart * PARTB + half * 8192;
        }
        __device__ __forceinline__ void issue(lds_c* buf, int kt) const {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const char* g = src[e] + (long)min(kt * BK + krow0 + 4 * e, zrows - 1) * ldb2;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                                 (void __attribute__((address_space(3)))*)(buf + dst0 + e * 1024), 16, 0, 0);
            }
        }
    };

    // The ping-pong schedule shared by run and run_bt.  fill(kt) issues this wave's share of tile kt, frags(kt) its fragment
    // reads, mfmas() the 24 MFMAs.  Slot 2 kt: group 0 fills tile kt + 1 and reads tile kt while group 1 fills its share of
    // tile kt + 1 and multiplies tile kt - 1; slot 2 kt + 1: group 0 multiplies tile kt, group 1 reads it; both drain their
    // fills before the barrier that ends the odd slot.  Hazards: tile kt + 1 lands in the buffer of tile kt - 1, whose last
    // reader (group 1, slot 2 kt - 1) drained its LDS reads before that slot's barrier; tile kt + 1 is first read in slot
    // 2 kt + 2, after both shares were drained in slot 2 kt + 1.  Every wave passes 2 nk + 1 barriers.
    template <class Fill0, class Fill1, class FragF, class MfmaF>
    __device__ static __forceinline__ void pingpong(int grp, int nk, Fill0 fill0, Fill1 fill1, FragF frags, MfmaF mfmas) {
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);  // MFMAs are register-only: without this the scheduler moves them across
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto drain = [&]() { __builtin_amdgcn_s_waitcnt(0x0070); };  // vmcnt(0) lgkmcnt(0)
        if (grp == 0) {
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) fill0(kt + 1);
                frags(kt);
                bar();
                mfmas();
                drain();
                bar();
            }
            bar();
        } else {
            if (1 < nk) fill1(1);
            bar();
            for (int kt = 0; kt < nk; ++kt) {
                frags(kt);
                drain();
                bar();
                if (kt + 2 < nk) fill1(kt + 2);
                mfmas();
                bar();
            }
        }
        __syncthreads();
    }

    // Row sums of A = Ah + Al from the fragments themselves (there are no staging registers to take them from): the waves
    // (grp, R, 0) and (grp, R, 1) read the same A fragments, so wave C sums the fragments of its k16 step 2 grp + C only --
    // 64 VALU operations per K tile and wave, issued beside the MFMAs.  (`v_dot2c_f32_bf16` against a vector of ones would be
    // one instruction per pair, but returned wrong sums in this loop on gfx950 -- tools/check_rowsum_big.hip -- while the same
    // instruction in a plain loop is exact; the shift / mask / add form is used.)
    __device__ static __forceinline__ float frag_sum(const u32x4& h, const u32x4& l, float s) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s += __uint_as_float(h[e] << 16) + __uint_as_float(h[e] & 0xFFFF0000u);
            s += __uint_as_float(l[e] << 16) + __uint_as_float(l[e] & 0xFFFF0000u);
        }
        return s;
    }
    template <class Frag>
    __device__ static __forceinline__ void rowsum_step(const Frag (&ah)[2][2], const Frag (&al)[2][2], int C, float (&rsum)[2]) {
        if (C == 0) {  // wave-uniform
            rsum[0] = frag_sum(ah[0][0], al[0][0], rsum[0]);
            rsum[1] = frag_sum(ah[0][1], al[0][1], rsum[1]);
        } else {
            rsum[0] = frag_sum(ah[1][0], al[1][0], rsum[0]);
            rsum[1] = frag_sum(ah[1][1], al[1][1], rsum[1]);
        }
    }
    // tmp: 512 floats of LDS scratch, [grp][C][128 rows]; a lane pair (fh = 0, 1) holds the two k halves of a step
    __device__ static __forceinline__ void rowsum_publish(const float (&rsum)[2], lds_f* tmp, int grp, int R, int C, int fi, int fh) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float s = rsum[i] + __shfl_xor(rsum[i], 32, 64);
            if (fh == 0) tmp[(grp * 2 + C) * 128 + R * 64 + i * 32 + fi] = s;
        }
    }
    __device__ static __forceinline__ void rowsum_fold(const lds_f* tmp, float* rs_generic) {
        if (threadIdx.x < 128) {
            const int t = threadIdx.x;
            ((lds_f*)rs_generic)[t] = (tmp[t] + tmp[128 + t]) + (tmp[256 + t] + tmp[384 + t]);
        }
        __syncthreads();
    }

    // acc[i] (+)= (A[m0 .., :] . B[n0 .., :]^T)[sub_row(i, .), sub_col()] over K (a multiple of 64).
    // SIDE_A: rs_generic[128] = sum_k A[m0 + m, k] (the row sums the backward product needs).
    template <bool SIDE_A = false>
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K,
                                               char* lds_generic, f32x16 (&acc)[2], float* rs_generic = nullptr) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int grp = wave >> 2, R = (wave >> 1) & 1, C = wave & 1;
        const int fi = lane & 31, fh = lane >> 5;
        const int part = wave >> 1;  // this wave fills half (wave & 1) of part Ah, Al, Bh or Bl
        FillRows fl;
        fl.init(part == 0 ? Ah : part == 1 ? Al : part == 2 ? Bh : Bl, part < 2 ? lda : ldb, part < 2 ? m0 : n0, part < 2 ? M : N, part, wave & 1,
                lane);
        const int nk = K / BK;
        Quad qd;
        qd.zero();
        float rsum[2] = {0.f, 0.f};
        fl.issue(lds, 0);
        __syncthreads();
        const int sw = (fi >> 1) & 7;
        u32x4 ah[2][2], al[2][2], bh[2][2], bl[2][2];  // [step][i or j]
        auto fill = [&](int kt) { fl.issue(lds + (kt & 1) * BUF, kt); };
        auto frags = [&](int kt) {
            const lds_c* buf = lds + (kt & 1) * BUF;
            const lds_c* pa = buf + (R * 64 + fi) * 128;
            const lds_c* pb = buf + 2 * PART + (C * 64 + fi) * 128;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pos = ((2 * (2 * grp + s) + fh) ^ sw) << 4;  // k16 steps 2 grp, 2 grp + 1
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[s][i] = *(const lds_u4*)(pa + i * 32 * 128 + pos);
                    al[s][i] = *(const lds_u4*)(pa + PART + i * 32 * 128 + pos);
                    bh[s][i] = *(const lds_u4*)(pb + i * 32 * 128 + pos);
                    bl[s][i] = *(const lds_u4*)(pb + PART + i * 32 * 128 + pos);
                }
            }
        };
        auto mfmas = [&]() {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 xh[2] = {__builtin_bit_cast(bf16x8, ah[s][0]), __builtin_bit_cast(bf16x8, ah[s][1])};
                const bf16x8 xl[2] = {__builtin_bit_cast(bf16x8, al[s][0]), __builtin_bit_cast(bf16x8, al[s][1])};
                const bf16x8 yh[2] = {__builtin_bit_cast(bf16x8, bh[s][0]), __builtin_bit_cast(bf16x8, bh[s][1])};
                const bf16x8 yl[2] = {__builtin_bit_cast(bf16x8, bl[s][0]), __builtin_bit_cast(bf16x8, bl[s][1])};
                qd.mac(xh, xl, yh, yl);
            }
            if constexpr (SIDE_A) rowsum_step(ah, al, C, rsum);
        };
        pingpong(grp, nk, fill, fill, frags, mfmas);
        if constexpr (SIDE_A) rowsum_publish(rsum, (lds_f*)(lds + 65536), grp, R, C, fi, fh);
        qd.exchange(lds, acc);
        if constexpr (SIDE_A) rowsum_fold((const lds_f*)(lds + 65536), rs_generic);
    }

    // Bh/Bl: [zrows, ldb] row-major with `ncols` (a multiple of 8) valid columns; K a multiple of 64.
    template <bool SIDE_A = false>
    __device__ static __forceinline__ void run_bt(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                                  const unsigned short* Bl, long ldb, int ncols, int zrows, int m0, int n0, int M, int K,
                                                  char* lds_generic, f32x16 (&acc)[2], float* rs_generic = nullptr) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int grp = wave >> 2, R = (wave >> 1) & 1, C = wave & 1;
        const int fi = lane & 31, fh = lane >> 5;
        const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
        // 32-column block cb = 2 C + j of the tile: 16-byte chunks 4 cb .. 4 cb + 3 of a 256-byte image row
        int offB[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int chunk = 4 * (2 * C + j) + 2 * (g & 1) + (pp >> 1);
            offB[j] = (8 * (g >> 1) + qq) * 256 + ((chunk ^ (qq << 2)) << 4) + (pp & 1) * 8;
        }
        FillRows fa;  // group 0 fills the A parts, group 1 the row-major B parts
        FillRM fb;
        const int hp = (wave >> 1) & 1;  // hi or lo part
        if (grp == 0)
            fa.init(hp == 0 ? Ah : Al, lda, m0, M, hp, wave & 1, lane);
        else
            fb.init(hp == 0 ? Bh : Bl, ldb, n0, ncols, zrows, hp, wave & 1, lane);
        const int nk = K / BK;
        Quad qd;
        qd.zero();
        float rsum[2] = {0.f, 0.f};
        if (grp == 0)
            fa.issue(lds, 0);
        else
            fb.issue(lds, 0);
        __syncthreads();
        const int sw = (fi >> 1) & 7;
        u32x4 ah[2][2], al[2][2];
        bf16x8 yh[2][2], yl[2][2];
        auto fill0 = [&](int kt) { fa.issue(lds + (kt & 1) * BUFT, kt); };
        auto fill1 = [&](int kt) { fb.issue(lds + (kt & 1) * BUFT, kt); };
        auto frags = [&](int kt) {
            const lds_c* buf = lds + (kt & 1) * BUFT;
            const lds_c* pa = buf + (R * 64 + fi) * 128;
            const lds_c* pb = buf + 2 * PART + grp * 2 * 4096;  // k16 step s: 16 image rows of 256 B
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pos = ((2 * (2 * grp + s) + fh) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[s][i] = *(const lds_u4*)(pa + i * 32 * 128 + pos);
                    al[s][i] = *(const lds_u4*)(pa + PART + i * 32 * 128 + pos);
                    yh[s][i] = tr_read_k8(pb + offB[i] + s * 4096, 1024);
                    yl[s][i] = tr_read_k8(pb + PARTB + offB[i] + s * 4096, 1024);
                }
            }
        };
        auto mfmas = [&]() {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 xh[2] = {__builtin_bit_cast(bf16x8, ah[s][0]), __builtin_bit_cast(bf16x8, ah[s][1])};
                const bf16x8 xl[2] = {__builtin_bit_cast(bf16x8, al[s][0]), __builtin_bit_cast(bf16x8, al[s][1])};
                qd.mac(xh, xl, yh[s], yl[s]);
            }
            if constexpr (SIDE_A) rowsum_step(ah, al, C, rsum);
        };
        pingpong(grp, nk, fill0, fill1, frags, mfmas);
        if constexpr (SIDE_A) rowsum_publish(rsum, (lds_f*)(lds + 65536), grp, R, C, fi, fh);
        qd.exchange(lds, acc);
        if constexpr (SIDE_A) rowsum_fold((const lds_f*)(lds + 65536), rs_generic);
    }

    // the block a wave holds AFTER the exchange: rows 64 R + 32 i .., columns 64 C + 32 grp ..
    __device__ static __forceinline__ int sub_row(int i, int r) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return ((wave >> 1) & 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ static __forceinline__ int sub_col() {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return ((wave & 1) * 2 + (wave >> 2)) * 32 + (lane & 31);
    }
};

}  // namespace vgan
