// Split-bf16 tile main loop on 256 x 128 tiles with dedicated loader waves (c4 / c5 sizes): the third tile engine of
// mmd_bf16.hip beside GemmBF3 (64 x 64, c3) and GemmBF3Big (128 x 128).
//
// Why (tools/wide_proto.hip, MI355X, 8192 x 8192 x 4096, warm operands): the 128 x 128 loop moves 64 KB per K tile through
// LDS for 6.3 MFLOP and keeps ONE tile of prefetch in flight -- its fill alone runs at 0.67 of the bf16 peak, its MFMAs alone at
// 0.68-0.70, together 0.49-0.52.  Here
//   * the tile is 256 x 128: 3/4 of the fill bytes per flop;
//   * a K stage is 32 deep (48 KB: Ah | Al 256 rows, Bh | Bl 128 rows, 64-byte rows) and THREE stages live in LDS (144 KB),
//     so two stages are in flight behind the one being multiplied.  A raw s_barrier ends a stage: __syncthreads() is a
//     workgroup-scope fence and drains vmcnt(0), i.e. the very loads that are meant to stay in flight (measured: 0.68 -> 0.80
//     for the fill alone);
//   * the workgroup has 768 threads: 8 CONSUMER waves (a 64 x 64 quadrant each as 4 x 4 blocks of v_mfma_f32_16x16x32_bf16,
//     which holds a higher clock under load than 32x32x16 -- MI355X_MICROARCH.md, "DVFS give-back" item 7 -- 0.81 against
//     0.68-0.70 for the bare MFMA loops) and 4 LOADER waves that issue the whole fill (12 one-KB global_load_lds pieces each per
//     stage), own the vmcnt waits and run at raised priority.  A consumer never issues a DMA (~100 cycles of issue each
//     inside a busy phase) and never waits on vmcnt: 0.53 -> 0.57-0.61 on the same box, bit-identical results;
//   * the two consumer groups (waves 0-3, 4-7) run half a stage apart (two barriers per stage: `consumers` below), +2.7 %;
//   * the loaders also take the row sums of A the backward product needs, from the landed LDS image (they are idle otherwise).
// Same box, binaries alternated, ten warm-up launches: 0.595-0.610 of the nominal 2.5 PFLOP/s executed against 0.512-0.517 for
// GemmBF3Big (+17-19 %).
#pragma once
#include "gemm_bf3.hpp"

namespace vgan {

typedef f32x4 f32x4w;  // the C/D fragment of v_mfma_f32_16x16x32_bf16

struct GemmBF3Wide {
    static constexpr int BM = 256, BN = 128, BK = 32, NTH = 768, NCONS = 512, NST = 3;
    static constexpr int PA = BM * BK * 2, PB = BN * BK * 2;  // bytes of an A / B part of one stage: 16 KB / 8 KB
    static constexpr int STAGE = 2 * PA + 2 * PB;             // Ah | Al | Bh | Bl = 48 KB
    static constexpr int kLdsBytes = NST * STAGE;             // 147,456 B: one workgroup per CU
    typedef char __attribute__((address_space(3))) lds_c;

    // A-type part (Ah, Al, and Bh, Bl of `run`): rows of 64 B = 4 chunks of 16 B (8 k each); chunk c of row r sits at position
    // c ^ hsw(r).  A ds_read_b128 of a 16-row fragment block (lane = row + 16 * chunk) is served in the instruction's four
    // groups of 16 lanes -- {0-3, 12-15, 20-27}, ... -- and hsw spreads each group over the sixteen 16-byte bank slots.
    __device__ static __forceinline__ int hsw(int r) { return (-(r >> 2)) & 3; }
    // row-major B part of `run_bt`: 32 k rows of 256 B (128 n); chunk c (of 16) of row r at c ^ tsw(r): a 32-lane half of a
    // ds_read_b64_tr_b16 touches 8 rows x 32 B (rows 8 g + q, g in {0, 1} or {2, 3}) and tsw sends them to the eight distinct
    // 32-byte bank ranges.
    __device__ static __forceinline__ int tsw(int r) { return ((r & 3) << 1) | (((r >> 3) & 1) << 3); }

    __device__ static __forceinline__ void bar() {
        __builtin_amdgcn_sched_barrier(0);  // (MFMAs are register-only: without this the scheduler moves them across)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ static __forceinline__ bool is_loader() { return threadIdx.x >= NCONS; }

    // One loader wave's share of a stage: 12 of the 48 one-KB pieces [Ah: 16 | Al: 16 | Bh: 8 | Bl: 8].
    // A-type piece = 16 rows x 64 B (lane l: row l >> 2, position l & 3); row-major B piece = 4 k rows x 256 B (lane l: row
    // l >> 4, position l & 15).  The LDS image is lane-linear; the swizzle sits in the SOURCE address.
    template <bool RM>
    struct Fill {
        const char* src[12];
        int dst[12];
        long kstep[12];      // byte advance of the source per K stage
        int krow[12];        // RM B pieces: k row inside the stage (for the clamp against zrows), else -1
        long ldb2;
        int zrows;
        __device__ __forceinline__ void init(int lw, int lane, const unsigned short* Ah, const unsigned short* Al, long lda, int m0, int M,
                                             const unsigned short* Bh, const unsigned short* Bl, long ldb, int n0, int N, int zrows_) {
            ldb2 = 2 * ldb;
            zrows = zrows_;
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int pc = 12 * lw + e;
                const int part = pc < 16 ? 0 : pc < 32 ? 1 : pc < 40 ? 2 : 3;
                const int pin = part == 0 ? pc : part == 1 ? pc - 16 : part == 2 ? pc - 32 : pc - 40;
                dst[e] = (part == 0 ? 0 : part == 1 ? PA : part == 2 ? 2 * PA : 2 * PA + PB) + pin * 1024;
                if (part < 2 || !RM) {
                    const int row = 16 * pin + (lane >> 2);
                    const int c = (lane & 3) ^ hsw(row);
                    const unsigned short* base = part == 0 ? Ah : part == 1 ? Al : part == 2 ? Bh : Bl;
                    const long ld = part < 2 ? lda : ldb;
                    const int g0 = part < 2 ? m0 : n0, lim = part < 2 ? M : N;
                    src[e] = reinterpret_cast<const char*>(base + (long)min(g0 + row, lim - 1) * ld) + 16 * c;
                    kstep[e] = 2 * BK;
                    krow[e] = -1;
                } else {  // B [k, n] row-major: N = valid columns (a multiple of 8); rows are clamped per stage in issue()
                    const int row = 4 * pin + (lane >> 4);
                    const int c = (lane & 15) ^ tsw(row);
                    src[e] = reinterpret_cast<const char*>((part == 2 ? Bh : Bl) + min(n0 + 8 * c, N - 8));
                    kstep[e] = 0;
                    krow[e] = row;
                }
            }
        }
        __device__ __forceinline__ void issue(lds_c* lds, int kt) const {
            lds_c* d = lds + (kt % NST) * STAGE;
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const char* g = src[e] + kstep[e] * (long)kt;
                if (RM && krow[e] >= 0) g += (long)min(kt * BK + krow[e], zrows - 1) * ldb2;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                                 (void __attribute__((address_space(3)))*)(d + dst[e]), 16, 0, 0);
            }
        }
    };

    // sum over the stage's 32 k of (Ah + Al)[row]: the loader's lane reads its row's four chunks of both parts (any order:
    // a sum)
    __device__ static __forceinline__ float row_part(const lds_c* st, int row, float s) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
        // eight independent chains, dependent instructions eight apart (v_dot2c_f32_bf16 accumulates in place)
        u32x4 h[4], l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // the lane walks its row's four chunks from a rotated start: 16 lanes of a ds_read_b128
            const int pos = ((c + (row >> 2)) & 3) << 4;  // group then hit 16 different 16-byte bank slots (plain order: 4-way conflicts)
            h[c] = *(const lds_u4*)(st + row * 64 + pos);
            l[c] = *(const lds_u4*)(st + PA + row * 64 + pos);
        }
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // (each dword passes through an empty asm: hipcc 7.2 otherwise folds the four element extracts of a vector into
                //  ONE source register for this builtin -- the "wrong sums" noted at GemmBF3Big::frag_sum)
                unsigned he = h[c][e], le = l[c][e];
                asm volatile("" : "+v"(he), "+v"(le));
                a[2 * c] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, he), ones, a[2 * c], false);
                a[2 * c + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, le), ones, a[2 * c + 1], false);
            }
        const float s2 = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        return s + s2;
    }

    // Schedule of a tile (prototype: tools/wide_proto.hip, WidePP: 0.615-0.623 against 0.600-0.605 with all eight consumers in
    // lockstep, bit-identical).  The consumer waves form two GROUPS, waves 0-3 and 4-7 -- one wave of each on every SIMD -- that run
    // HALF A STAGE APART behind two raw barriers per stage: while one group multiplies a stage the other reads its fragments, so a
    // SIMD's matrix pipe always has exactly one wave on it instead of two waves reading together and then sharing the pipe.
    // After the prologue barrier P the barriers are B(0) .. B(2 nk):
    //   group 0:  [read kt] B(2 kt) [multiply kt] B(2 kt + 1) ...              and one closing barrier;
    //   group 1:  B(0), then [read kt] B(2 kt + 1) [multiply kt] B(2 kt + 2) ...
    //   loaders:  stage kt must have landed before B(2 kt - 1) (group 0 reads it behind that barrier) and its buffer is free
    //             behind B(2 kt + 1) (both groups have read it): stage kt + 3 is issued there -- two stages in flight.
    __device__ static __forceinline__ void wait_landed(bool younger_in_flight) {
        if (younger_in_flight) __builtin_amdgcn_s_waitcnt(0x0F70 | 12); else __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(12) / vmcnt(0)
    }

    // The loader wave's whole life inside a tile.  SIDE_A: rs[256] (LDS floats) = sum_k A[m0 + m, k].
    template <bool RM, bool SIDE_A>
    __device__ static __forceinline__ void loader(const Fill<RM>& fl, lds_c* lds, int nk, float* rs_generic) {
        __builtin_amdgcn_s_setprio(3);
        const int lane = threadIdx.x & 63, lw = (threadIdx.x - NCONS) >> 6;
        float rsum = 0.f;
        fl.issue(lds, 0);
        if (nk > 1) fl.issue(lds, 1);
        if (nk > 2) fl.issue(lds, 2);
        if (nk > 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (24 & 15) | ((24 >> 4) << 14));  // vmcnt(24): stage 0 has landed
        else wait_landed(nk > 1);
        bar();  // P
        if constexpr (SIDE_A) rsum = row_part(lds, 64 * lw + lane, rsum);  // stage 0 (its buffer lives until B(1))
        for (int j = 0; j <= 2 * nk; ++j) {
            if (j & 1) {
                const int need = (j + 1) >> 1, next = (j + 5) >> 1;
                if (need < nk) wait_landed(need + 1 < nk);
                bar();
                if (next < nk) fl.issue(lds, next);  // into the buffer of stage next - 3, read for the last time before this barrier
                if constexpr (SIDE_A)
                    if (need < nk) rsum = row_part(lds + (need % NST) * STAGE, 64 * lw + lane, rsum);  // lives until B(j + 2)
            } else {
                bar();
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (SIDE_A) ((lds_f*)rs_generic)[64 * lw + lane] = rsum;
    }

    // the consumer waves' two-group schedule around `frags(kt)` (fragment reads of stage kt, drained) and `mfmas()`
    template <class FragF, class MfmaF>
    __device__ static __forceinline__ void consumers(int grp, int nk, FragF frags, MfmaF mfmas) {
        bar();  // P
        if (grp == 0) {
            for (int kt = 0; kt < nk; ++kt) {
                frags(kt);
                bar();
                mfmas();
                bar();
            }
            bar();
        } else {
            bar();
            for (int kt = 0; kt < nk; ++kt) {
                frags(kt);
                bar();
                mfmas();
                bar();
            }
        }
    }

    // The LOCKSTEP schedule (one barrier per stage, all eight consumers in the same phase), kept for the backward product: its
    // loaders also form the row sums, and with two barriers per stage that work sits between two barriers the consumers wait at
    // -- same box, alternating: backward 2.99-3.04 ms staggered against 2.85-2.87 lockstep at c5 (the Gram, whose loaders only
    // load, gains: 2.74-2.77 against 2.79-2.81).
    template <bool RM, bool SIDE_A>
    __device__ static __forceinline__ void loader_lockstep(const Fill<RM>& fl, lds_c* lds, int nk, float* rs_generic) {
        __builtin_amdgcn_s_setprio(3);
        const int lane = threadIdx.x & 63, lw = (threadIdx.x - NCONS) >> 6;
        float rsum = 0.f;
        fl.issue(lds, 0);
        if (nk > 1) fl.issue(lds, 1);
        wait_landed(nk > 1);  // stage 0 has landed
        bar();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 2 < nk) fl.issue(lds, kt + 2);   // into the buffer of stage kt - 1: its readers passed the last barrier
            if constexpr (SIDE_A) rsum = row_part(lds + (kt % NST) * STAGE, 64 * lw + lane, rsum);
            // stage kt + 1 must have landed before the barrier lets anyone read it; stage kt + 2 (the 12 youngest) stays in flight
            wait_landed(kt + 2 < nk);
            bar();
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (SIDE_A) ((lds_f*)rs_generic)[64 * lw + lane] = rsum;
    }
    // the 48 MFMAs of a stage: three products per 16 x 16 block, small terms first
    __device__ static __forceinline__ void mac(f32x4w (&acc)[4][4], const bf16x8 (&xh)[4], const bf16x8 (&xl)[4], const bf16x8 (&yh)[4],
                                               const bf16x8 (&yl)[4], bool first_only, bool rest_only) {
        if (!rest_only) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[i], yh[j], acc[i][j], 0, 0, 0);
        }
        if (first_only) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[i], yl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[i], yh[j], acc[i][j], 0, 0, 0);
    }


    // acc[i][j] (+)= (A[m0 .., :] . B[n0 .., :]^T) over K (a multiple of 32), block (i, j) of this consumer wave's quadrant:
    // element r of acc[i][j] is row sub_row(i, r), column sub_col(j).  All 768 threads call; loader threads come back with acc
    // untouched.  Ends with one __syncthreads(): the LDS is free for the caller's epilogue.
    template <bool SIDE_A = false>
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x4w (&acc)[4][4], float* rs_generic = nullptr) {
        lds_c* lds = (lds_c*)lds_generic;
        const int nk = K / BK;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (wave >= 8) {
            Fill<false> fl;
            fl.init(wave - 8, lane, Ah, Al, lda, m0, M, Bh, Bl, ldb, n0, N, 0);
            loader<false, SIDE_A>(fl, lds, nk, rs_generic);
        } else {
            const int R = wave >> 1, C = wave & 1;
            const int fr = lane & 15, fc = lane >> 4;
            const int pos = (fc ^ hsw(fr)) << 4;  // rows 16 b + fr: (row >> 2) & 3 = (fr >> 2) & 3 for every block b
            bf16x8 xh[4], xl[4], yh[4], yl[4];
            auto frags = [&](int kt) {
                const lds_c* st = lds + (kt % NST) * STAGE;
                const lds_c* pa = st + (R * 64 + fr) * 64 + pos;
                const lds_c* pb = st + 2 * PA + (C * 64 + fr) * 64 + pos;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    xl[b] = __builtin_bit_cast(bf16x8, *(const lds_u4*)(pa + PA + b * 1024));
                    yh[b] = __builtin_bit_cast(bf16x8, *(const lds_u4*)(pb + b * 1024));
                    xh[b] = __builtin_bit_cast(bf16x8, *(const lds_u4*)(pa + b * 1024));
                    yl[b] = __builtin_bit_cast(bf16x8, *(const lds_u4*)(pb + PB + b * 1024));
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): read before the barrier that lets the buffer go
            };
            auto mfmas = [&]() { mac(acc, xh, xl, yh, yl, false, false); };
            consumers(wave >> 2, nk, frags, mfmas);
        }
        __syncthreads();
    }

    // The same with B given ROW-MAJOR over the contraction index: acc (+)= A[m0 .., k] . B[k, n0 ..]  (the backward product
    // W . Z on Z's row-major split images).  Bh / Bl: [zrows, ldb] with `ncols` (a multiple of 8) valid columns.
    template <bool SIDE_A = false>
    __device__ static __forceinline__ void run_bt(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                                  const unsigned short* Bl, long ldb, int ncols, int zrows, int m0, int n0, int M, int K,
                                                  char* lds_generic, f32x4w (&acc)[4][4], float* rs_generic = nullptr) {
        lds_c* lds = (lds_c*)lds_generic;
        const int nk = K / BK;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (wave >= 8) {
            Fill<true> fl;
            fl.init(wave - 8, lane, Ah, Al, lda, m0, M, Bh, Bl, ldb, n0, ncols, zrows);
            loader_lockstep<true, SIDE_A>(fl, lds, nk, rs_generic);
        } else {
            const int R = wave >> 1, C = wave & 1;
            const int fr = lane & 15, fc = lane >> 4;
            const int pos = (fc ^ hsw(fr)) << 4;
            // transposed-read geometry (tr_read_k8): group g = lane >> 4 reads k rows 8 g + q (then + 4) of the stage, lane
            // 4 q + p of the group supplies row q, columns 4 p .. 4 p + 3 of the 16-column block jb = 4 C + j
            const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
            int offB[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 8 * g + qq, chunk = 2 * (4 * C + j) + (pp >> 1);
                offB[j] = row * 256 + ((chunk ^ tsw(row)) << 4) + (pp & 1) * 8;  // (tsw(row + 4) = tsw(row): bit 2 is not in it)
            }
            bf16x8 xh[4], xl[4], yh[4], yl[4];
            // lockstep schedule; the fragment reads are ordered so that the first product (xl . yh) starts when HALF of them
            // have landed
            bar();
            for (int kt = 0; kt < nk; ++kt) {
                const lds_c* st = lds + (kt % NST) * STAGE;
                const lds_c* pa = st + (R * 64 + fr) * 64 + pos;
                const lds_c* pb = st + 2 * PA;
#pragma unroll
                for (int b = 0; b < 4; ++b) xl[b] = __builtin_bit_cast(bf16x8, *(const lds_u4*)(pa + PA + b * 1024));
#pragma unroll
                for (int j = 0; j < 4; ++j) yh[j] = tr_read_k8(pb + offB[j], 1024);
#pragma unroll
                for (int b = 0; b < 4; ++b) xh[b] = __builtin_bit_cast(bf16x8, *(const lds_u4*)(pa + b * 1024));
#pragma unroll
                for (int j = 0; j < 4; ++j) yl[j] = tr_read_k8(pb + PB + offB[j], 1024);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F | (12 << 8));  // lgkmcnt(12): xl (4) and yh (8 reads) are here
                __builtin_amdgcn_sched_barrier(0);
                mac(acc, xh, xl, yh, yl, true, false);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
                mac(acc, xh, xl, yh, yl, false, true);
                bar();
            }
        }
        __syncthreads();
    }

    // element r of block (i, j) of a consumer wave: row 64 R + 16 i + 4 (lane >> 4) + r, column 64 C + 16 j + (lane & 15)
    __device__ static __forceinline__ int sub_row(int i, int r) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave >> 1) * 64 + 16 * i + 4 * (lane >> 4) + r;
    }
    __device__ static __forceinline__ int sub_col(int j) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        return (wave & 1) * 64 + 16 * j + (lane & 15);
    }
};

}  // namespace vgan
