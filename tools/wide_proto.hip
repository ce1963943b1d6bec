// Prototype (tools only): split-bf16 tile main loop on a 256 x 128 tile with a K stage of 32, THREE stages of direct-to-LDS
// fill in flight and v_mfma_f32_16x16x32_bf16 -- against GemmBF3Big::run (128 x 128, K tile 64, two stages) on the same warm
// operands.  Why: per flop the 128 x 128 tile moves 64 KB per K tile through the CU's ~68 GB/s intake (fill alone = 0.67 of the
// bf16 peak) with ONE tile of prefetch distance; 256 x 128 moves 3/4 of the bytes per flop, three 48 KB stages keep two stages
// (96 KB) in flight, and the 16x16x32 shape holds a higher clock under load (MI355X_MICROARCH.md, DVFS give-back item 7).
// Tile 0 of the new loop is checked against a host reference.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "v-gan_amd/csrc/gemm_bf3.hpp"
using namespace vgan;
namespace vgan { void set_error(const char*, ...) {} }

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool DO_FILL = true, bool DO_MFMA = true, bool DO_FRAGS = true>
struct WideT {
    static constexpr int BM = 256, BN = 128, BK = 32, NTH = 512, NST = 3;
    static constexpr int PA = BM * BK * 2, PB = BN * BK * 2;  // bytes of one A / B part of a stage (16 KB / 8 KB)
    static constexpr int STAGE = 2 * PA + 2 * PB;             // Ah | Al | Bh | Bl = 48 KB
    static constexpr int kLdsBytes = NST * STAGE;             // 144 KB
    typedef char __attribute__((address_space(3))) lds_c;
    // LDS image of a part: rows of 64 B (4 chunks of 16 B = 8 k each); the chunk c of row r sits at position c ^ h(r),
    // h(r) = (-(r >> 2)) & 3: a ds_read_b128 of a 16-row fragment block (lane = row + 16 * chunk) is then conflict-free in each
    // of the instruction's four lane groups.  A 1-KB fill piece = 16 rows, lane l -> row l >> 2, position l & 3.
    __device__ static __forceinline__ int hsw(int r) { return (-(r >> 2)) & 3; }

    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x4v (&acc)[4][4]) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int R = wave >> 1, C = wave & 1;
        // fill: 48 pieces per stage, 6 per wave: pieces 6 w .. 6 w + 5 of [Ah: 16 | Al: 16 | Bh: 8 | Bl: 8]
        const char* src[6];
        int dst[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int pc = 6 * wave + e;
            const int part = pc < 16 ? 0 : pc < 32 ? 1 : pc < 40 ? 2 : 3;
            const int pin = part == 0 ? pc : part == 1 ? pc - 16 : part == 2 ? pc - 32 : pc - 40;  // piece inside the part
            const int row = 16 * pin + (lane >> 2);
            const int c = (lane & 3) ^ hsw(row);
            const unsigned short* base = part == 0 ? Ah : part == 1 ? Al : part == 2 ? Bh : Bl;
            const long ld = part < 2 ? lda : ldb;
            const int g0 = part < 2 ? m0 : n0, lim = part < 2 ? M : N;
            src[e] = reinterpret_cast<const char*>(base + (long)min(g0 + row, lim - 1) * ld) + 16 * c;
            dst[e] = (part == 0 ? 0 : part == 1 ? PA : part == 2 ? 2 * PA : 2 * PA + PB) + pin * 1024;
        }
        auto fill = [&](int kt) {
            lds_c* d = lds + (kt % NST) * STAGE;
#pragma unroll
            for (int e = 0; e < 6; ++e)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[e] + 2 * (long)kt * BK),
                                                 (void __attribute__((address_space(3)))*)(d + dst[e]), 16, 0, 0);
        };
        const int nk = K / BK;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        fill(0);
        if (nk > 1) fill(1);
        if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 6);  // vmcnt(6): stage 0 has landed (stage 1 may be in flight)
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        // fragment addresses: 16-row block b of the wave's 64 rows: row 64 R + 16 b + (lane & 15), chunk lane >> 4
        const int fr = lane & 15, fc = lane >> 4;
        const int posA = (fc ^ hsw(fr)) << 4;  // (rows 16 b + fr: (row >> 2) & 3 = (fr >> 2) & 3 for every b)
        u32x4 ah[4], al[4], bh[4], bl[4];
        for (int kt = 0; kt < nk; ++kt) {
            if (DO_FILL && kt + 2 < nk) fill(kt + 2);
            const lds_c* st = lds + (kt % NST) * STAGE;
            const lds_c* pa = st + (R * 64 + fr) * 64 + posA;
            const lds_c* pb = st + 2 * PA + (C * 64 + fr) * 64 + posA;
            if (DO_FRAGS || kt == 0)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                ah[b] = *(const lds_u4*)(pa + b * 1024);
                al[b] = *(const lds_u4*)(pa + PA + b * 1024);
                bh[b] = *(const lds_u4*)(pb + b * 1024);
                bl[b] = *(const lds_u4*)(pb + PB + b * 1024);
            }
            if (!DO_MFMA) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][0][0] += __builtin_bit_cast(f32x4v, ah[i] ^ al[i] ^ bh[i] ^ bl[i])[0];
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
            }
            // stage kt + 1 must have landed before anyone reads it; stage kt + 2 (this wave's 6 youngest) may stay in flight
            if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | 6);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            // a RAW barrier: __syncthreads() is a workgroup-scope fence and drains vmcnt(0), i.e. the stage in flight
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // acc[i][j][r]: row 64 R + 16 i + 4 (lane >> 4) + r, column 64 C + 16 j + (lane & 15)
};
typedef WideT<> Wide;

template <class G>
__global__ __launch_bounds__(512, 2) void k_wide(const unsigned short* Zh, const unsigned short* Zl, int kp, int ld, int N, int tiles_per_row, float* out,
                                                float* tile0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // XCD-aware order: block b runs on XCD b % 8; each XCD walks its own contiguous chunk of 4 x 8 tile blocks, so the 32 tiles
    // resident on an XCD at a time share 4 row panels and 8 column panels
    const int per = gridDim.x / 8, t = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int blk = t / 32, in = t % 32, bpr = tiles_per_row / 8;
    const int r0 = ((blk / bpr) * 4 + in / 8) * 256, c0 = ((blk % bpr) * 8 + in % 8) * 128;
    f32x4v acc[4][4];
    G::run(Zh, Zl, ld, Zh, Zl, ld, r0, c0, N, N, kp, lds, acc);
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (r0 == 0 && c0 == 128 && tile0 != nullptr) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, R = wave >> 1, C = wave & 1;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r)
            tile0[(64 * R + 16 * i + 4 * (lane >> 4) + r) * 128 + 64 * C + 16 * j + (lane & 15)] = acc[i][j][r];
    }
}


// Variant with DEDICATED loader waves: 768 threads = 8 consumer waves (64 x 64 quadrants, as above, no fill) + 4 loader waves
// that issue the whole fill (12 pieces each per stage) and own the vmcnt waits; one raw barrier per stage for all twelve.
struct WideLW {
    static constexpr int BM = 256, BN = 128, BK = 32, NTH = 768, NST = 3;
    static constexpr int PA = BM * BK * 2, PB = BN * BK * 2, STAGE = 2 * PA + 2 * PB, kLdsBytes = NST * STAGE;
    typedef char __attribute__((address_space(3))) lds_c;
    __device__ static __forceinline__ int hsw(int r) { return (-(r >> 2)) & 3; }
    __device__ static __forceinline__ void bar() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x4v (&acc)[4][4]) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nk = K / BK;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
#ifdef LW_PRIO
        if (wave >= 8) __builtin_amdgcn_s_setprio(3);
#endif
        if (wave >= 8) {  // ---- loader wave lw: pieces 12 lw .. 12 lw + 11 of [Ah: 16 | Al: 16 | Bh: 8 | Bl: 8]
            const int lw = wave - 8;
            const char* src[12];
            int dst[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int pc = 12 * lw + e;
                const int part = pc < 16 ? 0 : pc < 32 ? 1 : pc < 40 ? 2 : 3;
                const int pin = part == 0 ? pc : part == 1 ? pc - 16 : part == 2 ? pc - 32 : pc - 40;
                const int row = 16 * pin + (lane >> 2);
                const int c = (lane & 3) ^ hsw(row);
                const unsigned short* base = part == 0 ? Ah : part == 1 ? Al : part == 2 ? Bh : Bl;
                const long ld = part < 2 ? lda : ldb;
                const int g0 = part < 2 ? m0 : n0, lim = part < 2 ? M : N;
                src[e] = reinterpret_cast<const char*>(base + (long)min(g0 + row, lim - 1) * ld) + 16 * c;
                dst[e] = (part == 0 ? 0 : part == 1 ? PA : part == 2 ? 2 * PA : 2 * PA + PB) + pin * 1024;
            }
            auto fill = [&](int kt) {
                lds_c* d = lds + (kt % NST) * STAGE;
#pragma unroll
                for (int e = 0; e < 12; ++e)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[e] + 2 * (long)kt * BK),
                                                     (void __attribute__((address_space(3)))*)(d + dst[e]), 16, 0, 0);
            };
            fill(0);
            if (nk > 1) fill(1);
            if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 12); else __builtin_amdgcn_s_waitcnt(0x0F70);
            bar();
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 2 < nk) fill(kt + 2);
                if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | 12); else __builtin_amdgcn_s_waitcnt(0x0F70);
                bar();
            }
            return;
        }
        const int R = wave >> 1, C = wave & 1;
        const int fr = lane & 15, fc = lane >> 4;
        const int posA = (fc ^ hsw(fr)) << 4;
        u32x4 ah[4], al[4], bh[4], bl[4];
        bar();
        for (int kt = 0; kt < nk; ++kt) {
            const lds_c* st = lds + (kt % NST) * STAGE;
            const lds_c* pa = st + (R * 64 + fr) * 64 + posA;
            const lds_c* pb = st + 2 * PA + (C * 64 + fr) * 64 + posA;
#ifdef LW_ORDERED
#pragma unroll
            for (int b = 0; b < 4; ++b) al[b] = *(const lds_u4*)(pa + PA + b * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) bh[b] = *(const lds_u4*)(pb + b * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) ah[b] = *(const lds_u4*)(pa + b * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) bl[b] = *(const lds_u4*)(pb + PB + b * 1024);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F | (8 << 8));  // lgkmcnt(8): al, bh have landed
            __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                ah[b] = *(const lds_u4*)(pa + b * 1024);
                al[b] = *(const lds_u4*)(pa + PA + b * 1024);
                bh[b] = *(const lds_u4*)(pb + b * 1024);
                bl[b] = *(const lds_u4*)(pb + PB + b * 1024);
            }
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
#ifdef LW_ORDERED
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
            bar();
        }
    }
};

__global__ __launch_bounds__(768, 3) void k_lw(const unsigned short* Zh, const unsigned short* Zl, int kp, int ld, int N, int tiles_per_row, float* out, float* tile0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int per = gridDim.x / 8, t = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int blk = t / 32, in = t % 32, bpr = tiles_per_row / 8;
    const int r0 = ((blk / bpr) * 4 + in / 8) * 256, c0 = ((blk % bpr) * 8 + in % 8) * 128;
    f32x4v acc[4][4];
    WideLW::run(Zh, Zl, ld, Zh, Zl, ld, r0, c0, N, N, kp, lds, acc);
    if (threadIdx.x >= 512) return;
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (r0 == 0 && c0 == 128 && tile0 != nullptr) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, R = wave >> 1, C = wave & 1;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r)
            tile0[(64 * R + 16 * i + 4 * (lane >> 4) + r) * 128 + 64 * C + 16 * j + (lane & 15)] = acc[i][j][r];
    }
}

// Variant: loader waves + the two consumer groups (waves 0-3, waves 4-7: one wave of each per SIMD) HALF A STAGE APART, two raw
// barriers per stage: while one group multiplies stage kt the other reads its fragments of stage kt (or kt + 1), so the SIMD's
// matrix pipe always has exactly one wave on it.
struct WidePP {
    static constexpr int BM = 256, BN = 128, BK = 32, NTH = 768, NST = 3;
    static constexpr int PA = BM * BK * 2, PB = BN * BK * 2, STAGE = 2 * PA + 2 * PB, kLdsBytes = NST * STAGE;
    typedef char __attribute__((address_space(3))) lds_c;
    __device__ static __forceinline__ int hsw(int r) { return (-(r >> 2)) & 3; }
    __device__ static __forceinline__ void bar() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x4v (&acc)[4][4]) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nk = K / BK;  // (>= 3 assumed in this prototype)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (wave >= 8) {
            __builtin_amdgcn_s_setprio(3);
            const int lw = wave - 8;
            const char* src[12];
            int dst[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int pc = 12 * lw + e;
                const int part = pc < 16 ? 0 : pc < 32 ? 1 : pc < 40 ? 2 : 3;
                const int pin = part == 0 ? pc : part == 1 ? pc - 16 : part == 2 ? pc - 32 : pc - 40;
                const int row = 16 * pin + (lane >> 2);
                const int c = (lane & 3) ^ hsw(row);
                const unsigned short* base = part == 0 ? Ah : part == 1 ? Al : part == 2 ? Bh : Bl;
                const long ld = part < 2 ? lda : ldb;
                const int g0 = part < 2 ? m0 : n0, lim = part < 2 ? M : N;
                src[e] = reinterpret_cast<const char*>(base + (long)min(g0 + row, lim - 1) * ld) + 16 * c;
                dst[e] = (part == 0 ? 0 : part == 1 ? PA : part == 2 ? 2 * PA : 2 * PA + PB) + pin * 1024;
            }
            auto fill = [&](int kt) {
                lds_c* d = lds + (kt % NST) * STAGE;
#pragma unroll
                for (int e = 0; e < 12; ++e)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[e] + 2 * (long)kt * BK),
                                                     (void __attribute__((address_space(3)))*)(d + dst[e]), 16, 0, 0);
            };
            fill(0); fill(1); fill(2);
            __builtin_amdgcn_s_waitcnt(0x0F70 | (24 & 15) | ((24 >> 4) << 14));  // vmcnt(24): stage 0 landed
            bar();  // P
            // barriers B(0) .. B(2 nk): before an odd B(j), stage (j + 1) / 2 must have landed; behind it, the buffer of stage
            // (j - 1) / 2 is free (both groups have read it): stage (j + 5) / 2 goes there
            for (int j = 0; j <= 2 * nk; ++j) {
                if (j & 1) {
                    const int need = (j + 1) / 2, next = (j + 5) / 2;
#ifndef PP_NOWAIT
                    if (need < nk) {
                        if (need + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | 12); else __builtin_amdgcn_s_waitcnt(0x0F70);
                    }
#endif
                    bar();
                    if (next < nk) fill(next);
                } else {
                    bar();
                }
            }
            return;
        }
        const int grp = wave >> 2, R = wave >> 1, C = wave & 1;
        const int fr = lane & 15, fc = lane >> 4;
        const int posA = (fc ^ hsw(fr)) << 4;
        u32x4 ah[4], al[4], bh[4], bl[4];
        auto frags = [&](int kt) {
            const lds_c* st = lds + (kt % NST) * STAGE;
            const lds_c* pa = st + (R * 64 + fr) * 64 + posA;
            const lds_c* pb = st + 2 * PA + (C * 64 + fr) * 64 + posA;
#pragma unroll
            for (int b = 0; b < 4; ++b) al[b] = *(const lds_u4*)(pa + PA + b * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) bh[b] = *(const lds_u4*)(pb + b * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) ah[b] = *(const lds_u4*)(pa + b * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) bl[b] = *(const lds_u4*)(pb + PB + b * 1024);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the reads are done before the barrier that frees the buffer
        };
        auto mfmas = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
        };
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
};

__global__ __launch_bounds__(768, 3) void k_pp(const unsigned short* Zh, const unsigned short* Zl, int kp, int ld, int N, int tiles_per_row, float* out, float* tile0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int per = gridDim.x / 8, t = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int blk = t / 32, in = t % 32, bpr = tiles_per_row / 8;
    const int r0 = ((blk / bpr) * 4 + in / 8) * 256, c0 = ((blk % bpr) * 8 + in % 8) * 128;
    f32x4v acc[4][4];
    WidePP::run(Zh, Zl, ld, Zh, Zl, ld, r0, c0, N, N, kp, lds, acc);
    if (threadIdx.x >= 512) return;
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (r0 == 0 && c0 == 128 && tile0 != nullptr) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, R = wave >> 1, C = wave & 1;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r)
            tile0[(64 * R + 16 * i + 4 * (lane >> 4) + r) * 128 + 64 * C + 16 * j + (lane & 15)] = acc[i][j][r];
    }
}

// Variant: ONE consumer wave per SIMD holding a 128 x 64 quadrant (8 x 4 blocks of 16 x 16: 128 accumulator registers), 4 loader
// waves, ONE barrier per stage.  A stage's 24 fragment reads (B first, then the A blocks in the order they are used) are all issued
// at its head -- behind the last MFMAs of the stage before -- and the MFMAs of row block i start when ITS two reads have landed
// (counted lgkmcnt); the barrier that hands the buffer back sits behind the last read, in front of the last 12 MFMAs.  Per stage
// and CU: 96 KB of fragment reads instead of 128, no read phase that the matrix pipe waits out.
struct WideQ {
    static constexpr int BM = 256, BN = 128, BK = 32, NTH = 512, NST = 3;
    static constexpr int PA = BM * BK * 2, PB = BN * BK * 2, STAGE = 2 * PA + 2 * PB, kLdsBytes = NST * STAGE;
    typedef char __attribute__((address_space(3))) lds_c;
    __device__ static __forceinline__ int hsw(int r) { return (-(r >> 2)) & 3; }
    __device__ static __forceinline__ void bar() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int N>
    __device__ static __forceinline__ void wait_lgkm() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x4v (&acc)[8][4]) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nk = K / BK;  // (>= 3 assumed in this prototype)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (wave >= 4) {
            __builtin_amdgcn_s_setprio(3);
            const int lw = wave - 4;
            const char* src[12];
            int dst[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int pc = 12 * lw + e;
                const int part = pc < 16 ? 0 : pc < 32 ? 1 : pc < 40 ? 2 : 3;
                const int pin = part == 0 ? pc : part == 1 ? pc - 16 : part == 2 ? pc - 32 : pc - 40;
                const int row = 16 * pin + (lane >> 2);
                const int c = (lane & 3) ^ hsw(row);
                const unsigned short* base = part == 0 ? Ah : part == 1 ? Al : part == 2 ? Bh : Bl;
                const long ld = part < 2 ? lda : ldb;
                const int g0 = part < 2 ? m0 : n0, lim = part < 2 ? M : N;
                src[e] = reinterpret_cast<const char*>(base + (long)min(g0 + row, lim - 1) * ld) + 16 * c;
                dst[e] = (part == 0 ? 0 : part == 1 ? PA : part == 2 ? 2 * PA : 2 * PA + PB) + pin * 1024;
            }
            auto fill = [&](int kt) {
                lds_c* d = lds + (kt % NST) * STAGE;
#pragma unroll
                for (int e = 0; e < 12; ++e)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[e] + 2 * (long)kt * BK),
                                                     (void __attribute__((address_space(3)))*)(d + dst[e]), 16, 0, 0);
            };
            fill(0); fill(1); fill(2);
            __builtin_amdgcn_s_waitcnt(0x0F70 | (24 & 15) | ((24 >> 4) << 14));  // vmcnt(24): stage 0 landed
            bar();  // P
            for (int kt = 0; kt < nk; ++kt) {  // B(kt): everyone has read stage kt; stage kt + 1 has landed
                if (kt + 1 < nk) {
                    if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | 12); else __builtin_amdgcn_s_waitcnt(0x0F70);
                }
                bar();
                if (kt + 3 < nk) fill(kt + 3);
            }
            return;
        }
        const int R = wave >> 1, C = wave & 1;
        const int fr = lane & 15, fc = lane >> 4;
        const int posA = (fc ^ hsw(fr)) << 4;
        u32x4 ah[8], al[8], bh[4], bl[4];
        bar();  // P
        for (int kt = 0; kt < nk; ++kt) {
            const lds_c* st = lds + (kt % NST) * STAGE;
            const lds_c* pa = st + (R * 128 + fr) * 64 + posA;
            const lds_c* pb = st + 2 * PA + (C * 64 + fr) * 64 + posA;
#pragma unroll
            for (int b = 0; b < 4; ++b) bh[b] = *(const lds_u4*)(pb + b * 1024);
#pragma unroll
            for (int b = 0; b < 4; ++b) bl[b] = *(const lds_u4*)(pb + PB + b * 1024);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                al[b] = *(const lds_u4*)(pa + PA + b * 1024);
                ah[b] = *(const lds_u4*)(pa + b * 1024);
            }
            auto group = [&](int i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
            };
            wait_lgkm<14>(); group(0);
            wait_lgkm<12>(); group(1);
            wait_lgkm<10>(); group(2);
            wait_lgkm<8>(); group(3);
            wait_lgkm<6>(); group(4);
            wait_lgkm<4>(); group(5);
            wait_lgkm<2>(); group(6);
            wait_lgkm<0>();
            bar();       // B(kt): the buffer goes back to the loaders
            group(7);
        }
    }
};

__global__ __launch_bounds__(512, 2) void k_q(const unsigned short* Zh, const unsigned short* Zl, int kp, int ld, int N, int tiles_per_row, float* out, float* tile0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int per = gridDim.x / 8, t = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int blk = t / 32, in = t % 32, bpr = tiles_per_row / 8;
    const int r0 = ((blk / bpr) * 4 + in / 8) * 256, c0 = ((blk % bpr) * 8 + in % 8) * 128;
    f32x4v acc[8][4];
    WideQ::run(Zh, Zl, ld, Zh, Zl, ld, r0, c0, N, N, kp, lds, acc);
    if (threadIdx.x >= 256) return;
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (r0 == 0 && c0 == 128 && tile0 != nullptr) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, R = wave >> 1, C = wave & 1;
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r)
            tile0[(128 * R + 16 * i + 4 * (lane >> 4) + r) * 128 + 64 * C + 16 * j + (lane & 15)] = acc[i][j][r];
    }
}

__global__ __launch_bounds__(512, 2) void k_big(const unsigned short* Zh, const unsigned short* Zl, int kp, int ld, int N, int tiles_per_row, float* out) {
    using G = GemmBF3Big;
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    const int per = gridDim.x / 8, t = (blockIdx.x % 8) * per + blockIdx.x / 8;
    const int blk = t / 32, in = t % 32, bpr = tiles_per_row / 8;
    const int r0 = ((blk / bpr) * 4 + in / 8) * 128, c0 = ((blk % bpr) * 8 + in % 8) * 128;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    G::run<false>(Zh, Zl, ld, Zh, Zl, ld, r0, c0, N, N, kp, lds, acc);
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

static float bf(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8192, kp = argc > 2 ? atoi(argv[2]) : 4096, pad = argc > 3 ? atoi(argv[3]) : 0, ld = kp + pad;
    const int tprW = N / 128, ntW = (N / 256) * tprW, tprB = N / 128, ntB = tprB * tprB;
    unsigned short *Zh, *Zl; float *out, *tile0;
    hipMalloc(&Zh, (size_t)N * ld * 2); hipMalloc(&Zl, (size_t)N * ld * 2); hipMalloc(&out, (size_t)ntB * 512 * 4); hipMalloc(&tile0, 256 * 128 * 4);
    std::vector<unsigned short> hh((size_t)N * ld), hl((size_t)N * ld);
    for (auto& v : hh) v = 0x3F00 + rand() % 128 + ((rand() & 1) << 15);
    for (auto& v : hl) v = 0x3B00 + rand() % 128;
    hipMemcpy(Zh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(Zl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide<Wide>), hipFuncAttributeMaxDynamicSharedMemorySize, Wide::kLdsBytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw), hipFuncAttributeMaxDynamicSharedMemorySize, Wide::kLdsBytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pp), hipFuncAttributeMaxDynamicSharedMemorySize, Wide::kLdsBytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_q), hipFuncAttributeMaxDynamicSharedMemorySize, Wide::kLdsBytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide<WideT<false, true, true>>), hipFuncAttributeMaxDynamicSharedMemorySize, Wide::kLdsBytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide<WideT<true, false, true>>), hipFuncAttributeMaxDynamicSharedMemorySize, Wide::kLdsBytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide<WideT<false, true, false>>), hipFuncAttributeMaxDynamicSharedMemorySize, Wide::kLdsBytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide<WideT<true, false, false>>), hipFuncAttributeMaxDynamicSharedMemorySize, Wide::kLdsBytes);
    // correctness of tile 1 (rows 0..255, columns 128..255)
    hipLaunchKernelGGL(k_wide<Wide>, dim3(ntW), dim3(512), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, tile0);
    std::vector<float> t0(256 * 128);
    hipMemcpy(t0.data(), tile0, t0.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, big = 0;
    for (int i = 0; i < 256; i += 37)
        for (int j = 0; j < 128; j += 11) {
            double s = 0;
            for (int k = 0; k < kp; ++k) {
                const double ah = bf(hh[(size_t)i * ld + k]), al = bf(hl[(size_t)i * ld + k]);
                const double bh = bf(hh[(size_t)(128 + j) * ld + k]), bl = bf(hl[(size_t)(128 + j) * ld + k]);
                s += al * bh + ah * bl + ah * bh;
            }
            worst = fmax(worst, fabs(s - t0[i * 128 + j]));
            big = fmax(big, fabs(s));
        }
    printf("wide tile check: max |err| %.3e (largest |value| %.3e) %s\n", worst, big, worst <= 2e-5 * big + 1e-3 ? "OK" : "MISMATCH");
    hipMemset(tile0, 0, 256 * 128 * 4);
    hipLaunchKernelGGL(k_lw, dim3(ntW), dim3(768), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, tile0);
    std::vector<float> t1(256 * 128);
    hipMemcpy(t1.data(), tile0, t1.size() * 4, hipMemcpyDeviceToHost);
    printf("loader-wave variant equals the plain wide loop bit for bit: %s\n", memcmp(t0.data(), t1.data(), t0.size() * 4) == 0 ? "yes" : "NO");
    hipMemset(tile0, 0, 256 * 128 * 4);
    hipLaunchKernelGGL(k_pp, dim3(ntW), dim3(768), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, tile0);
    hipMemcpy(t1.data(), tile0, t1.size() * 4, hipMemcpyDeviceToHost);
    printf("staggered-group variant equals the plain wide loop bit for bit: %s\n", memcmp(t0.data(), t1.data(), t0.size() * 4) == 0 ? "yes" : "NO");
    hipMemset(tile0, 0, 256 * 128 * 4);
    hipLaunchKernelGGL(k_q, dim3(ntW), dim3(512), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, tile0);
    hipMemcpy(t1.data(), tile0, t1.size() * 4, hipMemcpyDeviceToHost);
    printf("one-consumer-wave-per-SIMD variant equals the plain wide loop bit for bit: %s\n", memcmp(t0.data(), t1.data(), t0.size() * 4) == 0 ? "yes" : "NO");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const char* names[] = {"big 128x128x64 (2 st)", "wide 256x128x32 (3 st)", "wide, no fill", "wide, no MFMA", "wide, MFMA only", "wide, fill only", "wide + 4 loader waves", "loaders + staggered groups", "4 x (128 x 64) consumers"};
        for (int which = 0; which < 9; ++which) {
            auto launch = [&]() {
                float* nul = nullptr;
                if (which == 0) hipLaunchKernelGGL(k_big, dim3(ntB), dim3(512), 0, 0, Zh, Zl, kp, ld, N, tprB, out);
                else if (which == 1) hipLaunchKernelGGL(k_wide<Wide>, dim3(ntW), dim3(512), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, nul);
                else if (which == 2) hipLaunchKernelGGL((k_wide<WideT<false, true, true>>), dim3(ntW), dim3(512), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, nul);
                else if (which == 3) hipLaunchKernelGGL((k_wide<WideT<true, false, true>>), dim3(ntW), dim3(512), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, nul);
                else if (which == 4) hipLaunchKernelGGL((k_wide<WideT<false, true, false>>), dim3(ntW), dim3(512), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, nul);
                else if (which == 5) hipLaunchKernelGGL((k_wide<WideT<true, false, false>>), dim3(ntW), dim3(512), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, nul);
                else if (which == 6) hipLaunchKernelGGL(k_lw, dim3(ntW), dim3(768), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, nul);
                else if (which == 7) hipLaunchKernelGGL(k_pp, dim3(ntW), dim3(768), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, nul);
                else hipLaunchKernelGGL(k_q, dim3(ntW), dim3(512), Wide::kLdsBytes, 0, Zh, Zl, kp, ld, N, tprW, out, nul);
            };
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e0);
            const int it = 10;
            for (int i = 0; i < it; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2.0 * N * (double)N * kp, t = ms / it * 1e-3;
            printf("%-22s %.3f ms/launch, executed %.3f of 2.5 PF\n", names[which], t * 1e3, 3.0 * fl / t / 2.5e15);
        }
    }
    return 0;
}
