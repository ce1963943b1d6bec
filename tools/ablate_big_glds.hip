// Prototype of the 128x128 split-bf16 tile main loop with direct-to-LDS staging (global_load_lds_dwordx4), checked against
// GemmBF3Big::run on the same operands (tools only, not part of the library).
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../v-gan_amd/csrc/gemm_bf3.hpp"
using namespace vgan;
namespace vgan { void set_error(const char*, ...) {} }

template <int MODE, bool DO_FILL = true, bool DO_FRAGS = true, bool DO_MFMA = true>  // 0: fill, reads, MFMAs, one barrier; 1: the same with all fragment reads issued before the first MFMA; 2: ping-pong groups
struct BigGT {
    static constexpr int BK = 64, PART = 128 * 128, BUF = 4 * PART, kLdsBytes = 2 * BUF;
    typedef char __attribute__((address_space(3))) lds_c;
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x16 (&acc)[2]) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int grp = wave >> 2, R = (wave >> 1) & 1, C = wave & 1;
        const int fi = lane & 31, fh = lane >> 5;
        // staging: wave w fills part w >> 1 (Ah, Al, Bh, Bl), rows (w & 1) * 64 + 8 e + (lane >> 3), e = 0..7; the lane's
        // 16-byte chunk of the row is (lane & 7) ^ swz(row): the LDS image is lane-linear, the swizzle sits in the source
        const int part = wave >> 1;
        const unsigned short* base = part == 0 ? Ah : part == 1 ? Al : part == 2 ? Bh : Bl;
        const long ld = part < 2 ? lda : ldb;
        const int r0g = part < 2 ? m0 : n0, lim = part < 2 ? M : N;
        const char* src[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = (wave & 1) * 64 + 8 * e + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            src[e] = reinterpret_cast<const char*>(base + (long)min(r0g + row, lim - 1) * ld) + 16 * c;
        }
        const int dst0 = part * PART + (wave & 1) * 8192;  // + e * 1024
        auto fill = [&](int kt) {
            if (!DO_FILL && kt > 1) return;
            lds_c* d = lds + (kt & 1) * BUF + dst0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[e] + 2 * (long)kt * BK),
                                                 (void __attribute__((address_space(3)))*)(d + e * 1024), 16, 0, 0);
        };
        GemmBF3Big::Quad qd;
        qd.zero();
        const int nk = K / BK;
        fill(0);
        __syncthreads();
        const int sw = (fi >> 1) & 7;
        u32x4 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        auto frags = [&](int kt) {
            if (!DO_FRAGS && kt > 0) return;
            const lds_c* buf = lds + (kt & 1) * BUF;
            const lds_c* pa = buf + (R * 64 + fi) * 128;
            const lds_c* pb = buf + 2 * PART + (C * 64 + fi) * 128;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pos = ((2 * (2 * grp + s) + fh) ^ sw) << 4;
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
            if constexpr (!DO_MFMA) {
                qd.a[0][0][0] += __builtin_bit_cast(f32x4, ah[0][0] ^ al[0][1] ^ bh[0][0] ^ bl[0][1] ^ ah[1][0] ^ al[1][1] ^ bh[1][0] ^ bl[1][1])[0];
                qd.a[1][1][1] += __builtin_bit_cast(f32x4, ah[0][1] ^ al[0][0] ^ bh[0][1] ^ bl[0][0] ^ ah[1][1] ^ al[1][0] ^ bh[1][1] ^ bl[1][0])[1];
                return;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 xh[2] = {__builtin_bit_cast(bf16x8, ah[s][0]), __builtin_bit_cast(bf16x8, ah[s][1])};
                const bf16x8 xl[2] = {__builtin_bit_cast(bf16x8, al[s][0]), __builtin_bit_cast(bf16x8, al[s][1])};
                const bf16x8 yh[2] = {__builtin_bit_cast(bf16x8, bh[s][0]), __builtin_bit_cast(bf16x8, bh[s][1])};
                const bf16x8 yl[2] = {__builtin_bit_cast(bf16x8, bl[s][0]), __builtin_bit_cast(bf16x8, bl[s][1])};
                qd.mac(xh, xl, yh, yl);
            }
        };
        if constexpr (MODE < 2) {
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) fill(kt + 1);
                frags(kt);
                if constexpr (MODE == 1) __builtin_amdgcn_sched_barrier(0);
                mfmas();
                __syncthreads();
            }
        } else {
            // the two wave groups half an iteration apart: one of them on the MFMA pipe, the other on the LDS / VMEM side
            auto bar = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            auto drain = [&]() { __builtin_amdgcn_s_waitcnt(0x0070); };  // vmcnt(0) lgkmcnt(0)
            if (grp == 0) {
                for (int kt = 0; kt < nk; ++kt) {
                    if (kt + 1 < nk) fill(kt + 1);
                    frags(kt);
                    bar();
                    mfmas();
                    drain();
                    bar();
                }
                bar();
            } else {
                if (1 < nk) fill(1);
                bar();
                for (int kt = 0; kt < nk; ++kt) {
                    frags(kt);
                    drain();
                    bar();
                    if (kt + 2 < nk) fill(kt + 2);
                    mfmas();
                    bar();
                }
            }
            __syncthreads();
        }
        qd.exchange(lds, acc);
    }
};
typedef BigGT<0> BigG;

template <class G>
__global__ __launch_bounds__(512, 2) void k(const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int tiles_per_row, float* out, long long* stamps) {
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    const int r0 = (blockIdx.x / tiles_per_row) * 128, c0 = (blockIdx.x % tiles_per_row) * 128;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long c0s = clock64(), w0s = wall_clock64();
    G::run(Zh, Zl, kp, Zh, Zl, kp, r0, c0, N, N, kp, lds, acc);
    if (stamps != nullptr && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
        stamps[0] = clock64() - c0s;
        stamps[1] = wall_clock64() - w0s;
    }
    // every element is written: [tile][i][r][thread]
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) out[(((size_t)blockIdx.x * 2 + i) * 16 + r) * 512 + threadIdx.x] = acc[i][r];
}
struct BaseG {
    static constexpr int kLdsBytes = GemmBF3Big::kLdsBytes;
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds, f32x16 (&acc)[2]) {
        GemmBF3Big::run<false>(Ah, Al, lda, Bh, Bl, ldb, m0, n0, M, N, K, lds, acc);
    }
};
static long long* g_stamps = nullptr;
template <class G>
static double bench(const char* name, const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int ntiles, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tpr = N / 128;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<G>, dim3(ntiles), dim3(512), 0, 0, Zh, Zl, kp, N, tpr, out, g_stamps);
    hipEventRecord(e0);
    const int it = 5;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k<G>, dim3(ntiles), dim3(512), 0, 0, Zh, Zl, kp, N, tpr, out, g_stamps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 128 * 128 * kp * (double)ntiles, t = ms / it * 1e-3;
    long long hs[2] = {0, 0};
    hipMemcpy(hs, g_stamps, 16, hipMemcpyDeviceToHost);
    printf("[clock64/wall_clock64 over one tile: %.2f; tile %.1f us if wall is 100 MHz] ", (double)hs[0] / (double)hs[1], hs[1] / 100.0);
    printf("%-8s tiles=%d: %.3f ms/launch, %.0f algorithmic TFLOP/s, executed %.3f of 2.5 PF, %.0f cycles per K tile and CU at 2.4 GHz\n", name, ntiles,
           t * 1e3, fl / t / 1e12, 3.0 * fl / t / 2.5e15, t / ((double)ntiles * (kp / 64) / 256.0) * 2.4e9);
    return t;
}
int main(int argc, char** argv) {
    const int N = 8192, kp = 4096, tpr = N / 128, ntiles = argc > 1 ? atoi(argv[1]) : tpr * tpr;
    unsigned short *Zh, *Zl;
    float *o1, *o2;
    hipMalloc(&Zh, (size_t)N * kp * 2);
    hipMalloc(&Zl, (size_t)N * kp * 2);
    const size_t no = (size_t)ntiles * 32 * 512;
    hipMalloc(&o1, no * 4);
    hipMalloc(&o2, no * 4);
    hipMalloc(&g_stamps, 16);
    std::vector<unsigned short> h((size_t)N * kp);
    for (auto& v : h) v = 0x3F00 + rand() % 128 + ((rand() & 1) << 15);
    hipMemcpy(Zh, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (auto& v : h) v = 0x3B00 + rand() % 128;
    hipMemcpy(Zl, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    bench<BaseG>("base", Zh, Zl, kp, N, ntiles, o1);
    bench<BigG>("glds", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<1>>("glds_rd1", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<2>>("glds_pp", Zh, Zl, kp, N, ntiles, o2);
    std::vector<float> c(no);
    hipMemcpy(c.data(), o2, no * 4, hipMemcpyDeviceToHost);
    bench<BigGT<0, false, true, true>>("g0_nofill", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<0, true, false, true>>("g0_nofrag", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<0, true, true, false>>("g0_nomfma", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<0, false, false, true>>("g0_mfmaonly", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<2, false, true, true>>("pp_nofill", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<2, true, false, true>>("pp_nofrag", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<2, true, true, false>>("pp_nomfma", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<2, false, false, true>>("pp_mfmaonly", Zh, Zl, kp, N, ntiles, o2);
    bench<BaseG>("library", Zh, Zl, kp, N, ntiles, o2);
    bench<BigGT<2>>("glds_pp", Zh, Zl, kp, N, ntiles, o2);
    bench<BaseG>("library", Zh, Zl, kp, N, ntiles, o2);
    std::vector<float> a(no), b(no);
    hipMemcpy(a.data(), o1, no * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o2, no * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < no; ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); }
    double mp = 0;
    for (size_t i = 0; i < no; ++i) mp = fmax(mp, fabs((double)a[i] - c[i]));
    printf("max |base - glds| = %.3g, |base - glds_pp| = %.3g (max |value| %.3g)\n", md, mp, mx);
    return 0;
}
