// Prototype (tools only): the loader-wave / multi-stage idea of csrc/gemm_bf3w.hpp on the 64 x 64 tiles of the c3 kernels.
// GemmBF3<64> (register staging, K tile 64, two buffers, two 256-thread workgroups per CU) against "mini-wide" loops: K stage 32
// (16 KB: 64-byte rows, hsw swizzle), NST stages in LDS, 4 consumer waves (a 32 x 32 quadrant each as 2 x 2 blocks of
// v_mfma_f32_16x16x32_bf16) + NLW loader waves that issue the direct-to-LDS fill and own the vmcnt waits, raw s_barrier per
// stage.  Shapes of the c3 Gram: K = 832, 392 / 512 / 528 tiles.  (Round 2's direct-to-LDS prototype, tools/ablate_bf3_glds.hip,
// kept __syncthreads(), whose fence drains vmcnt(0): it could not show what staging in flight is worth.)
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "v-gan_amd/csrc/gemm_bf3.hpp"
using namespace vgan;
namespace vgan { void set_error(const char*, ...) {} }

template <int NST, int NLW>
struct Mini {
    static constexpr int BK = 32, NCW = 4, NTH = 64 * (NCW + NLW);
    static constexpr int PART = 64 * BK * 2;           // 4 KB: 64 rows of 64 B
    static constexpr int STAGE = 4 * PART;             // Ah | Al | Bh | Bl = 16 KB
    static constexpr int kLdsBytes = NST * STAGE;
    static constexpr int NPL = 16 / NLW;               // pieces per loader wave and stage
    typedef char __attribute__((address_space(3))) lds_c;
    __device__ static __forceinline__ int hsw(int r) { return (-(r >> 2)) & 3; }
    __device__ static __forceinline__ void bar() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int N>
    __device__ static __forceinline__ void wait_vm() {  // s_waitcnt vmcnt(N), nothing else
        __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
    }
    // acc[i][j][r]: row 32 R + 16 i + 4 (lane >> 4) + r, column 32 C + 16 j + (lane & 15)
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x4 (&acc)[2][2]) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nk = K / BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave >= NCW) {
            __builtin_amdgcn_s_setprio(3);
            const int lw = wave - NCW;
            const char* src[NPL];
            int dst[NPL];
#pragma unroll
            for (int e = 0; e < NPL; ++e) {
                const int pc = NPL * lw + e, part = pc >> 2, pin = pc & 3;
                const int row = 16 * pin + (lane >> 2);
                const int c = (lane & 3) ^ hsw(row);
                const unsigned short* base = part == 0 ? Ah : part == 1 ? Al : part == 2 ? Bh : Bl;
                const long ld = part < 2 ? lda : ldb;
                const int g0 = part < 2 ? m0 : n0, lim = part < 2 ? M : N;
                src[e] = reinterpret_cast<const char*>(base + (long)min(g0 + row, lim - 1) * ld) + 16 * c;
                dst[e] = part * PART + pin * 1024;
            }
            auto fill = [&](int kt) {
                lds_c* d = lds + (kt % NST) * STAGE;
#pragma unroll
                for (int e = 0; e < NPL; ++e)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[e] + 2 * (long)kt * BK),
                                                     (void __attribute__((address_space(3)))*)(d + dst[e]), 16, 0, 0);
            };
            // NST - 1 stages ahead: the prologue issues stages 0 .. NST - 2; iteration kt issues stage kt + NST - 1 and waits for
            // stage kt + 1 (all but the NST - 2 youngest stages' pieces)
#pragma unroll
            for (int s = 0; s < NST - 1; ++s)
                if (s < nk) fill(s);
            if (nk >= NST - 1) wait_vm<(NST - 2) * NPL>(); else wait_vm<0>();
            bar();
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + NST - 1 < nk) { fill(kt + NST - 1); wait_vm<(NST - 2) * NPL>(); }
                else wait_vm<0>();
                bar();
            }
            return;
        }
        const int R = wave >> 1, C = wave & 1;
        const int fr = lane & 15, fc = lane >> 4;
        const int pos = (fc ^ hsw(fr)) << 4;
        u32x4 ah[2], al[2], bh[2], bl[2];
        bar();
        for (int kt = 0; kt < nk; ++kt) {
            const lds_c* st = lds + (kt % NST) * STAGE;
            const lds_c* pa = st + (R * 32 + fr) * 64 + pos;
            const lds_c* pb = st + 2 * PART + (C * 32 + fr) * 64 + pos;
#pragma unroll
            for (int b = 0; b < 2; ++b) al[b] = *(const lds_u4*)(pa + PART + b * 1024);
#pragma unroll
            for (int b = 0; b < 2; ++b) bh[b] = *(const lds_u4*)(pb + b * 1024);
#pragma unroll
            for (int b = 0; b < 2; ++b) ah[b] = *(const lds_u4*)(pa + b * 1024);
#pragma unroll
            for (int b = 0; b < 2; ++b) bl[b] = *(const lds_u4*)(pb + PART + b * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
            bar();
        }
    }
};

template <class G>
__global__ __launch_bounds__(G::NTH, 2) void k_mini(const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int tiles_per_row, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    const int r0 = (blockIdx.x / tiles_per_row) * 64, c0 = (blockIdx.x % tiles_per_row) * 64;
    f32x4 acc[2][2];
    G::run(Zh, Zl, kp, Zh, Zl, kp, r0, c0, N, N, kp, lds, acc);
    if (threadIdx.x >= 256) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, R = wave >> 1, C = wave & 1;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r)
        out[(size_t)blockIdx.x * 4096 + (32 * R + 16 * i + 4 * (lane >> 4) + r) * 64 + 32 * C + 16 * j + (lane & 15)] = acc[i][j][r];
}
__global__ __launch_bounds__(256, 2) void k_base(const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int tiles_per_row, float* out) {
    using G = GemmBF3<64>;
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    const int r0 = (blockIdx.x / tiles_per_row) * 64, c0 = (blockIdx.x % tiles_per_row) * 64;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    G::run<false>(Zh, Zl, kp, Zh, Zl, kp, r0, c0, N, N, kp, lds, nullptr, acc);
    for (int r = 0; r < 16; ++r) out[(size_t)blockIdx.x * 4096 + G::sub_row(r) * 64 + G::sub_col()] = acc[r];
}

template <class F>
static float time_us(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e0);
    const int it = 50;
    for (int i = 0; i < it; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / it * 1e3f;
}

int main() {
    const int N = 2048, kp = 832, tpr = N / 64;
    unsigned short *Zh, *Zl; float *o1, *o2;
    hipMalloc(&Zh, (size_t)N * kp * 2); hipMalloc(&Zl, (size_t)N * kp * 2);
    const size_t no = (size_t)528 * 4096;
    hipMalloc(&o1, no * 4); hipMalloc(&o2, no * 4);
    std::vector<unsigned short> h((size_t)N * kp);
    for (auto& v : h) v = 0x3F00 + rand() % 128 + ((rand() & 1) << 15);
    hipMemcpy(Zh, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (auto& v : h) v = 0x3B00 + rand() % 128;
    hipMemcpy(Zl, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int nt : {392, 512, 528}) {
            const double fl = 2.0 * 64 * 64 * kp * nt;
            float t0 = time_us([&] { hipLaunchKernelGGL(k_base, dim3(nt), dim3(256), 0, 0, Zh, Zl, kp, N, tpr, o1); });
            float t1 = time_us([&] { hipLaunchKernelGGL((k_mini<Mini<4, 1>>), dim3(nt), dim3(Mini<4, 1>::NTH), 0, 0, Zh, Zl, kp, N, tpr, o2); });
            float t2 = time_us([&] { hipLaunchKernelGGL((k_mini<Mini<4, 2>>), dim3(nt), dim3(Mini<4, 2>::NTH), 0, 0, Zh, Zl, kp, N, tpr, o2); });
            float t3 = time_us([&] { hipLaunchKernelGGL((k_mini<Mini<3, 1>>), dim3(nt), dim3(Mini<3, 1>::NTH), 0, 0, Zh, Zl, kp, N, tpr, o2); });
            printf("tiles=%d: GemmBF3<64> %.1f us (%.0f TF/s alg) | mini 4 stages + 1 loader %.1f | 4 stages + 2 loaders %.1f | 3 stages + 1 loader %.1f\n", nt, t0,
                   fl / t0 / 1e6, t1, t2, t3);
        }
    hipLaunchKernelGGL(k_base, dim3(512), dim3(256), 0, 0, Zh, Zl, kp, N, tpr, o1);
    hipLaunchKernelGGL((k_mini<Mini<4, 1>>), dim3(512), dim3(Mini<4, 1>::NTH), 0, 0, Zh, Zl, kp, N, tpr, o2);
    std::vector<float> a(512 * 4096), b(512 * 4096);
    hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); }
    printf("max |base - mini| = %.3g (largest |value| %.3g) %s\n", md, mx, md <= 2e-5 * mx ? "OK" : "MISMATCH");
    return 0;
}
