// Prototype of the 64x64 split-bf16 tile main loop with direct-to-LDS staging (global_load_lds_dwordx4), checked against
// GemmBF3<64>::run on the same operands (tools only, not part of the library).  256 threads, two workgroups per CU.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../v-gan_amd/csrc/gemm_bf3.hpp"
using namespace vgan;
namespace vgan { void set_error(const char*, ...) {} }

struct G64 {
    static constexpr int BK = 64, PART = 64 * 128, BUF = 4 * PART, kLdsBytes = 2 * BUF;  // 65,536 B: two workgroups per CU
    typedef char __attribute__((address_space(3))) lds_c;
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x16& acc) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
        const int fi = lane & 31, fh = lane >> 5;
        // wave w fills part w (Ah, Al, Bh, Bl): 64 rows = 8 pieces of 8 rows x 128 B; lane's chunk (lane & 7) ^ swz(row)
        const unsigned short* base = wave == 0 ? Ah : wave == 1 ? Al : wave == 2 ? Bh : Bl;
        const long ld = wave < 2 ? lda : ldb;
        const int r0g = wave < 2 ? m0 : n0, lim = wave < 2 ? M : N;
        const char* src[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = 8 * e + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            src[e] = reinterpret_cast<const char*>(base + (long)min(r0g + row, lim - 1) * ld) + 16 * c;
        }
        auto fill = [&](int kt) {
            lds_c* d = lds + (kt & 1) * BUF + wave * PART;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[e] + 2 * (long)kt * BK),
                                                 (void __attribute__((address_space(3)))*)(d + e * 1024), 16, 0, 0);
        };
        const int nk = K / BK;
        fill(0);
        __syncthreads();
        const int sw = (fi >> 1) & 7;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) fill(kt + 1);
            const lds_c* buf = lds + (kt & 1) * BUF;
            const lds_c* pa = buf + (wm0 + fi) * 128;
            const lds_c* pb = buf + 2 * PART + (wn0 + fi) * 128;
            u32x4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int pos = ((2 * s + fh) ^ sw) << 4;
                ah[s] = *(const lds_u4*)(pa + pos);
                al[s] = *(const lds_u4*)(pa + PART + pos);
                bh[s] = *(const lds_u4*)(pb + pos);
                bl[s] = *(const lds_u4*)(pb + PART + pos);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[s]), xl = __builtin_bit_cast(bf16x8, al[s]);
                const bf16x8 yh = __builtin_bit_cast(bf16x8, bh[s]), yl = __builtin_bit_cast(bf16x8, bl[s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
            }
            __syncthreads();
        }
    }
};
// 64x64 tile on EIGHT waves: two groups of 2 x 2 waves, group g multiplying the k16 steps 2 g, 2 g + 1 of every K tile (register
// staging as GemmBF3: four 16-byte pieces per thread instead of eight); group 1 hands its accumulator to group 0 at the end.
struct K2 {
    static constexpr int BK = 64, ROWB = (BK + 8) * 2, PART = 64 * ROWB, BUF = 4 * PART, kLdsBytes = 2 * BUF;
    typedef char __attribute__((address_space(3))) lds_c;
    static constexpr int kThreads = 512;
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x16& acc) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int grp = wave >> 2, w4 = wave & 3;
        const int wm0 = (w4 >> 1) * 32, wn0 = (w4 & 1) * 32;
        const int fi = lane & 31, fh = lane >> 5;
        // staging: 4 parts x 64 rows x 8 pieces = 2048 pieces / 512 threads = 4: piece f = tid + 512 r -> part f >> 9 ... use part = r
        const char* src[4];
        int lofs;
        {
            const int row = tid >> 3, q = tid & 7;
            const long ra = (long)min(m0 + row, M - 1) * lda + 8 * q, rb = (long)min(n0 + row, N - 1) * ldb + 8 * q;
            src[0] = reinterpret_cast<const char*>(Ah + ra);
            src[1] = reinterpret_cast<const char*>(Al + ra);
            src[2] = reinterpret_cast<const char*>(Bh + rb);
            src[3] = reinterpret_cast<const char*>(Bl + rb);
            lofs = row * ROWB + q * 16;
        }
        u32x4 v[4];
        auto load = [&](int k0) {
#pragma unroll
            for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const u32x4*>(src[p] + 2 * (long)k0);
        };
        auto store = [&](lds_c* buf) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *(lds_u4*)(buf + p * PART + lofs) = v[p];
        };
        const int nk = K / BK;
        load(0);
        store(lds);
        if (nk > 1) load(BK);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const lds_c* buf = lds + (kt & 1) * BUF;
            const lds_c* pa = buf + (wm0 + fi) * ROWB + fh * 16 + grp * 64;
            const lds_c* pb = buf + 2 * PART + (wn0 + fi) * ROWB + fh * 16 + grp * 64;
            u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ah[s] = *(const lds_u4*)(pa + s * 32);
                al[s] = *(const lds_u4*)(pa + PART + s * 32);
                bh[s] = *(const lds_u4*)(pb + s * 32);
                bl[s] = *(const lds_u4*)(pb + PART + s * 32);
            }
            if (kt + 1 < nk) store(lds + ((kt & 1) ^ 1) * BUF);
            if (kt + 2 < nk) load((kt + 2) * BK);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[s]), xl = __builtin_bit_cast(bf16x8, al[s]);
                const bf16x8 yh = __builtin_bit_cast(bf16x8, bh[s]), yl = __builtin_bit_cast(bf16x8, bl[s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
            }
            __builtin_amdgcn_iglp_opt(0);
            __syncthreads();
        }
        lds_f* xch = (lds_f*)lds;
        if (grp == 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(w4 * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
        if (grp == 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += xch[(w4 * 16 + r) * 64 + lane];
    }
};
// 64 x 128 tile on eight waves: waves 0-3 the left 64 columns, 4-7 the right 64, each group 2 x 2 waves of 32 x 32; the A
// rows are staged once for both halves (48 KB of fill per K tile instead of 64 KB for two 64x64 workgroups), one workgroup per CU.
struct W128 {
    static constexpr int BK = 64, ROWB = (BK + 8) * 2, PA = 64 * ROWB, PB = 128 * ROWB, BUF = 2 * PA + 2 * PB, kLdsBytes = 2 * BUF;
    typedef char __attribute__((address_space(3))) lds_c;
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds_generic,
                                               f32x16& acc) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int half = wave >> 2, w4 = wave & 3;
        const int wm0 = (w4 >> 1) * 32, wn0 = half * 64 + (w4 & 1) * 32;
        const int fi = lane & 31, fh = lane >> 5;
        // pieces: A parts 2 x 64 rows x 8 = 1024, B parts 2 x 128 x 8 = 2048: 3072 / 512 threads = 6 per thread
        const char* src[6];
        int dst[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int f = tid + 512 * r;  // 0..3071
            if (f < 1024) {               // A: part f >> 9, row (f >> 3) & 63
                const int part = f >> 9, row = (f >> 3) & 63, q = f & 7;
                src[r] = reinterpret_cast<const char*>((part ? Al : Ah) + (long)min(m0 + row, M - 1) * lda + 8 * q);
                dst[r] = part * PA + row * ROWB + q * 16;
            } else {
                const int g = f - 1024, part = g >> 10, row = (g >> 3) & 127, q = g & 7;
                src[r] = reinterpret_cast<const char*>((part ? Bl : Bh) + (long)min(n0 + row, N - 1) * ldb + 8 * q);
                dst[r] = 2 * PA + part * PB + row * ROWB + q * 16;
            }
        }
        u32x4 v[6];
        auto load = [&](int k0) {
#pragma unroll
            for (int r = 0; r < 6; ++r) v[r] = *reinterpret_cast<const u32x4*>(src[r] + 2 * (long)k0);
        };
        auto store = [&](lds_c* buf) {
#pragma unroll
            for (int r = 0; r < 6; ++r) *(lds_u4*)(buf + dst[r]) = v[r];
        };
        const int nk = K / BK;
        load(0);
        store(lds);
        if (nk > 1) load(BK);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const lds_c* buf = lds + (kt & 1) * BUF;
            const lds_c* pa = buf + (wm0 + fi) * ROWB + fh * 16;
            const lds_c* pb = buf + 2 * PA + (wn0 + fi) * ROWB + fh * 16;
            u32x4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ah[s] = *(const lds_u4*)(pa + s * 32);
                al[s] = *(const lds_u4*)(pa + PA + s * 32);
                bh[s] = *(const lds_u4*)(pb + s * 32);
                bl[s] = *(const lds_u4*)(pb + PB + s * 32);
            }
            if (kt + 1 < nk) store(lds + ((kt & 1) ^ 1) * BUF);
            if (kt + 2 < nk) load((kt + 2) * BK);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[s]), xl = __builtin_bit_cast(bf16x8, al[s]);
                const bf16x8 yh = __builtin_bit_cast(bf16x8, bh[s]), yl = __builtin_bit_cast(bf16x8, bl[s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
            }
            __builtin_amdgcn_iglp_opt(0);
            __syncthreads();
        }
    }
};
template <class G>
__global__ __launch_bounds__(512, 2) void kw(const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int tiles_per_row, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    const int r0 = (blockIdx.x / tiles_per_row) * 64, c0 = (blockIdx.x % tiles_per_row) * 128;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    G::run(Zh, Zl, kp, Zh, Zl, kp, r0, c0, N, N, kp, lds, acc);
    for (int r = 0; r < 16; ++r) out[((size_t)blockIdx.x * 16 + r) * 512 + threadIdx.x] = acc[r];
}
struct B64 {
    static constexpr int kLdsBytes = GemmBF3<64>::kLdsBytes;
    __device__ static __forceinline__ void run(const unsigned short* Ah, const unsigned short* Al, long lda, const unsigned short* Bh,
                                               const unsigned short* Bl, long ldb, int m0, int n0, int M, int N, int K, char* lds, f32x16& acc) {
        GemmBF3<64>::run<false>(Ah, Al, lda, Bh, Bl, ldb, m0, n0, M, N, K, lds, nullptr, acc);
    }
};
template <class G, int NT>
__global__ __launch_bounds__(NT, 2) void k(const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int tiles_per_row, float* out) {
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    const int r0 = (blockIdx.x / tiles_per_row) * 64, c0 = (blockIdx.x % tiles_per_row) * 64;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    G::run(Zh, Zl, kp, Zh, Zl, kp, r0, c0, N, N, kp, lds, acc);
    if (threadIdx.x < 256)
        for (int r = 0; r < 16; ++r) out[((size_t)blockIdx.x * 16 + r) * 256 + threadIdx.x] = acc[r];
}
template <class G, int NT = 256>
static void bench(const char* name, const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int ntiles, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tpr = N / 64;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<G, NT>), dim3(ntiles), dim3(NT), 0, 0, Zh, Zl, kp, N, tpr, out);
    hipEventRecord(e0);
    const int it = 50;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<G, NT>), dim3(ntiles), dim3(NT), 0, 0, Zh, Zl, kp, N, tpr, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-6s tiles=%d: %.1f us/launch, %.1f algorithmic TFLOP/s\n", name, ntiles, ms / it * 1e3, 2.0 * 64 * 64 * kp * ntiles / (ms / it * 1e-3) / 1e12);
}
int main() {
    const int N = 2048, kp = 832;
    unsigned short *Zh, *Zl;
    float *o1, *o2, *o3;
    hipMalloc(&Zh, (size_t)N * kp * 2);
    hipMalloc(&Zl, (size_t)N * kp * 2);
    const size_t no = (size_t)528 * 16 * 256;
    hipMalloc(&o1, no * 4);
    hipMalloc(&o2, no * 4);
    hipMalloc(&o3, no * 4);
    std::vector<unsigned short> h((size_t)N * kp);
    for (auto& v : h) v = 0x3F00 + rand() % 128 + ((rand() & 1) << 15);
    hipMemcpy(Zh, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (auto& v : h) v = 0x3B00 + rand() % 128;
    hipMemcpy(Zl, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    {
        float* ow; hipMalloc(&ow, (size_t)264 * 16 * 512 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int nt : {128, 200, 256, 264}) {
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kw<W128>, dim3(nt), dim3(512), 0, 0, Zh, Zl, kp, N, N / 128, ow);
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kw<W128>, dim3(nt), dim3(512), 0, 0, Zh, Zl, kp, N, N / 128, ow);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("64x128 tiles=%d (= %d 64x64 tiles of work): %.1f us/launch, %.1f algorithmic TFLOP/s\n", nt, 2 * nt, ms / 50 * 1e3,
                   2.0 * 64 * 128 * kp * nt / (ms / 50 * 1e-3) / 1e12);
        }
    }
    for (int nt : {256, 392, 512, 528}) {
        bench<B64>("base", Zh, Zl, kp, N, nt, o1);
        bench<G64>("glds", Zh, Zl, kp, N, nt, o2);
        bench<K2, 512>("8waves", Zh, Zl, kp, N, nt, o3);
    }
    std::vector<float> a(no), b(no);
    hipMemcpy(a.data(), o1, no * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o2, no * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < no; ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); }
    printf("max |base - glds| = %.3g (max |value| %.3g)\n", md, mx);
    hipMemcpy(b.data(), o3, no * 4, hipMemcpyDeviceToHost);
    md = 0;
    for (size_t i = 0; i < no; ++i) md = fmax(md, fabs((double)a[i] - b[i]));
    printf("max |base - 8waves| = %.3g\n", md);
    return 0;
}
