#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <string.h>
static double bf(unsigned short v) { union { unsigned u; float x; } q; q.u = (unsigned)v << 16; return (double)q.x; }
#include "../v-gan_amd/csrc/gemm_bf3.hpp"
using namespace vgan;
namespace vgan { void set_error(const char*, ...) {} }
__global__ __launch_bounds__(512, 2) void k(const unsigned short* Zh, const unsigned short* Zl, int kp, int N, float* out) {
    using G = GemmBF3Big;
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    __shared__ float rs[128];
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    G::run<true>(Zh, Zl, kp, Zh, Zl, kp, blockIdx.x * 128, 0, N, N, kp, lds, acc, rs);
    if (threadIdx.x < 128) out[blockIdx.x * 128 + threadIdx.x] = rs[threadIdx.x];
}
int main() {
    const int N = 512, kp = 256;
    unsigned short *Zh, *Zl; float* out;
    hipMalloc(&Zh, (size_t)N * kp * 2); hipMalloc(&Zl, (size_t)N * kp * 2); hipMalloc(&out, N * 4);
    std::vector<unsigned short> h((size_t)N * kp), l((size_t)N * kp);
    for (auto& v : h) v = 0x3F00 + rand() % 128;
    for (auto& v : l) v = 0x3B00 + rand() % 128;
    hipMemcpy(Zh, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(Zl, l.data(), l.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(N / 128), dim3(512), 0, 0, Zh, Zl, kp, N, out);
    std::vector<float> o(N);
    hipMemcpy(o.data(), out, N * 4, hipMemcpyDeviceToHost);
    auto f = [](unsigned short v) { return bf(v); };
    int bad = 0;
    for (int r = 0; r < N; ++r) {
        double s = 0;
        for (int c = 0; c < kp; ++c) s += f(h[(size_t)r * kp + c]) + f(l[(size_t)r * kp + c]);
        if (fabs(o[r] - s) > 1e-3 * fabs(s)) {
            if (bad < 10) {
                // which row does it match?
                int match = -1;
                for (int q = 0; q < N; ++q) { double t = 0; for (int c = 0; c < kp; ++c) t += f(h[(size_t)q * kp + c]) + f(l[(size_t)q * kp + c]); if (fabs(o[r] - t) < 1e-5 * fabs(t)) match = q; }
                printf("row %d: got %.6f want %.6f (matches row %d)\n", r, o[r], s, match);
            }
            ++bad;
        }
    }
    printf("bad rows: %d of %d\n", bad, N);
    return 0;
}
