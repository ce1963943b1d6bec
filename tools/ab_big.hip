// A/B harness (tools only): times GemmBF3Big::run<false> of whichever v-gan_amd/csrc/gemm_bf3.hpp is first on the include path
// (-I. for the tree, -I<dir holding another revision> for `git show REV:v-gan_amd/csrc/gemm_bf3.hpp`), ten warm-up launches before
// every timed block: the first launches after an idle period run ~20 % slower (clock ramp).
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "v-gan_amd/csrc/gemm_bf3.hpp"
using namespace vgan;
namespace vgan { void set_error(const char*, ...) {} }
#ifndef NAME
#define NAME "?"
#endif
__global__ __launch_bounds__(512, 2) void k(const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int tiles_per_row, float* out) {
    using G = GemmBF3Big;
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    const int r0 = (blockIdx.x / tiles_per_row) * 128, c0 = (blockIdx.x % tiles_per_row) * 128;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    G::run<false>(Zh, Zl, kp, Zh, Zl, kp, r0, c0, N, N, kp, lds, acc);
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    const int N = 8192, kp = 4096, tpr = N / 128, ntiles = tpr * tpr;
    unsigned short *Zh, *Zl; float* out;
    hipMalloc(&Zh, (size_t)N * kp * 2); hipMalloc(&Zl, (size_t)N * kp * 2); hipMalloc(&out, (size_t)ntiles * 512 * 4);
    std::vector<unsigned short> h((size_t)N * kp);
    for (auto& v : h) v = 0x3F00 + rand() % 128 + ((rand() & 1) << 15);
    hipMemcpy(Zh, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (auto& v : h) v = 0x3B00 + rand() % 128;
    hipMemcpy(Zl, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(ntiles), dim3(512), 0, 0, Zh, Zl, kp, N, tpr, out);
        hipEventRecord(e0);
        const int it = 10;
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(ntiles), dim3(512), 0, 0, Zh, Zl, kp, N, tpr, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * 128 * 128 * kp * (double)ntiles, t = ms / it * 1e-3;
        printf("%-10s %.3f ms/launch, executed %.3f of 2.5 PF\n", NAME, t * 1e3, 3.0 * fl / t / 2.5e15);
    }
    return 0;
}
