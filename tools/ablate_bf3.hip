// Ablation micro-benchmark of the split-bf16 tile main loop (tools only; not part of the library).
// Build variants with -DVGAN_ABLATE_NO_GLOBAL / _NO_MFMA / _ONE_PRODUCT / _NO_LDS_STORE / _NO_BARRIER, -DTBK=32|64, -DOCC_=n.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../v-gan_amd/csrc/gemm_bf3.hpp"
using namespace vgan;
namespace vgan { void set_error(const char*, ...) {} }
#ifndef TBK
#define TBK 64
#endif

#ifndef OCC_
#define OCC_ 2
#endif
__global__ __launch_bounds__(kBlock, OCC_) void k(const unsigned short* Zh, const unsigned short* Zl, int kp, int N, int tiles_per_row, float* out) {
    using G = GemmBF3<TBK>;
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    const int r0 = (blockIdx.x / tiles_per_row) * 64, c0 = (blockIdx.x % tiles_per_row) * 64;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    G::template run<false>(Zh, Zl, kp, Zh, Zl, kp, r0, c0, N, N, kp, lds, nullptr, acc);
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int N = 2048, kp = 832, ntiles = argc > 1 ? atoi(argv[1]) : 528;
    if (ntiles < 1 || ntiles > (N / 64) * (N / 64)) { printf("ntiles out of range\n"); return 1; }
    unsigned short *Zh, *Zl;
    float* out;
    hipMalloc(&Zh, (size_t)N * kp * 2);
    hipMalloc(&Zl, (size_t)N * kp * 2);
    hipMalloc(&out, (size_t)ntiles * 256 * 4);
    std::vector<unsigned short> h((size_t)N * kp);
    for (auto& v : h) v = 0x3F00 + rand() % 128;
    hipMemcpy(Zh, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (auto& v : h) v = 0x3B00 + rand() % 128;
    hipMemcpy(Zl, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tpr = N / 64;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(ntiles), dim3(256), 0, 0, Zh, Zl, kp, N, tpr, out);
    hipEventRecord(e0);
    const int it = 50;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(ntiles), dim3(256), 0, 0, Zh, Zl, kp, N, tpr, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 64 * 64 * kp * ntiles;
    printf("tiles=%d BK=%d occ=%d: %.1f us/launch, %.1f algorithmic TFLOP/s\n", ntiles, TBK, OCC_, ms / it * 1e3, fl / (ms / it * 1e-3) / 1e12);
    return 0;
}
