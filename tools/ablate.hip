// Ablation micro-benchmark of the fp32-MFMA tile main loop (tools only; not part of the library).
// Build variants with -DVGAN_ABLATE_NO_GLOBAL / _NO_MFMA / _NO_LDS_STORE / _NO_BARRIER and compare.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../v-gan_amd/csrc/gemm_core.hpp"
using namespace vgan;
namespace vgan { void set_error(const char*, ...) {} }

#ifndef BK_
#define BK_ 32
#endif
#ifndef OCC_
#define OCC_ 2
#endif
#ifndef BM_
#define BM_ 64
#endif
#ifndef BN_
#define BN_ 64
#endif
template <int VEC>
__global__ __launch_bounds__(kBlock, OCC_) void k(const float* Z, int ldz, int p, int tiles_per_row, float* out) {
    using G = GemmTile<BM_, BN_, BK_, KC, KC, VEC>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    const int r0 = (blockIdx.x / tiles_per_row) * BM_, c0 = (blockIdx.x % tiles_per_row) * BN_;
    f32x16 acc[G::WM][G::WN];
    zero_acc(acc);
    G::template run<false>(Z, ldz, Z, ldz, r0, c0, 1 << 30, 1 << 30, p, lds, nullptr, acc);
    float s = 0;
    for (int i = 0; i < G::WM; ++i) for (int j = 0; j < G::WN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    int N = 2048, p = 800, ntiles = argc > 1 ? atoi(argv[1]) : 528;
    if (ntiles < 1 || ntiles > (N / BM_) * (N / BN_)) { printf("ntiles out of range\n"); return 1; }
    float *Z, *out;
    hipMalloc(&Z, (size_t)N * p * 4);
    hipMalloc(&out, (size_t)ntiles * 256 * 4);
    std::vector<float> h((size_t)N * p);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(Z, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tpr = N / BN_;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<4>, dim3(ntiles), dim3(256), 0, 0, Z, p, p, tpr, out);
    hipEventRecord(e0);
    const int it = 50;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k<4>, dim3(ntiles), dim3(256), 0, 0, Z, p, p, tpr, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * BM_ * BN_ * p * ntiles;
    printf("tiles=%d %dx%dx%d: %.1f us/launch, %.1f TFLOP/s\n", ntiles, BM_, BN_, BK_, ms / it * 1e3, fl / (ms / it * 1e-3) / 1e12);
    return 0;
}
