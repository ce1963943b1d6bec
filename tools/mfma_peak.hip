// Pure-register fp32 MFMA issue-rate probe (tools only): what does v_mfma_f32_32x32x2_f32 sustain on THIS device?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, int iters) {
    float* out; (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    double fl = 2.0 * 32 * 32 * 2 * 8.0 * NACC * iters * 4.0 * blocks;
    printf("NACC=%d blocks=%d iters=%d: %.1f us, %.1f TFLOP/s\n", NACC, blocks, iters, ms * 1e3, fl / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    run<1>(256, 400); run<4>(256, 100); run<1>(512, 400); run<4>(1024, 100); run<4>(256, 2000); run<4>(256, 20000);
    return 0;
}
