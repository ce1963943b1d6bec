// Probe (tools only): what v_dot2c_f32_bf16 (the lowering of __builtin_amdgcn_fdot2_f32_bf16 on gfx950) returns for
// D = a.x * 1 + a.y * 1 + C, against the exact fp32 sum, on random bf16 pairs of mixed magnitude.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, const float* cin, float* out, float* out_reg, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
    out[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, in[i]), ones, cin[i], false);
    // ones held in a register that the compiler cannot fold into a literal
    unsigned o = 0x3f803f80u;
    asm volatile("v_mov_b32 %0, %0" : "+v"(o));
    out_reg[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, in[i]), __builtin_bit_cast(bf16x2, o), cin[i], false);
}
static float bf(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 16;
    std::vector<unsigned> h(n); std::vector<float> c(n);
    for (int i = 0; i < n; ++i) {
        unsigned short a = (unsigned short)((0x3800 + rand() % 0x800) | ((rand() & 1) << 15)), b = (unsigned short)((0x3000 + rand() % 0x1000) | ((rand() & 1) << 15));
        h[i] = a | ((unsigned)b << 16);
        c[i] = (i & 1) ? 0.f : (float)(rand() % 1000) * 1e-3f;
    }
    unsigned* din; float *dc, *dout, *dreg;
    hipMalloc(&din, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dreg, n * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, din, dc, dout, dreg, n);
    std::vector<float> o(n), r(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dreg, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, badr = 0, shown = 0;
    for (int i = 0; i < n; ++i) {
        const float x = bf(h[i] & 0xFFFF), y = bf(h[i] >> 16);
        const double want = (double)x + (double)y + (double)c[i];
        const double tol = 2e-7 * (fabs(x) + fabs(y) + fabs(c[i]));
        if (fabs(o[i] - want) > tol) { ++bad; if (shown++ < 6) printf("  x %.9g y %.9g c %.9g -> literal %.9g register %.9g want %.9g\n", x, y, c[i], o[i], r[i], want); }
        if (fabs(r[i] - want) > tol) ++badr;
    }
    printf("v_dot2c_f32_bf16 with ones: %d of %d wrong (literal operand), %d wrong (register operand)\n", bad, n, badr);
    return 0;
}
