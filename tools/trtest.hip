// Probe of ds_read_b64_tr_b16 (gfx950): which elements of an LDS tile each lane receives.  Measurement aid.
//   hipcc --offload-arch=gfx950 -O2 tools/trtest.hip -o tools/bin/trtest && tools/bin/trtest
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s4;
__global__ void probe(short* out) {
    __shared__ __attribute__((aligned(16))) short tile[16 * 64];  // [row k][col n], value = 100 * k + n
    for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) tile[i] = (short)(100 * (i / 64) + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
    const int c0 = 16 * (g & 1), R = 8 * (g >> 1);
    lds_s4* a = (lds_s4*)(tile + (R + q) * 64 + c0 + 4 * p);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * sizeof(short));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, li = l & 15, c0 = 16 * (g & 1), R = 8 * (g >> 1);
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            printf(" %4d", h[l * 4 + e]);
            if (h[l * 4 + e] != 100 * (R + e) + c0 + li) ++bad;  // expectation: lane li of a group gets COLUMN c0 + li, rows R .. R+3
        }
        printf("\n");
    }
    printf("expectation (lane i of group: column c0 + i, element e = row R + e): %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return 0;
}
