// Device helpers shared by the optimiser / noise kernels (optim.hip) and the launches they ride in (grouped.hip).
#pragma once
#include "vgan_common.hpp"

namespace vgan {

// torch.optim.Adadelta, one element (src/vgan.py:567-568, :619): g += wd p; v = rho v + (1-rho) g^2;
// delta = sqrt(a + eps) / sqrt(v + eps) g; a = rho a + (1-rho) delta^2; p -= lr delta
__device__ __forceinline__ void adadelta_one(float& p, float g, float& v, float& a, float lr, float rho, float eps, float wd,
                                             float gs) {
    g = fmaf(wd, p, g * gs);
    v = fmaf(rho, v, (1.f - rho) * g * g);
    const float std = sqrtf(v + eps);
    const float delta = sqrtf(a + eps) / std * g;
    a = fmaf(rho, a, (1.f - rho) * delta * delta);
    p = fmaf(-lr, delta, p);
}

// Philox4x32-10 (Salmon et al., SC'11): counter = (index, stream_id), key = seed ^ step-derived words.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    const unsigned n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

__device__ __forceinline__ void noise_normal_body(float* __restrict__ z, int rows, int cols, int ld, int ones_col,
                                                  unsigned long long seed, const unsigned long long* __restrict__ step_counter,
                                                  unsigned long long stream_id, int vblock, int vgrid) {
    const unsigned long long step = step_counter ? step_counter[0] : 0ull;
    const long count = (long)rows * cols;
    const long nq = (count + 3) >> 2;
    for (long q = (long)vblock * blockDim.x + threadIdx.x; q < nq; q += (long)vgrid * blockDim.x) {
        unsigned c[4] = {(unsigned)q, (unsigned)((unsigned long long)q >> 32), (unsigned)step, (unsigned)(step >> 32)};
        philox4x32_10(c, (unsigned)seed ^ (unsigned)stream_id, (unsigned)(seed >> 32) ^ (unsigned)(stream_id >> 32) ^ 0x5bd1e995u);
        float o[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // Box-Muller on two uniform pairs
            const float r = sqrtf(-2.0f * logf(u01(c[2 * h])));
            float sn, cs;
            sincosf(6.283185307179586f * u01(c[2 * h + 1]), &sn, &cs);
            o[2 * h] = r * cs;
            o[2 * h + 1] = r * sn;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // element index is row-major over [rows, cols], independent of ld
            const long idx = 4 * q + e;
            if (idx < count) z[(idx / cols) * ld + (idx % cols)] = o[e];
        }
    }
    if (ones_col >= 0)  // homogeneous coordinate [z | 1] of the collapsed generator chain
        for (long r = (long)vblock * blockDim.x + threadIdx.x; r < rows; r += (long)vgrid * blockDim.x) z[r * ld + ones_col] = 1.0f;
}


}  // namespace vgan
