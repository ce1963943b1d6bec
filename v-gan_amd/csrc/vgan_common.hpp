// Shared device/host helpers for the V-GAN gfx950 kernels (CDNA4: 64-wide waves, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vgan_hip.h"

namespace vgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves, one per SIMD; two such workgroups per CU hide epilogues

void set_error(const char* fmt, ...);

#define VGAN_CHECK_ARG(cond)                                                        \
    do {                                                                            \
        if (!(cond)) {                                                              \
            ::vgan::set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond); \
            return VGAN_ERR_ARG;                                                    \
        }                                                                           \
    } while (0)

#define VGAN_CHECK_LAUNCH()                                                                     \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) {                                                                 \
            ::vgan::set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return VGAN_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// Column-max key: U > 0 always, so its IEEE bits order like the value; ties go to the LOWEST row.
__device__ __forceinline__ unsigned long long colkey_pack(float u, unsigned row) {
    return ((unsigned long long)__float_as_uint(u) << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float colkey_value(unsigned long long k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ unsigned colkey_row(unsigned long long k) { return 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull); }

// Column arg-max of U over one chunk of kColChunkRows rows for 64 columns (block bx covers columns 64*bx.., chunk by).
// Shared by the stand-alone kernel (rows.hip) and by the Gram launch, which runs it in surplus workgroups.
constexpr int kColChunkRows = 64;
template <int NW>  // waves in the calling workgroup: ALL of them take part (rows are dealt round-robin over the waves)
__device__ __forceinline__ void colmax_partial_body(const float* __restrict__ S, int lds, int row_offset,
                                                    unsigned long long* __restrict__ part, int n, int d, int from_softmax, int bx,
                                                    int by) {
    __shared__ unsigned long long colmax_red[NW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = bx * 64 + lane;
    const int r0 = by * kColChunkRows;
    const float tau = from_softmax ? 1.0f / (float)d : INFINITY;  // U given directly: no threshold
    unsigned long long best = 0ull;
    if (j < d) {
        // all loads of the chunk are issued before the first compare (the blocks that run this inside the Gram launch are
        // latency-bound); rows past the end are clamped to the last row, whose key they merely repeat
        // (NW need not divide the chunk: a wave's surplus turn repeats the chunk's last row, and a max does not mind)
        constexpr int PER = (kColChunkRows + NW - 1) / NW;
        float sv[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) sv[e] = S[(long)min(r0 + min(wave + e * NW, kColChunkRows - 1), n - 1) * lds + j];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const unsigned long long k = colkey_pack(sv[e] < tau ? sv[e] : 1.0f,
                                                     (unsigned)(row_offset + min(r0 + min(wave + e * NW, kColChunkRows - 1), n - 1)));
            best = k > best ? k : best;
        }
    }
    colmax_red[wave][lane] = best;
    __syncthreads();
    if (wave == 0 && j < d) {
        unsigned long long b = colmax_red[0][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) b = colmax_red[w][lane] > b ? colmax_red[w][lane] : b;
        part[(long)by * d + j] = b;
    }
}

// z = hi + lo with hi = bf16(z) (round to nearest even), lo = bf16(z - hi): the operand split of the bf16x3 MMD kernels
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float bf16_val(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ void split_bf16(float v, unsigned short& hi, unsigned short& lo) {
    hi = bf16_bits(v);
    lo = bf16_bits(v - bf16_val(hi));
}

// the value the split operands carry: hi + lo (16 significant bits of v); row norms of the bf16x3 kernels are taken from it
__device__ __forceinline__ float split_value(float v) {
    unsigned short hi, lo;
    split_bf16(v, hi, lo);
    return bf16_val(hi) + bf16_val(lo);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace vgan
