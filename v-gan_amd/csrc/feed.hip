// Input pipeline and sampling post-processing on the device.
//   vgan_shuffle_epoch   replaces DataLoader(shuffle=True, drop_last=True)'s per-epoch randperm (src/vgan.py:578-584) with a
//                        counter-based permutation evaluated on the GPU: no host draw, no sort, no H2D copy per epoch.
//   vgan_mask_unique     replaces np.unique(masks, axis=0, return_counts=True) of approx_subspace_dist (src/vgan.py:372-382).
#include "vgan_common.hpp"

namespace vgan {

// ---- counter-based random permutation of [0, N): balanced Feistel network + cycle walking ---------------------------
// A 2w-bit balanced Feistel network (2^(2w) >= N, < 4N) with a keyed 32-bit mixer as round function is a bijection of
// [0, 2^(2w)) for ANY round function; walking the cycle until the value drops below N restricts it to a bijection of [0, N)
// (expected < 4 evaluations).  Every index is computed independently: 16 B of state, no table, no sort, no atomics.
__host__ __device__ inline unsigned feistel_mix(unsigned x, unsigned k) {
    x ^= k;
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    x *= 0xC2B2AE3Du;
    x ^= x >> 16;
    return x;
}
constexpr int kFeistelRounds = 8;
__host__ __device__ inline unsigned long long feistel_perm(unsigned long long i, unsigned long long N, int w, unsigned long long seed,
                                                           unsigned long long epoch) {
    const unsigned mask = (w >= 32) ? 0xFFFFFFFFu : ((1u << w) - 1u);
    const unsigned k0 = (unsigned)seed ^ 0xA511E9B3u, k1 = (unsigned)(seed >> 32) ^ (unsigned)epoch, k2 = (unsigned)(epoch >> 32) ^ 0x63D83595u;
    unsigned long long v = i;
    do {
        unsigned l = (unsigned)(v >> w) & mask, r = (unsigned)v & mask;
#pragma unroll
        for (int q = 0; q < kFeistelRounds; ++q) {
            const unsigned f = feistel_mix(r, feistel_mix(k0 + 0x9E3779B9u * (unsigned)q, k1) ^ k2) & mask;
            const unsigned nl = r;
            r = l ^ f;
            l = nl;
        }
        v = ((unsigned long long)l << w) | r;
    } while (v >= N);
    return v;
}

__global__ __launch_bounds__(kBlock) void shuffle_epoch_kernel(int* __restrict__ perm, long count, unsigned long long N, int w,
                                                              unsigned long long seed, unsigned long long epoch) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) perm[i] = (int)feistel_perm((unsigned long long)i, N, w, seed, epoch);
}

// ---- unique rows of a boolean mask matrix with counts ---------------------------------------------------------------
// Rows are packed MSB-first into 64-bit words (feature 0 -> bit 63 of word 0), so that unsigned word-by-word comparison is
// numpy's lexicographic row order (False < True).  n is the number of sampled subspaces (500 by default): all-pairs
// comparison, one workgroup per row -- O(n^2 W) word compares, microseconds at that size, and fully deterministic.
__global__ __launch_bounds__(kBlock) void mask_pack_kernel(const unsigned char* __restrict__ masks, int ldm, int n, int d, int W,
                                                          unsigned long long* __restrict__ keys) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * W) return;
    const int i = (int)(idx / W), wq = (int)(idx % W);
    unsigned long long k = 0ull;
    for (int b = 0; b < 64; ++b) {
        const int j = wq * 64 + b;
        if (j < d && masks[(long)i * ldm + j] != 0) k |= 1ull << (63 - b);
    }
    keys[idx] = k;
}

__device__ __forceinline__ int key_compare(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, int W) {
    for (int q = 0; q < W; ++q) {
        const unsigned long long x = a[q], y = b[q];
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}

// pass 1: count[i] = #{j : key_j == key_i}, first[i] = 1 iff no j < i has the same key
__global__ __launch_bounds__(kBlock) void mask_unique_count_kernel(const unsigned long long* __restrict__ keys, int n, int W,
                                                                  int* __restrict__ count, int* __restrict__ first) {
    __shared__ int red[2][4];
    const int i = blockIdx.x;
    const unsigned long long* ki = keys + (long)i * W;
    int same = 0, earlier = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x)
        if (key_compare(keys + (long)j * W, ki, W) == 0) {
            ++same;
            earlier += j < i;
        }
    same = wave_sum(same);
    earlier = wave_sum(earlier);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = same;
        red[1][threadIdx.x >> 6] = earlier;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        count[i] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        first[i] = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) == 0;
    }
}

// pass 2: for the first row of every distinct key, its rank among the distinct keys; out_row[rank] = i, out_count[rank] = count
__global__ __launch_bounds__(kBlock) void mask_unique_rank_kernel(const unsigned long long* __restrict__ keys, int n, int W,
                                                                 const int* __restrict__ count, const int* __restrict__ first,
                                                                 int* __restrict__ out_row, int* __restrict__ out_count) {
    __shared__ int red[4];
    const int i = blockIdx.x;
    if (!first[i]) return;  // block-uniform
    const unsigned long long* ki = keys + (long)i * W;
    int below = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x)
        if (first[j] && key_compare(keys + (long)j * W, ki, W) < 0) ++below;
    below = wave_sum(below);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = below;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int rank = (red[0] + red[1]) + (red[2] + red[3]);
        out_row[rank] = i;
        out_count[rank] = count[i];
    }
}

}  // namespace vgan

using namespace vgan;

static int feistel_half_bits(unsigned long long N) {
    int w = 1;
    while (w < 32 && (1ull << (2 * w)) < N) ++w;
    return w;
}

extern "C" int vgan_shuffle_epoch(int32_t* perm, int64_t count, int64_t train_size, uint64_t seed, uint64_t epoch, vgan_stream_t stream) {
    VGAN_CHECK_ARG(perm && count > 0 && train_size > 0 && count <= train_size && train_size <= 0x7FFFFFFF);
    hipLaunchKernelGGL(shuffle_epoch_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, perm,
                       (long)count, (unsigned long long)train_size, feistel_half_bits((unsigned long long)train_size), seed, epoch);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

// host-side evaluation of the same permutation (tests, and callers that want an index without a launch)
extern "C" int64_t vgan_shuffle_index(int64_t i, int64_t train_size, uint64_t seed, uint64_t epoch) {
    if (i < 0 || train_size <= 0 || i >= train_size) return -1;
    return (int64_t)feistel_perm((unsigned long long)i, (unsigned long long)train_size, feistel_half_bits((unsigned long long)train_size), seed,
                                 epoch);
}

extern "C" int vgan_mask_unique(const uint8_t* masks, int ldm, int n, int d, uint64_t* keys, int32_t* work, int32_t* out_row,
                                int32_t* out_count, vgan_stream_t stream) {
    VGAN_CHECK_ARG(masks && keys && work && out_row && out_count && n > 0 && d > 0 && ldm >= d);
    const int W = (d + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* k = reinterpret_cast<unsigned long long*>(keys);
    hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)(((long)n * W + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, masks, ldm, n, d, W, k);
    VGAN_CHECK_LAUNCH();
    hipLaunchKernelGGL(mask_unique_count_kernel, dim3(n), dim3(kBlock), 0, s, k, n, W, work, work + n);
    VGAN_CHECK_LAUNCH();
    hipLaunchKernelGGL(mask_unique_rank_kernel, dim3(n), dim3(kBlock), 0, s, k, n, W, work, work + n, out_row, out_count);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}
