// Kernels of the myopicity two-sample test (reference: check_if_myopic, src/vgan.py:384-431, which calls
// torch-two-sample's MMDStatistic(...)(x, y, alphas=[bw], ret_matrix=True) and .pval(matrix)).
//
// That dependency is not part of the reference checkout and is unpinned: the arithmetic below follows its
// published algorithm as restated in oracle/vgan_oracle.py -- PARITY UNPINNED against the dependency itself.
//   1. kernel matrix of the pooled sample Z = [x ; y] ([m, p]):  K_ij = exp(-alpha |z_i - z_j|^2), K_ii = 1
//   2. for every row u of a 0/1 assignment matrix Ut [P, m]:  T = Ut . K  (vgan_gemm_grouped, "NN"), then
//      a = <u, T_u> = u^T K u  and  b = <u, r> with r = 1^T K (the T row of the all-ones assignment): vgan_rows_dot.
#include "gemm_core.hpp"

namespace vgan {

template <int VEC>
__global__ __launch_bounds__(kBlock, 2) void rbf_kernel_matrix_kernel(const float* __restrict__ Z, int ldz, int m, int p,
                                                                     const float* __restrict__ sq, float alpha,
                                                                     float* __restrict__ K, int ldk) {
    using G = GemmTile<64, 64, 32, KC, KC, VEC>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    f32x16 acc[1][1];
    zero_acc(acc);
    G::template run<false>(Z, ldz, Z, ldz, m0, n0, m, m, p, lds, nullptr, acc);
    const int j = n0 + G::sub_col(0);
    if (j >= m) return;
    const float sj = sq[j];
    const float c2 = -alpha * 1.4426950408889634f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = m0 + G::sub_row(0, r);
        if (i < m) {
            const float L = fmaxf(sq[i] + sj - 2.f * acc[0][0][r], 0.f);
            K[(long)i * ldk + j] = (i == j) ? 1.f : __builtin_amdgcn_exp2f(L * c2);
        }
    }
}

// RBF.forward (src/models/Mmd_loss_constrained.py:24-26): K_ij = sum_k exp(-L_ij / (bw * mult_k)) as an N x N matrix, for callers
// of the stand-alone module (the training step never materialises it).  dK (optional) = dK/dL, what the module's autograd needs.
struct RbfScaleList {
    int nk;
    float mult[VGAN_RBF_MAX_KERNELS];
};
template <int VEC>
__global__ __launch_bounds__(kBlock, 2) void rbf_multi_kernel_matrix_kernel(const float* __restrict__ Z, int ldz, int m, int p,
                                                                           const float* __restrict__ sq, const float* __restrict__ bw_ptr,
                                                                           RbfScaleList sl, float* __restrict__ K, int ldk,
                                                                           float* __restrict__ dK, int lddk) {
    using G = GemmTile<64, 64, 32, KC, KC, VEC>;
    __shared__ __attribute__((aligned(16))) float lds[G::kLdsFloats];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float bw = bw_ptr[0];
    f32x16 acc[1][1];
    zero_acc(acc);
    G::template run<false>(Z, ldz, Z, ldz, m0, n0, m, m, p, lds, nullptr, acc);
    const int j = n0 + G::sub_col(0);
    if (j >= m) return;
    const float sj = sq[j];
    float ck[VGAN_RBF_MAX_KERNELS], ik[VGAN_RBF_MAX_KERNELS];
#pragma unroll
    for (int k = 0; k < VGAN_RBF_MAX_KERNELS; ++k) {
        const float scale = bw * sl.mult[k < sl.nk ? k : 0];  // float32 product, as `bandwidth * bandwidth_multipliers`
        ik[k] = 1.f / scale;
        ck[k] = -1.4426950408889634f / scale;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = m0 + G::sub_row(0, r);
        if (i < m) {
            const float L = fmaxf(sq[i] + sj - 2.f * acc[0][0][r], 0.f);
            float kv = 0.f, dk = 0.f;
#pragma unroll
            for (int k = 0; k < VGAN_RBF_MAX_KERNELS; ++k)
                if (k < sl.nk) {
                    const float e = __builtin_amdgcn_exp2f(L * ck[k]);
                    kv += e;
                    dk = fmaf(-e, ik[k], dk);
                }
            K[(long)i * ldk + j] = kv;
            if (dK != nullptr) dK[(long)i * lddk + j] = dk;
        }
    }
}

// out[r] = sum_c A[r, c] * B[r * ldb + c]  (ldb = 0 broadcasts one row of B); one wave per row, float64 accumulation
__global__ void rows_dot_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, double* __restrict__ out,
                                int rows, int cols) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* a = A + (long)row * lda;
    const float* b = B + (long)row * ldb;
    double s = 0.0;
    for (int c = threadIdx.x & 63; c < cols; c += 64) s += (double)a[c] * (double)b[c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) out[row] = s;
}

}  // namespace vgan

using namespace vgan;

extern "C" int vgan_rbf_kernel_matrix(const float* Z, int ldz, int m, int p, const float* sq, float alpha, float* K, int ldk,
                                      vgan_stream_t stream) {
    VGAN_CHECK_ARG(Z && sq && K && m > 0 && p > 0 && ldz >= p && ldk >= m && alpha > 0.f);
    const bool vec = (p % 4 == 0) && (ldz % 4 == 0) && aligned16(Z);
    dim3 grid((m + 63) / 64, (m + 63) / 64);
    if (vec)
        hipLaunchKernelGGL(rbf_kernel_matrix_kernel<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, m, p, sq, alpha, K, ldk);
    else
        hipLaunchKernelGGL(rbf_kernel_matrix_kernel<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, m, p, sq, alpha, K, ldk);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_rows_dot(const float* A, int lda, const float* B, int ldb, double* out, int rows, int cols, vgan_stream_t stream) {
    VGAN_CHECK_ARG(A && B && out && rows > 0 && cols > 0 && lda >= cols && (ldb == 0 || ldb >= cols));
    hipLaunchKernelGGL(rows_dot_kernel, dim3((rows + 3) / 4), dim3(kBlock), 0, (hipStream_t)stream, A, lda, B, ldb, out, rows, cols);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_rbf_multi_kernel_matrix(const float* Z, int ldz, int m, int p, const float* sq, const float* bw,
                                            const float* multipliers, int n_kernels, float* K, int ldk, float* dK, int lddk,
                                            vgan_stream_t stream) {
    VGAN_CHECK_ARG(Z && sq && bw && K && multipliers && m > 0 && p > 0 && ldz >= p && ldk >= m && (dK == nullptr || lddk >= m));
    VGAN_CHECK_ARG(n_kernels >= 1 && n_kernels <= VGAN_RBF_MAX_KERNELS);
    RbfScaleList sl{};
    sl.nk = n_kernels;
    for (int k = 0; k < n_kernels; ++k) {
        VGAN_CHECK_ARG(multipliers[k] > 0.f);
        sl.mult[k] = multipliers[k];
    }
    const bool vec = (p % 4 == 0) && (ldz % 4 == 0) && aligned16(Z);
    dim3 grid((m + 63) / 64, (m + 63) / 64);
    if (vec)
        hipLaunchKernelGGL(rbf_multi_kernel_matrix_kernel<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, m, p, sq, bw, sl, K, ldk, dK, lddk);
    else
        hipLaunchKernelGGL(rbf_multi_kernel_matrix_kernel<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, m, p, sq, bw, sl, K, ldk, dK, lddk);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}
