// Adadelta over one flat parameter buffer, the Philox noise feed, and the detector's MSE term.
// Reference ops replaced: torch.optim.Adadelta.step (src/vgan.py:567-568, :619; 207-210),
// noise_tensor.normal_() (src/vgan.py:610, :270, :307), __distance(x, y, 'L2') (src/vgan.py:58-59).
// All HBM-streaming: 16 B per lane, grid-stride over <= 2048 workgroups.
#include "optim_common.hpp"

namespace vgan {

// g may be `nslabs` split-K slabs `slab_stride` apart: they are summed here in ascending order (one pass less)
__global__ __launch_bounds__(kBlock) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                         float* __restrict__ acc, long count, float lr, float rho, float eps,
                                                         float wd, float gs, int vec, int nslabs, long slab_stride) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const long nv = count >> 2;
        for (long q = i; q < nv; q += stride) {
            float4 pv = reinterpret_cast<float4*>(p)[q], gv = reinterpret_cast<const float4*>(g)[q];
            for (int sl = 1; sl < nslabs; ++sl) {
                const float4 b = reinterpret_cast<const float4*>(g + sl * slab_stride)[q];
                gv.x += b.x; gv.y += b.y; gv.z += b.z; gv.w += b.w;
            }
            float4 vv = reinterpret_cast<float4*>(sq)[q], av = reinterpret_cast<float4*>(acc)[q];
            adadelta_one(pv.x, gv.x, vv.x, av.x, lr, rho, eps, wd, gs);
            adadelta_one(pv.y, gv.y, vv.y, av.y, lr, rho, eps, wd, gs);
            adadelta_one(pv.z, gv.z, vv.z, av.z, lr, rho, eps, wd, gs);
            adadelta_one(pv.w, gv.w, vv.w, av.w, lr, rho, eps, wd, gs);
            reinterpret_cast<float4*>(p)[q] = pv;
            reinterpret_cast<float4*>(sq)[q] = vv;
            reinterpret_cast<float4*>(acc)[q] = av;
        }
        i += nv << 2;  // tail
    }
    for (long q = i; q < count; q += stride) {
        float gq = g[q];
        for (int sl = 1; sl < nslabs; ++sl) gq += g[sl * slab_stride + q];
        adadelta_one(p[q], gq, sq[q], acc[q], lr, rho, eps, wd, gs);
    }
}

// dst[i] = sum_s src[s*slab_stride + i], s ascending (fixed order)
__global__ __launch_bounds__(kBlock) void reduce_slabs_kernel(const float* __restrict__ src, long slab_stride, int nslabs,
                                                             float* __restrict__ dst, long count, int vec) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const long nv = count >> 2;
        for (long q = i; q < nv; q += stride) {
            float4 a = reinterpret_cast<const float4*>(src)[q];
            for (int s = 1; s < nslabs; ++s) {
                const float4 b = reinterpret_cast<const float4*>(src + s * slab_stride)[q];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            reinterpret_cast<float4*>(dst)[q] = a;
        }
        i += nv << 2;
    }
    for (long q = i; q < count; q += stride) {
        float a = src[q];
        for (int s = 1; s < nslabs; ++s) a += src[s * slab_stride + q];
        dst[q] = a;
    }
}

__global__ __launch_bounds__(kBlock) void noise_normal_kernel(float* __restrict__ z, int rows, int cols, int ld, int ones_col,
                                                             unsigned long long seed, const unsigned long long* __restrict__ step_counter,
                                                             unsigned long long stream_id) {
    noise_normal_body(z, rows, cols, ld, ones_col, seed, step_counter, stream_id, blockIdx.x, gridDim.x);
}

// Adadelta for the collapsed generator chain: the gradient of flat element i is read from the packed (homogeneous)
// gradient image at pmap[i], and the updated parameter is ALSO written to the packed weight image at the same offset,
// so neither an unpack launch before nor a pack launch after the optimiser is needed.  pmap[i] < 0: layout padding.
__global__ __launch_bounds__(kBlock) void adadelta_packed_kernel(float* __restrict__ p, const int* __restrict__ pmap,
                                                                const float* __restrict__ gpacked, float* __restrict__ wpacked,
                                                                float* __restrict__ sq, float* __restrict__ acc, long count, float lr,
                                                                float rho, float eps, float wd, float gs, float* __restrict__ z,
                                                                int zrows, int zcols, int zld, int zones, unsigned long long seed,
                                                                const unsigned long long* __restrict__ step_counter, int elem_blocks,
                                                                int noise_blocks) {
    const int b = blockIdx.x;
    if (b < elem_blocks) {
        // every load of an element is issued at once (the state reads do not wait for the index map); only the packed
        // gradient needs the map -- two dependent accesses instead of three
        const long stride = (long)elem_blocks * blockDim.x;
        for (long q = (long)b * blockDim.x + threadIdx.x; q < count; q += stride) {
            const int m = pmap[q];
            float pv = p[q], v = sq[q], a = acc[q];
            if (m < 0) continue;
            adadelta_one(pv, gpacked[m], v, a, lr, rho, eps, wd, gs);
            p[q] = pv;
            sq[q] = v;
            acc[q] = a;
            wpacked[m] = pv;
        }
    } else {
        // The optimiser is the last kernel of a step and the noise draw the first of the next one: the draw for the NEXT
        // step (the step counter was already advanced by the loss kernel) rides along here, in workgroups of its own.
        // (Also gathering the next step's batch rows here was measured: this launch 5.1 -> 8.6 us for 0.2 us off the
        // mask/projection kernel -- the gather's cold HBM rows are what costs, wherever it runs.)
        noise_normal_body(z, zrows, zcols, zld, zones, seed, step_counter, 0ull, b - elem_blocks, noise_blocks);
    }
}

// Homogeneous packing of Linear layers: Wt = [[W, b], [0, 1]] (zero padded to multiples of 4), so that a chain of
// bias-Linear layers is a chain of plain matrix products; unpack = the reverse for the gradients.
// desc[8*k ..]: {W ptr, b ptr, packed ptr, rows(out), cols(in), ldw, ldp, unused}
__global__ __launch_bounds__(kBlock) void homogeneous_pack_kernel(const long long* __restrict__ desc, int unpack) {
    const long long* e = desc + 8 * blockIdx.y;
    float* W = reinterpret_cast<float*>(e[0]);
    float* b = reinterpret_cast<float*>(e[1]);
    float* P = reinterpret_cast<float*>(e[2]);
    const int rows = (int)e[3], cols = (int)e[4], ldw = (int)e[5], ldp = (int)e[6];
    const long total = (long)(rows + 1) * (cols + 1);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / (cols + 1)), c = (int)(idx % (cols + 1));
        if (unpack) {
            if (r < rows) {
                const float v = P[(long)r * ldp + c];
                if (c < cols) W[(long)r * ldw + c] = v; else b[r] = v;
            }
        } else {
            P[(long)r * ldp + c] = r < rows ? (c < cols ? W[(long)r * ldw + c] : b[r]) : (c == cols ? 1.0f : 0.0f);
        }
    }
}

// out[0] (+)= scale * sum((a-b)^2)
__global__ __launch_bounds__(kBlock) void mse_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int n,
                                                    int d, float scale, float* __restrict__ out, int accumulate) {
    __shared__ double red[4];
    double s = 0.0;
    const long total = (long)n * d;
    for (long idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int i = (int)(idx / d), j = (int)(idx % d);
        const float df = a[(long)i * lda + j] - b[(long)i * ldb + j];
        s += (double)df * (double)df;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(((red[0] + red[1]) + (red[2] + red[3])) * (double)scale);
        out[0] = accumulate ? out[0] + v : v;
    }
}

// Squared-error term of VGAN.fit's detector loss, value and gradient in one pass (src/vgan.py:58-59, :276-277):
// part[block] = sum over the block's rows of (pred - target)^2 (float64), g = gscale * (pred - target).
__global__ __launch_bounds__(kBlock) void mse_grad_kernel(const float* __restrict__ target, int ldt, const float* __restrict__ pred,
                                                         int ldp, int n, int d, float gscale, double* __restrict__ part,
                                                         float* __restrict__ g, int ldg) {
    __shared__ double red[4];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    double s = 0.0;
    if (row < n) {
        const float* t = target + (long)row * ldt;
        const float* q = pred + (long)row * ldp;
        float* o = g + (long)row * ldg;
        for (int j = threadIdx.x & 63; j < d; j += 64) {
            const float df = q[j] - t[j];
            s += (double)df * (double)df;
            o[j] = gscale * df;
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] (+)= scale * sum(in[0..count))  -- one workgroup, fixed summation order
__global__ __launch_bounds__(kBlock) void sum_f64_kernel(const double* __restrict__ in, int count, double scale, float* __restrict__ out,
                                                        int accumulate) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) s += in[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(((red[0] + red[1]) + (red[2] + red[3])) * scale);
        out[0] = accumulate ? out[0] + v : v;
    }
}

}  // namespace vgan

using namespace vgan;

static inline int stream_grid(long work_items) {
    long g = (work_items + kBlock - 1) / kBlock;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

extern "C" int vgan_adadelta_step(float* p, const float* g, int nslabs, int64_t slab_stride, float* sq_avg, float* acc_delta,
                                  int64_t count, float lr, float rho, float eps, float weight_decay, float grad_scale,
                                  vgan_stream_t stream) {
    VGAN_CHECK_ARG(p && g && sq_avg && acc_delta && count > 0 && nslabs >= 1 && (nslabs == 1 || slab_stride >= count));
    const int vec = aligned16(p) && aligned16(g) && aligned16(sq_avg) && aligned16(acc_delta) && (nslabs == 1 || slab_stride % 4 == 0);
    hipLaunchKernelGGL(adadelta_kernel, dim3(stream_grid(vec ? (count + 3) / 4 : count)), dim3(kBlock), 0, (hipStream_t)stream, p, g,
                       sq_avg, acc_delta, (long)count, lr, rho, eps, weight_decay, grad_scale, vec, nslabs, (long)slab_stride);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_adadelta_step_packed(float* p, const int32_t* pmap, const float* g_packed, float* w_packed, float* sq_avg,
                                         float* acc_delta, int64_t count, float lr, float rho, float eps, float weight_decay,
                                         float grad_scale, float* next_noise, int noise_rows, int noise_cols, int noise_ld,
                                         int noise_ones_col, uint64_t seed, const uint64_t* step_counter, vgan_stream_t stream) {
    VGAN_CHECK_ARG(p && pmap && g_packed && w_packed && sq_avg && acc_delta && count > 0);
    VGAN_CHECK_ARG(next_noise == nullptr || (noise_rows > 0 && noise_cols > 0 && noise_ld >= noise_cols && noise_ones_col < noise_ld));
    const int elem_blocks = stream_grid(count);
    const int noise_blocks = next_noise != nullptr ? stream_grid(((long)noise_rows * noise_cols + 3) / 4) : 0;
    hipLaunchKernelGGL(adadelta_packed_kernel, dim3(elem_blocks + noise_blocks), dim3(kBlock), 0, (hipStream_t)stream, p,
                       pmap, g_packed, w_packed, sq_avg, acc_delta, (long)count, lr, rho, eps, weight_decay, grad_scale, next_noise,
                       noise_rows, noise_cols, noise_ld, noise_ones_col, (unsigned long long)seed,
                       reinterpret_cast<const unsigned long long*>(step_counter), elem_blocks, noise_blocks);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_reduce_slabs(const float* src, int64_t slab_stride, int nslabs, float* dst, int64_t count, vgan_stream_t stream) {
    VGAN_CHECK_ARG(src && dst && nslabs >= 1 && count > 0 && (nslabs == 1 || slab_stride >= count));
    const int vec = aligned16(src) && aligned16(dst) && (slab_stride % 4 == 0);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(stream_grid(vec ? (count + 3) / 4 : count)), dim3(kBlock), 0, (hipStream_t)stream, src,
                       (long)slab_stride, nslabs, dst, (long)count, vec);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_noise_normal(float* z, int rows, int cols, int ld, int ones_col, uint64_t seed, const uint64_t* step_counter,
                                 uint64_t stream_id, vgan_stream_t stream) {
    VGAN_CHECK_ARG(z && rows > 0 && cols > 0 && ld >= cols && ones_col < ld && (ones_col < 0 || ones_col >= cols));
    const long count = (long)rows * cols;
    hipLaunchKernelGGL(noise_normal_kernel, dim3(stream_grid((count + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, z, rows, cols, ld,
                       ones_col, (unsigned long long)seed, reinterpret_cast<const unsigned long long*>(step_counter),
                       (unsigned long long)stream_id);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_homogeneous_pack(const int64_t* desc, int count, int max_elems, int unpack, vgan_stream_t stream) {
    VGAN_CHECK_ARG(desc && count > 0 && count <= 64 && max_elems > 0);
    hipLaunchKernelGGL(homogeneous_pack_kernel, dim3(stream_grid(max_elems), count), dim3(kBlock), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(desc), unpack);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mse(const float* a, int lda, const float* b, int ldb, int n, int d, float scale, float* out, int accumulate,
                        vgan_stream_t stream) {
    VGAN_CHECK_ARG(a && b && out && n > 0 && d > 0 && lda >= d && ldb >= d);
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, a, lda, b, ldb, n, d, scale, out, accumulate);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mse_grad(const float* target, int ldt, const float* pred, int ldp, int n, int d, float gscale, double* part,
                             float* g, int ldg, vgan_stream_t stream) {
    VGAN_CHECK_ARG(target && pred && part && g && n > 0 && d > 0 && ldt >= d && ldp >= d && ldg >= d);
    hipLaunchKernelGGL(mse_grad_kernel, dim3((n + 3) / 4), dim3(kBlock), 0, (hipStream_t)stream, target, ldt, pred, ldp, n, d, gscale, part,
                       g, ldg);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_sum_f64(const double* in, int count, double scale, float* out, int accumulate, vgan_stream_t stream) {
    VGAN_CHECK_ARG(in && out && count > 0);
    hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, in, count, scale, out, accumulate);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}
