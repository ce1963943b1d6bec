// What would ONE persistent launch for the step's chain of small dependent launches buy?  
// The six chain launches of the c3 step (two backward grouped launches, Adadelta, two forward grouped launches, logits) each
// consume what ALL workgroups of the previous one produced, so inside one launch every hand-over is a grid-wide dependency.
// This probe times the two forms on the same body and geometry:
//   (a) P dependent launches of the body, captured in one HIP graph (what the step does today);
//   (b) ONE launch running the P phases with a device-scope arrival counter between them (cheapest correct form: phase data
//       moved with sc0 sc1 stores / loads, a relaxed ticket per workgroup, a bounded spin on the counter).
// Body of a phase: workgroup w reads the 32 KB chunk that workgroup (w + 1) % G wrote in the previous phase, adds 1, writes its own
// chunk (cross-workgroup dependency, ~the bytes a chain tile moves).  After P phases every word equals P (checked).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/chain_sync_probe tools/chain_sync_probe.hip ; run: chain_sync_probe [G] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kChunkWords = 8192;  // 32 KB per workgroup and phase

__device__ __forceinline__ void body(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int G, bool coherent) {
    const int w = blockIdx.x, from = (w + 1) % G;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, G * kChunkWords * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, G * kChunkWords * 4, 0x00020000);
    for (int i = threadIdx.x; i < kChunkWords / 4; i += blockDim.x) {
        u32x4 v = coherent ? __builtin_amdgcn_raw_buffer_load_b128(rs, (from * kChunkWords + 4 * i) * 4, 0, 17)
                           : __builtin_amdgcn_raw_buffer_load_b128(rs, (from * kChunkWords + 4 * i) * 4, 0, 0);
        v += 1u;
        if (coherent) __builtin_amdgcn_raw_buffer_store_b128(v, rd, (w * kChunkWords + 4 * i) * 4, 0, 17);
        else __builtin_amdgcn_raw_buffer_store_b128(v, rd, (w * kChunkWords + 4 * i) * 4, 0, 0);
    }
}

__global__ void phase_kernel(const unsigned* src, unsigned* dst, int G) { body(src, dst, G, false); }

// every wave reaches the end: the spin is bounded, a timeout raises *fail and the phases go on (results then wrong, reported)
__global__ void persistent_kernel(unsigned* a, unsigned* b, int G, int P, unsigned* counter, unsigned base, unsigned* fail) {
    for (int p = 0; p < P; ++p) {
        body((p & 1) ? b : a, (p & 1) ? a : b, G, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = base + (unsigned)(p + 1) * G;  // the counter only ever grows: no reset between launches
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { *fail = 1; break; }
            }
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 128, T = argc > 2 ? atoi(argv[2]) : 1024, P = 6, REP = 200;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    if (G > cus) { printf("G = %d exceeds the %d CUs: the persistent form needs every workgroup resident\n", G, cus); return 1; }
    unsigned *a, *b, *counter, *fail;
    CK(hipMalloc(&a, G * kChunkWords * 4)); CK(hipMalloc(&b, G * kChunkWords * 4));
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&fail, 4));
    CK(hipMemset(fail, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    // (a) P launches in a graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int p = 0; p < P; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(T), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, G);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned> h(G * kChunkWords);
    auto check = [&](const char* what) {
        CK(hipMemcpy(h.data(), a, h.size() * 4, hipMemcpyDeviceToHost));  // P even: the result is back in `a`
        size_t bad = 0;
        for (unsigned v : h) bad += v != (unsigned)P;
        unsigned f = 0; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("  %s: %zu wrong words of %zu, spin timeout %u\n", what, bad, h.size(), f);
    };
    std::vector<float> ta, tb;
    for (int round = 0; round < 5; ++round) {
        CK(hipMemsetAsync(a, 0, G * kChunkWords * 4, st));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        if (round == 0) check("launch form");
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ta.push_back(ms * 1e3f / REP);
        // (b) one persistent launch per P phases
        CK(hipMemsetAsync(a, 0, G * kChunkWords * 4, st)); CK(hipMemsetAsync(counter, 0, 4, st));
        hipLaunchKernelGGL(persistent_kernel, dim3(G), dim3(T), 0, st, a, b, G, P, counter, 0u, fail); CK(hipStreamSynchronize(st));
        if (round == 0) check("persistent form");
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r)
            hipLaunchKernelGGL(persistent_kernel, dim3(G), dim3(T), 0, st, a, b, G, P, counter, (unsigned)(r + 1) * P * G, fail);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); tb.push_back(ms * 1e3f / REP);
    }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
    printf("G = %d workgroups of %d threads, %d phases of 32 KB per workgroup: %d launches in a graph %.1f us (%.2f per phase); one persistent "
           "launch + counter %.1f us (%.2f per phase, its own launch included)\n", G, T, P, P, ta[2], ta[2] / P, tb[2], tb[2] / P);
    return 0;
}
