// Split-bf16 ("bf16x3") variants of the two dense contractions of the MMD, for large problems.
//
// gfx950 has no TF32; its fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s), the bf16 MFMA 16x faster.  Writing
// every operand as z = hi + lo with hi = bf16(z), lo = bf16(z - hi) (16 significant bits together) and computing
//     g = hi.hi' + hi.lo' + lo.hi'                         (three v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// drops only the lo.lo' term (2^-18 relative per product, random sign) and the rounding of lo (2^-17): the Gram entry
// keeps ~3e-7 relative accuracy at K = 784 -- the level of an fp32 fma chain of that length -- at 3/16 of the fp32
// MFMA time.  The C/D fragment layout of the MFMA does not depend on the input type, so the fused epilogues (distance,
// exp, squaring chain, block sums, gradient weights) are the fp32 kernels' code, character for character.
//
// Operands are prepared once per step (inside the mask / projection launch, rows.hip: mask_forward_bf3_kernel, or by
// vgan_mmd_bf3_prepare where that kernel's shape contract does not hold): Z -> (Zh, Zl), ROW-MAJOR split images that BOTH
// contractions read -- the Gram directly, the backward product W . Z through transposed LDS reads (ds_read_b64_tr_b16,
// GemmBF3::run_bt), which hand each lane the 8 consecutive k (= Z rows) its B fragment needs.  The transposed copies
// (ZTh, ZTl) of round 1 are written only on request (vgan_mmd_backward_bf3, kept for measurement).
#include "gemm_bf3.hpp"
#include "gemm_bf3w.hpp"
#include "mmd_common.hpp"
#include "mmd_xx.hpp"

namespace vgan {


// ---- operand preparation: 64x64 tiles of Z -> row-major and transposed hi/lo images ---------------------------
// Zh/Zl [rows_pad, kp]  (kp = features padded to 64, zero filled);  ZTh/ZTl [kp, kn]  (kn = rows padded to 64)
__global__ __launch_bounds__(kBlock) void bf3_prepare_kernel(const float* __restrict__ Z, int ldz, int rows, int p,
                                                            unsigned short* __restrict__ Zh, unsigned short* __restrict__ Zl, int kp,
                                                            unsigned short* __restrict__ ZTh, unsigned short* __restrict__ ZTl, int kn,
                                                            int vec) {
    // one 64 (rows k) x 64 (features j) tile per workgroup; thread t owns the 4x4 patch rows 4*(t/16).., features 4*(t%16)..
    // so that both images leave as 8-byte (4 x bf16) stores: row-major straight away, transposed after a pass through LDS
    __shared__ unsigned short th[64][68], tl[64][68];  // [feature j][row k]
    const int j0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int tj = 4 * (threadIdx.x & 15), tk = 4 * (threadIdx.x >> 4);
    unsigned short hi[4][4], lo[4][4];  // [row][feature]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int k = k0 + tk + a;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k < rows) {
            if (vec && j0 + tj + 3 < p) {
                const float4 q = *reinterpret_cast<const float4*>(Z + (long)k * ldz + j0 + tj);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (j0 + tj + b < p) v[b] = Z[(long)k * ldz + j0 + tj + b];
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) split_bf16(v[b], hi[a][b], lo[a][b]);
        if (k < rows) {  // row-major image (kp is a multiple of 64: the 8-byte store is aligned)
            *reinterpret_cast<uint2*>(Zh + (long)k * kp + j0 + tj) =
                make_uint2((unsigned)hi[a][0] | ((unsigned)hi[a][1] << 16), (unsigned)hi[a][2] | ((unsigned)hi[a][3] << 16));
            *reinterpret_cast<uint2*>(Zl + (long)k * kp + j0 + tj) =
                make_uint2((unsigned)lo[a][0] | ((unsigned)lo[a][1] << 16), (unsigned)lo[a][2] | ((unsigned)lo[a][3] << 16));
        }
    }
    if (ZTh == nullptr) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            th[tj + b][tk + a] = hi[a][b];
            tl[tj + b][tk + a] = lo[a][b];
        }
    __syncthreads();
    // thread t now owns feature rows 4*(t/16).. x 4 consecutive k at 4*(t%16)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int j = tk + a, kk = tj;
        const long o = (long)(j0 + j) * kn + k0 + kk;
        *reinterpret_cast<uint2*>(ZTh + o) = make_uint2((unsigned)th[j][kk] | ((unsigned)th[j][kk + 1] << 16),
                                                        (unsigned)th[j][kk + 2] | ((unsigned)th[j][kk + 3] << 16));
        *reinterpret_cast<uint2*>(ZTl + o) = make_uint2((unsigned)tl[j][kk] | ((unsigned)tl[j][kk + 1] << 16),
                                                        (unsigned)tl[j][kk + 2] | ((unsigned)tl[j][kk + 3] << 16));
    }
}

// ---- Gram tile + fused epilogue (see mmd.hip's mmd_gram_kernel; Wg leaves as a hi/lo bf16 pair) ----------------
template <int BK>
__global__ __launch_bounds__(kBlock, BK == 64 ? 2 : 3) void mmd_gram_bf3_kernel(const unsigned short* __restrict__ Zh, const unsigned short* __restrict__ Zl,
                                                                int kp, const float* __restrict__ sq, int n,
                                                                const float* __restrict__ bw_ptr, const TileDesc* __restrict__ tiles,
                                                                int ntiles, unsigned short* __restrict__ Wh,
                                                                unsigned short* __restrict__ Wl, int ldw, int wrow0,
                                                                float* __restrict__ partial, ColmaxJob cj) {
    using G = GemmBF3<BK>;
    __shared__ __attribute__((aligned(16))) char lds[G::kLdsBytes];
    __shared__ float red[8];
    if ((int)blockIdx.x >= ntiles) {
        const int cb = blockIdx.x - ntiles;
        colmax_partial_body<4>(cj.S, cj.lds, cj.row_offset, cj.part, cj.n, cj.d, cj.from_softmax, cb % cj.nbx, cb / cj.nbx);
        return;
    }
    const TileDesc td = tiles[blockIdx.x];
    // the epilogue's operands (row norms, bandwidth) are requested BEFORE the main loop: issued after it they would add one
    // full memory latency (~1 us) to every tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = td.c0 + G::sub_col();
    const bool jok = j < td.clim;
    const float sj = sq[min(j, td.clim - 1)];
    const float bw = bw_ptr[0];
    float si_pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) si_pre[r] = sq[min(td.r0 + G::sub_row(r), td.rlim - 1)];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    G::template run<false>(Zh, Zl, kp, Zh, Zl, kp, td.r0, td.c0, td.rlim, td.clim, kp, lds, nullptr, acc);

    const float c2 = -1.4426950408889634f / (4.f * bw);
    const float wscale = -((td.flags & VGAN_TF_NEG) ? -1.f : 1.f) * 2.f / ((float)n * (float)n * bw);
    const bool store = (td.flags & VGAN_TF_STORE) && Wh != nullptr;
    const bool mirror = store && (td.flags & VGAN_TF_MIRROR);
    float ksum = 0.f;
    // The gradient weights leave through LDS (the staging buffers are free: the main loop ended in a barrier), so that
    // both the direct and the mirrored image go out as 32-byte pieces, four lanes covering one 128-byte row segment,
    // instead of 2-byte (direct) and row-scattered 8-byte (mirror) stores from the MFMA fragment layout.
    constexpr int LDT = 72, LDM = 65;  // Wt[i][j]: b128 reads, 4*LDT = 32 mod 64 banks; WtT[j][i]: scalar, odd stride
    lds_f* Wt = (lds_f*)(float*)lds;
    lds_f* WtT = Wt + 64 * LDT;
    static_assert((64 * LDT + 64 * LDM) * 4 <= G::kLdsBytes, "epilogue images must fit the staging buffers");
    const int lcol = G::sub_col();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lrow = G::sub_row(r), i = td.r0 + lrow;
        const bool ok = jok && (i < td.rlim);
        const float si = si_pre[r];
        const float L = fmaxf(si + sj - 2.f * acc[r], 0.f);
        const float t = __builtin_amdgcn_exp2f(L * c2);
        const float t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
        ksum += ok ? ((t + t2) + (t4 + t8)) + t16 : 0.f;
        const float w = wscale * (((0.25f * t + 0.5f * t2) + (t4 + 2.f * t8)) + 4.f * t16);
        if (store) Wt[lrow * LDT + lcol] = w;
        if (mirror) WtT[lcol * LDM + lrow] = w;
    }
    if (store) {  // uniform per workgroup
        __syncthreads();
        const int q16 = 16 * (threadIdx.x & 3), line = threadIdx.x >> 2;
        auto emit = [&](const float (&w)[16], long rowo, int c_first, int c_lim) {
            unsigned hp[8], lp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned short h0, l0, h1, l1;
                split_bf16(w[2 * e], h0, l0);
                split_bf16(w[2 * e + 1], h1, l1);
                hp[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                lp[e] = (unsigned)l0 | ((unsigned)l1 << 16);
            }
            if (c_first + 15 < c_lim) {  // ldw % 8 == 0 and tile origins are multiples of 64: 32-byte aligned
                uint4* dh = reinterpret_cast<uint4*>(Wh + rowo + c_first);
                uint4* dl = reinterpret_cast<uint4*>(Wl + rowo + c_first);
                dh[0] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                dh[1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
                dl[0] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
                dl[1] = make_uint4(lp[4], lp[5], lp[6], lp[7]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (c_first + e < c_lim) {
                        Wh[rowo + c_first + e] = (unsigned short)(hp[e >> 1] >> (16 * (e & 1)));
                        Wl[rowo + c_first + e] = (unsigned short)(lp[e >> 1] >> (16 * (e & 1)));
                    }
            }
        };
        float w[16];
        if (td.r0 + line < td.rlim) {  // direct image: row i = r0 + line, columns c0 + q16 ..
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 v = *(const lds_f4*)(Wt + line * LDT + q16 + 4 * e);
                w[4 * e] = v[0]; w[4 * e + 1] = v[1]; w[4 * e + 2] = v[2]; w[4 * e + 3] = v[3];
            }
            emit(w, (long)(td.r0 + line - wrow0) * ldw, td.c0 + q16, td.clim);
        }
        if (mirror && td.c0 + line < td.clim) {  // mirrored image: row j = c0 + line, columns r0 + q16 ..
#pragma unroll
            for (int e = 0; e < 16; ++e) w[e] = WtT[line * LDM + q16 + e];
            emit(w, (long)(td.c0 + line - wrow0) * ldw, td.r0 + q16, td.rlim);
        }
    }
    ksum = wave_sum(ksum);
    if (lane == 0) red[wave] = ksum;
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<float4*>(partial)[blockIdx.x] = make_float4((red[0] + red[1]) + (red[2] + red[3]), 0.f, 0.f, 0.f);
}

// ---- the same Gram tile + epilogue on 128x128 tiles (GemmBF3Big: 512 threads, one workgroup per CU) ------------------
__global__ __launch_bounds__(512, 2) void mmd_gram_bf3_big_kernel(const unsigned short* __restrict__ Zh, const unsigned short* __restrict__ Zl,
                                                                  int kp, const float* __restrict__ sq, int n,
                                                                  const float* __restrict__ bw_ptr, const TileDesc* __restrict__ tiles,
                                                                  int ntiles, unsigned short* __restrict__ Wh,
                                                                  unsigned short* __restrict__ Wl, int ldw, int wrow0,
                                                                  float* __restrict__ partial, ColmaxJob cj) {
    using G = GemmBF3Big;
    constexpr int LDT = 136, LDM = 129;  // epilogue images Wt[i][j]: b128 reads, 4*LDT = 32 mod 64 banks; WtT[j][i]: scalar, odd stride
    constexpr int kEpiBytes = (128 * LDT + 128 * LDM) * 4;
    __shared__ __attribute__((aligned(16))) char lds[kEpiBytes > G::kLdsBytes ? kEpiBytes : G::kLdsBytes];
    __shared__ float red[8];
    if ((int)blockIdx.x >= ntiles) {
        const int cb = blockIdx.x - ntiles;
        colmax_partial_body<8>(cj.S, cj.lds, cj.row_offset, cj.part, cj.n, cj.d, cj.from_softmax, cb % cj.nbx, cb / cj.nbx);
        return;
    }
    const TileDesc td = tiles[blockIdx.x];
    // epilogue operands requested before the main loop (see mmd_gram_bf3_kernel)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lcol = G::sub_col();
    const int j = td.c0 + lcol;
    const bool jok = j < td.clim;
    const float sj = sq[min(j, td.clim - 1)];
    const float bw = bw_ptr[0];
    float si_pre[2][16];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int r = 0; r < 16; ++r) si_pre[i2][r] = sq[min(td.r0 + G::sub_row(i2, r), td.rlim - 1)];
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    G::run<false>(Zh, Zl, kp, Zh, Zl, kp, td.r0, td.c0, td.rlim, td.clim, kp, lds, acc);

    const float c2 = -1.4426950408889634f / (4.f * bw);
    const float wscale = -((td.flags & VGAN_TF_NEG) ? -1.f : 1.f) * 2.f / ((float)n * (float)n * bw);
    const bool store = (td.flags & VGAN_TF_STORE) && Wh != nullptr;
    const bool mirror = store && (td.flags & VGAN_TF_MIRROR);
    float ksum = 0.f;
    lds_f* Wt = (lds_f*)(float*)lds;
    lds_f* WtT = Wt + 128 * LDT;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = G::sub_row(i2, r), i = td.r0 + lrow;
            const bool ok = jok && (i < td.rlim);
            const float si = si_pre[i2][r];
            const float L = fmaxf(si + sj - 2.f * acc[i2][r], 0.f);
            const float t = __builtin_amdgcn_exp2f(L * c2);
            const float t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
            ksum += ok ? ((t + t2) + (t4 + t8)) + t16 : 0.f;
            const float w = wscale * (((0.25f * t + 0.5f * t2) + (t4 + 2.f * t8)) + 4.f * t16);
            if (store) Wt[lrow * LDT + lcol] = w;
            if (mirror) WtT[lcol * LDM + lrow] = w;
        }
    if (store) {  // uniform per workgroup
        __syncthreads();
        const int line = threadIdx.x >> 2;  // 128 lines, four threads per line
        auto emit = [&](const float (&w)[16], long rowo, int c_first, int c_lim) {
            unsigned hp[8], lp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned short h0, l0, h1, l1;
                split_bf16(w[2 * e], h0, l0);
                split_bf16(w[2 * e + 1], h1, l1);
                hp[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                lp[e] = (unsigned)l0 | ((unsigned)l1 << 16);
            }
            if (c_first + 15 < c_lim) {
                uint4* dh = reinterpret_cast<uint4*>(Wh + rowo + c_first);
                uint4* dl = reinterpret_cast<uint4*>(Wl + rowo + c_first);
                dh[0] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                dh[1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
                dl[0] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
                dl[1] = make_uint4(lp[4], lp[5], lp[6], lp[7]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (c_first + e < c_lim) {
                        Wh[rowo + c_first + e] = (unsigned short)(hp[e >> 1] >> (16 * (e & 1)));
                        Wl[rowo + c_first + e] = (unsigned short)(lp[e >> 1] >> (16 * (e & 1)));
                    }
            }
        };
        float w[16];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int q16 = 64 * half + 16 * (threadIdx.x & 3);
            if (td.r0 + line < td.rlim) {  // direct image: row i = r0 + line, columns c0 + q16 ..
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 v = *(const lds_f4*)(Wt + line * LDT + q16 + 4 * e);
                    w[4 * e] = v[0]; w[4 * e + 1] = v[1]; w[4 * e + 2] = v[2]; w[4 * e + 3] = v[3];
                }
                emit(w, (long)(td.r0 + line - wrow0) * ldw, td.c0 + q16, td.clim);
            }
            if (mirror && td.c0 + line < td.clim) {  // mirrored image: row j = c0 + line, columns r0 + q16 ..
#pragma unroll
                for (int e = 0; e < 16; ++e) w[e] = WtT[line * LDM + q16 + e];
                emit(w, (long)(td.c0 + line - wrow0) * ldw, td.r0 + q16, td.rlim);
            }
        }
    }
    ksum = wave_sum(ksum);
    if (lane == 0) red[wave] = ksum;
    __syncthreads();
    if (threadIdx.x == 0)
        reinterpret_cast<float4*>(partial)[blockIdx.x] =
            make_float4(((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])), 0.f, 0.f, 0.f);
}


// ---- the same Gram tile + epilogue on 256 x 128 tiles (GemmBF3Wide: 768 threads = 8 consumer + 4 loader waves) -----------
// Tile tables built with tile = 256 (vgan_mmd_build_tiles): rows step 256, columns step 128; the symmetric blocks keep their
// mirrored / counted-twice tiles only OUTSIDE the 256 x 256 squares on the diagonal, whose two tiles are computed in full.
__global__ __launch_bounds__(768, 3) void mmd_gram_bf3_wide_kernel(const unsigned short* __restrict__ Zh, const unsigned short* __restrict__ Zl,
                                                                   int kp, const float* __restrict__ sq, int n,
                                                                   const float* __restrict__ bw_ptr, const TileDesc* __restrict__ tiles,
                                                                   int ntiles, unsigned short* __restrict__ Wh,
                                                                   unsigned short* __restrict__ Wl, int ldw, int wrow0,
                                                                   float* __restrict__ partial, ColmaxJob cj, TailSplit ts, float* __restrict__ rs_part, int ldrs) {
    using G = GemmBF3Wide;
    constexpr int LDT = 132, LDM = 260;  // epilogue images Wt[256][LDT] (direct) and WtT[128][LDM] (mirrored), one at a time
    static_assert(256 * LDT * 4 <= G::kLdsBytes && 128 * LDM * 4 <= G::kLdsBytes, "epilogue images reuse the stage buffers");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float red[12];
    __shared__ int last_part;
    const int nblk = ts.first + (ntiles - ts.first) * ts.parts;  // whole tiles, then `parts` workgroups per tail tile
    if ((int)blockIdx.x >= nblk) {
        const int cb = blockIdx.x - nblk;
        colmax_partial_body<12>(cj.S, cj.lds, cj.row_offset, cj.part, cj.n, cj.d, cj.from_softmax, cb % cj.nbx, cb / cj.nbx);
        return;
    }
    const bool split = (int)blockIdx.x >= ts.first;
    const int tq = split ? ((int)blockIdx.x - ts.first) / ts.parts : 0;
    const int ti = split ? ts.first + tq : (int)blockIdx.x;
    const int part = split ? ((int)blockIdx.x - ts.first) - tq * ts.parts : 0;
    const int k0 = part * ts.kchunk, klen = split ? min(ts.kchunk, kp - k0) : kp;
    const TileDesc td = tiles[ti];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool cons = !G::is_loader();
    // epilogue operands requested before the main loop (see mmd_gram_bf3_kernel); loader threads clamp to the tile's corner
    const float bw = bw_ptr[0];
    float sjv[4], siv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sjv[j] = sq[min(td.c0 + (cons ? G::sub_col(j) : 0), td.clim - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) siv[i][r] = sq[min(td.r0 + (cons ? G::sub_row(i, r) : 0), td.rlim - 1)];
    f32x4w acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};
    G::run<false>(Zh + k0, Zl + k0, kp, Zh + k0, Zl + k0, kp, td.r0, td.c0, td.rlim, td.clim, klen, lds, acc);
    if (split) {
        // One K part of a tail tile: the partial products go to this part's slab (thread-linear: block (i, j) of consumer thread
        // t at 16-byte index (4 i + j) * 512 + t), the LAST part to arrive sums all slabs in part order -- its own included, so
        // the result does not depend on who came last -- and runs the epilogue.  Nobody waits for anybody: no residency
        // assumption.  No fences either: a release / acquire fence is an L2-wide write-back / invalidate per WAVE, and 3 072 of
        // them at the end of a launch whose L2s are full of dirty W lines cost more than the round they were to save (c4: Gram
        // 0.40 -> 0.50 ms, measured).  Instead the slabs move with sc0 sc1 (write-through / coherent) stores and loads, each
        // storing thread drains its stores (vmcnt(0)) before the workgroup's relaxed device-scope ticket: MI355X_MICROARCH.md,
        // "publish-large".
        const __amdgpu_buffer_rsrc_t slabs = __builtin_amdgcn_make_buffer_rsrc(ts.slabs, 0, (int)(kTailSlabs * kTailSlabBytes), 0x00020000);
        constexpr int kCoherent = 17;  // aux: sc0 | sc1
        const int slab0 = tq * ts.parts * (int)kTailSlabBytes;
        if (cons) {
            const int mine = slab0 + part * (int)kTailSlabBytes + threadIdx.x * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), slabs, mine + (4 * i + j) * (G::NCONS * 16), 0, kCoherent);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(ts.tickets + tq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_part = old == ts.parts - 1;
            if (old == ts.parts - 1) __hip_atomic_store(ts.tickets + tq, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
        }
        __syncthreads();
        if (!last_part) return;
        if (cons) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < ts.parts; ++q) {
                const int theirs = slab0 + q * (int)kTailSlabBytes + threadIdx.x * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] += __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(slabs, theirs + (4 * i + j) * (G::NCONS * 16), 0, kCoherent));
            }
        }
    }

    const float c2 = -1.4426950408889634f / (4.f * bw);
    const float wscale = -((td.flags & VGAN_TF_NEG) ? -1.f : 1.f) * 2.f / ((float)n * (float)n * bw);
    const bool store = (td.flags & VGAN_TF_STORE) && Wh != nullptr;
    const bool mirror = store && (td.flags & VGAN_TF_MIRROR);
    float ksum = 0.f;
    if (cons) {  // acc <- gradient weight w, ksum <- kernel sum of the valid pairs
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = (td.r0 + G::sub_row(i, r) < td.rlim) && (td.c0 + G::sub_col(j) < td.clim);
                    const float L = fmaxf(siv[i][r] + sjv[j] - 2.f * acc[i][j][r], 0.f);
                    const float t = __builtin_amdgcn_exp2f(L * c2);
                    const float t2 = t * t, t4 = t2 * t2, t8 = t4 * t4, t16 = t8 * t8;
                    ksum += ok ? ((t + t2) + (t4 + t8)) + t16 : 0.f;
                    acc[i][j][r] = wscale * (((0.25f * t + 0.5f * t2) + (t4 + 2.f * t8)) + 4.f * t16);
                }
    }
    if (store) {  // uniform per workgroup
        lds_f* Wt = (lds_f*)(float*)lds;
        // 16 consecutive weights of one output row -> hi / lo bf16 images (two 16-byte stores each when the run is whole)
        // ... and returns the sum of the stored values (hi + lo as the backward product will read them) over the valid columns
        auto emit = [&](const float (&w)[16], long rowo, int c_first, int c_lim) -> float {
            unsigned hp[8], lp[8];
            float sums[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned short h0, l0, h1, l1;
                split_bf16(w[2 * e], h0, l0);
                split_bf16(w[2 * e + 1], h1, l1);
                hp[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                lp[e] = (unsigned)l0 | ((unsigned)l1 << 16);
                const float v0 = __uint_as_float((unsigned)h0 << 16) + __uint_as_float((unsigned)l0 << 16);
                const float v1 = __uint_as_float((unsigned)h1 << 16) + __uint_as_float((unsigned)l1 << 16);
                sums[e & 3] += (c_first + 2 * e < c_lim ? v0 : 0.f) + (c_first + 2 * e + 1 < c_lim ? v1 : 0.f);
            }
            if (c_first + 15 < c_lim) {
                uint4* dh = reinterpret_cast<uint4*>(Wh + rowo + c_first);
                uint4* dl = reinterpret_cast<uint4*>(Wl + rowo + c_first);
                dh[0] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
                dh[1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
                dl[0] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
                dl[1] = make_uint4(lp[4], lp[5], lp[6], lp[7]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (c_first + e < c_lim) {
                        Wh[rowo + c_first + e] = (unsigned short)(hp[e >> 1] >> (16 * (e & 1)));
                        Wl[rowo + c_first + e] = (unsigned short)(lp[e >> 1] >> (16 * (e & 1)));
                    }
            }
            return (sums[0] + sums[1]) + (sums[2] + sums[3]);
        };
        // row sums of the stored W over one 128-column slot: the eight consecutive lanes that hold a row's chunks of the slot
        auto slot_sum = [&](float v) -> float {
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            return v;
        };
        float w[16];
        if (cons) {  // direct image Wt[row][col]: for a fixed (i, j, r) the 64 lanes hit 64 different banks (4 * LDT = 16 mod 64)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Wt[G::sub_row(i, r) * LDT + G::sub_col(j)] = acc[i][j][r];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 256 * 8; t += G::NTH) {  // all twelve waves store: row r0 + line, columns c0 + 16 q ..
            const int line = t >> 3, q16 = 16 * (t & 7);
            float cs = 0.f;
            if (td.r0 + line < td.rlim && td.c0 + q16 < td.clim) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 v = *(const lds_f4*)(Wt + line * LDT + q16 + 4 * e);
                    w[4 * e] = v[0]; w[4 * e + 1] = v[1]; w[4 * e + 2] = v[2]; w[4 * e + 3] = v[3];
                }
                cs = emit(w, (long)(td.r0 + line - wrow0) * ldw, td.c0 + q16, td.clim);
            }
            if (rs_part != nullptr) {  // (uniform; the trip count is a whole number of waves)
                cs = slot_sum(cs);
                if ((t & 7) == 0 && td.r0 + line < td.rlim) rs_part[(long)(td.c0 >> 7) * ldrs + (td.r0 + line - wrow0)] = cs;
            }
        }
        if (mirror) {  // mirrored image WtT[col][row]: a lane's four r are four consecutive rows -> one 16-byte LDS store per block
            __syncthreads();
            if (cons) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *(lds_f4*)(Wt + G::sub_col(j) * LDM + G::sub_row(i, 0)) = f32x4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            }
            __syncthreads();
            for (int t = threadIdx.x; t < 128 * 16; t += G::NTH) {  // row c0 + line of W, columns r0 + 16 q ..
                const int line = t >> 4, q16 = 16 * (t & 15);
                float cs = 0.f;
                if (td.c0 + line < td.clim && td.r0 + q16 < td.rlim) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4 v = *(const lds_f4*)(Wt + line * LDM + q16 + 4 * e);
                        w[4 * e] = v[0]; w[4 * e + 1] = v[1]; w[4 * e + 2] = v[2]; w[4 * e + 3] = v[3];
                    }
                    cs = emit(w, (long)(td.c0 + line - wrow0) * ldw, td.r0 + q16, td.rlim);
                }
                if (rs_part != nullptr) {  // the row's sixteen chunks span two slots
                    cs = slot_sum(cs);
                    if ((t & 7) == 0 && td.c0 + line < td.clim && td.r0 + q16 < td.rlim)
                        rs_part[(long)((td.r0 + q16) >> 7) * ldrs + (td.c0 + line - wrow0)] = cs;
                }
            }
        }
    }
    ksum = wave_sum(ksum);
    if (lane == 0) red[wave] = ksum;
    __syncthreads();
    if (threadIdx.x == 0)
        reinterpret_cast<float4*>(partial)[ti] =
            make_float4(((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])), 0.f, 0.f, 0.f);
}

// ---- backward: out = 2 (rowsum(W) z - W . Z) * mul, W = Wh + Wl [nr, kn], Z^T = ZTh + ZTl [kp, kn] -------------
// RM: the B operand is Z's ROW-MAJOR split images (Zh, Zl [zrows, kp], what the Gram reads) instead of the transposed copies
// (ZTh, ZTl [kp, kn]); `kn` is then the padded contraction length (columns of W) and `brows` the rows of Zh that exist.
template <int BK, bool RM>
__global__ __launch_bounds__(kBlock, BK == 64 ? 2 : 3) void mmd_backward_bf3_kernel(const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl,
                                                                    int ldw, const unsigned short* __restrict__ ZTh,
                                                                    const unsigned short* __restrict__ ZTl, int kn, int ldb, int brows,
                                                                    const float* __restrict__ Z, int ldz, int wrow0, int nr, int p,
                                                                    int ptiles, const float* __restrict__ mul, int ldmul,
                                                                    const float* __restrict__ mul_shift, float* __restrict__ out, int ldo,
                                                                    int kchunk, long slab_stride, vgan_finalize_job job, int nfin,
                                                                    XXJob xx) {
    using G = GemmBF3<BK>;
    constexpr int kBytes = (RM ? G::kLdsBytesT : G::kLdsBytes) > G::kLdsBytes ? (RM ? G::kLdsBytesT : G::kLdsBytes) : G::kLdsBytes;
    __shared__ __attribute__((aligned(16))) char lds[kBytes];
    __shared__ float rs[64];
    // XCD-aware order as in mmd_backward_kernel: down 4 row panels, then the next feature panel
    const int gx = ptiles, gy = (nr + 63) / 64, total = gx * gy;
    if ((int)blockIdx.x >= total) {  // surplus workgroups of the launch (slab 0 only): the step tail (see vgan_finalize_job), then
        if (blockIdx.y != 0) return;  // X-X tiles of the Gram that found no slot in its own launch (vgan_mmd_backward_bf3_rm_xx)
        const int extra = (int)blockIdx.x - total;
        if (extra < nfin)
            finalize_body(job);
        else if constexpr (BK == 64)
            xx_tile_body(xx, extra - nfin, lds, rs);
        return;
    }
    const int xcd = blockIdx.x % 8, kidx = blockIdx.x / 8;
    const int q = total / 8, r8 = total % 8;
    const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + kidx;
    const int band = t / (4 * gx), rem = t - band * 4 * gx;
    const int rows_in_band = min(4, gy - band * 4);
    const int m0 = (band * 4 + rem % rows_in_band) * 64, n0 = (rem / rows_in_band) * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // split-K slice blockIdx.y of the row-of-W range (whole K tiles); slab = its share of rowsum and of the product
    const int k0 = blockIdx.y * kchunk, klen = min(kchunk, kn - k0);
    out += blockIdx.y * slab_stride;
    // the epilogue's operands (z and the multiplier, clamped addresses) are requested BEFORE the main loop: issued after it
    // they would add one full memory latency to every tile
    const int col = n0 + G::sub_col(), colc = min(col, p - 1);
    float z_pre[16], m_pre[16];
    const float mshift = mul_shift != nullptr ? mul_shift[colc] : 0.f;  // mul is stored centred (see vgan_mmd_backward)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rowc = min(m0 + G::sub_row(r), nr - 1);
        z_pre[r] = Z[(long)(wrow0 + rowc) * ldz + colc];
        m_pre[r] = mul != nullptr ? mul[(long)rowc * ldmul + colc] + mshift : 1.f;
    }
    if (klen > 0) {
        if constexpr (RM)
            G::template run_bt<true>(Wh + k0, Wl + k0, ldw, ZTh + (long)k0 * ldb, ZTl + (long)k0 * ldb, ldb, brows - k0, m0, n0, nr, klen, lds,
                                     rs, acc);
        else
            G::template run<true>(Wh + k0, Wl + k0, ldw, ZTh + k0, ZTl + k0, ldb, m0, n0, nr, gx * 64, klen, lds, rs, acc);
    }
    if (col >= p) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lrow = G::sub_row(r), row = m0 + lrow;
        if (row < nr) {
            const float v = klen > 0 ? 2.f * (rs[lrow] * z_pre[r] - acc[r]) : 0.f;
            out[(long)row * ldo + col] = v * m_pre[r];
        }
    }
}

// ---- the backward product on 128x128 tiles (large problems; same decomposition, GemmBF3Big) ---------------------
template <bool RM>
__global__ __launch_bounds__(512, 2) void mmd_backward_bf3_big_kernel(const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl,
                                                                      int ldw, const unsigned short* __restrict__ ZTh,
                                                                      const unsigned short* __restrict__ ZTl, int kn, int ldb, int brows,
                                                                      const float* __restrict__ Z, int ldz, int wrow0, int nr, int p,
                                                                      int ptiles, const float* __restrict__ mul, int ldmul,
                                                                      const float* __restrict__ mul_shift, float* __restrict__ out, int ldo,
                                                                      int kchunk, long slab_stride, int nb_rows, vgan_finalize_job job) {
    using G = GemmBF3Big;  // nb_rows: rows of ZT that exist (kp); feature rows past it are clamped, their columns discarded
    __shared__ __attribute__((aligned(16))) char lds[RM ? G::kLdsBytesT : G::kLdsBytes];
    __shared__ float rs[128];
    const int gx = ptiles, gy = (nr + 127) / 128, total = gx * gy;
    if ((int)blockIdx.x >= total) {
        if (blockIdx.y == 0) finalize_body(job);
        return;
    }
    // XCD-aware order as in the 64-wide kernel: down 4 row panels, then the next feature panel
    const int xcd = blockIdx.x % 8, kidx = blockIdx.x / 8;
    const int q = total / 8, r8 = total % 8;
    const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + kidx;
    const int band = t / (4 * gx), rem = t - band * 4 * gx;
    const int rows_in_band = min(4, gy - band * 4);
    const int m0 = (band * 4 + rem % rows_in_band) * 128, n0 = (rem / rows_in_band) * 128;
    const int k0 = blockIdx.y * kchunk, klen = min(kchunk, kn - k0);
    out += blockIdx.y * slab_stride;
    // (the epilogue's operands are requested AFTER the main loop here: its K loop is long -- these tiles run from c4 sizes up --
    //  and the 64 registers the early request holds across it spill)
    const int col = n0 + G::sub_col(), colc = min(col, p - 1);
    const float mshift = mul_shift != nullptr ? mul_shift[colc] : 0.f;
    f32x16 acc[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i2][r] = 0.f;
    if (klen > 0) {
        if constexpr (RM)
            G::template run_bt<true>(Wh + k0, Wl + k0, ldw, ZTh + (long)k0 * ldb, ZTl + (long)k0 * ldb, ldb, nb_rows, brows - k0, m0, n0, nr, klen,
                                     lds, acc, rs);
        else
            G::template run<true>(Wh + k0, Wl + k0, ldw, ZTh + k0, ZTl + k0, ldb, m0, n0, nr, nb_rows, klen, lds, acc, rs);
    }
    if (col >= p) return;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        float z_pre[16], m_pre[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowc = min(m0 + G::sub_row(i2, r), nr - 1);
            z_pre[r] = Z[(long)(wrow0 + rowc) * ldz + colc];
            m_pre[r] = mul != nullptr ? mul[(long)rowc * ldmul + colc] + mshift : 1.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = G::sub_row(i2, r), row = m0 + lrow;
            if (row < nr) {
                const float v = klen > 0 ? 2.f * (rs[lrow] * z_pre[r] - acc[i2][r]) : 0.f;
                out[(long)row * ldo + col] = v * m_pre[r];
            }
        }
    }
}


// ---- backward on 256 x 128 output tiles (GemmBF3Wide::run_bt: B = Z's row-major split images; row sums from the loader waves)
__global__ __launch_bounds__(768, 3) void mmd_backward_bf3_wide_kernel(const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl,
                                                                       int ldw, const unsigned short* __restrict__ Bh,
                                                                       const unsigned short* __restrict__ Bl, int kn, int ldb, int brows,
                                                                       const float* __restrict__ Z, int ldz, int wrow0, int nr, int p,
                                                                       int ptiles, const float* __restrict__ mul, int ldmul,
                                                                       const float* __restrict__ mul_shift, float* __restrict__ out, int ldo,
                                                                       int kchunk, long slab_stride, int nb_cols, vgan_finalize_job job,
                                                                       const float* __restrict__ rs_part, int ldrs) {
    using G = GemmBF3Wide;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float rs[256];
    const int gx = ptiles, gy = (nr + 255) / 256, total = gx * gy;
    if ((int)blockIdx.x >= total) {
        if (blockIdx.y == 0) finalize_body(job);
        return;
    }
    // XCD-aware order as in the other backward kernels: down 4 row panels, then the next feature panel
    const int xcd = blockIdx.x % 8, kidx = blockIdx.x / 8;
    const int q = total / 8, r8 = total % 8;
    const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + kidx;
    const int band = t / (4 * gx), rem = t - band * 4 * gx;
    const int rows_in_band = min(4, gy - band * 4);
    const int m0 = (band * 4 + rem % rows_in_band) * 256, n0 = (rem / rows_in_band) * 128;
    const int k0 = blockIdx.y * kchunk, klen = min(kchunk, kn - k0);
    out += blockIdx.y * slab_stride;
    f32x4w acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};
    if (klen > 0 && rs_part != nullptr) {
        // row sums of W over this K range from the Gram launch's per-slot sums (128 columns each: vgan_mmd_gram_bf3, rs_part):
        // 256 consumer threads fold one row each while the loaders bring in the first stages
        if (threadIdx.x < 256) {
            const float* src = rs_part + min(m0 + (int)threadIdx.x, nr - 1);
            const int s0 = k0 >> 7, s1 = (k0 + klen + 127) >> 7;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int sl = s0;
            for (; sl + 4 <= s1; sl += 4) {
                a0 += src[(long)sl * ldrs];
                a1 += src[(long)(sl + 1) * ldrs];
                a2 += src[(long)(sl + 2) * ldrs];
                a3 += src[(long)(sl + 3) * ldrs];
            }
            for (; sl < s1; ++sl) a0 += src[(long)sl * ldrs];
            rs[threadIdx.x] = (a0 + a1) + (a2 + a3);
        }
        G::run_bt<false>(Wh + k0, Wl + k0, ldw, Bh + (long)k0 * ldb, Bl + (long)k0 * ldb, ldb, nb_cols, brows - k0, m0, n0, nr, klen, lds, acc);
    } else if (klen > 0)
        G::run_bt<true>(Wh + k0, Wl + k0, ldw, Bh + (long)k0 * ldb, Bl + (long)k0 * ldb, ldb, nb_cols, brows - k0, m0, n0, nr, klen, lds, acc, rs);
    if (G::is_loader()) return;  // (no barrier below)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + G::sub_col(j);
        if (col >= p) continue;
        const float mshift = mul_shift != nullptr ? mul_shift[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float z_pre[4], m_pre[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowc = min(m0 + G::sub_row(i, r), nr - 1);
                z_pre[r] = Z[(long)(wrow0 + rowc) * ldz + col];
                m_pre[r] = mul != nullptr ? mul[(long)rowc * ldmul + col] + mshift : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = G::sub_row(i, r), row = m0 + lrow;
                if (row < nr) {
                    const float v = klen > 0 ? 2.f * (rs[lrow] * z_pre[r] - acc[i][j][r]) : 0.f;
                    out[(long)row * ldo + col] = v * m_pre[r];
                }
            }
        }
    }
}

}  // namespace vgan

using namespace vgan;

extern "C" int vgan_mmd_bf3_prepare(const float* Z, int ldz, int rows, int p, uint16_t* Zh, uint16_t* Zl, int kp, uint16_t* ZTh,
                                    uint16_t* ZTl, int kn, vgan_stream_t stream) {
    VGAN_CHECK_ARG(Z && Zh && Zl && rows > 0 && p > 0 && ldz >= p && kp >= p && kp % 64 == 0 && aligned16(Zh) && aligned16(Zl));
    VGAN_CHECK_ARG((ZTh == nullptr) == (ZTl == nullptr) && (ZTh == nullptr || (kn >= rows && kn % 64 == 0)));
    dim3 grid(kp / 64, (rows + 63) / 64);
    if (ZTh != nullptr) grid.y = kn / 64;
    const int vec = (ldz % 4 == 0) && aligned16(Z);
    hipLaunchKernelGGL(bf3_prepare_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, Z, ldz, rows, p, Zh, Zl, kp, ZTh, ZTl, kn, vec);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

#ifndef VGAN_TAIL_MAX_PARTS
#define VGAN_TAIL_MAX_PARTS 4
#endif
extern "C" int64_t vgan_mmd_gram_bf3_tail_ws_bytes(void) { return kTailSlabs * kTailSlabBytes + kTailTicketBytes; }

extern "C" int vgan_mmd_gram_bf3(const uint16_t* Zh, const uint16_t* Zl, int kp, const float* sq, int n, const float* bw,
                                 const int32_t* tiles, int ntiles, int tile, uint16_t* Wh, uint16_t* Wl, int ldw, int wrow0,
                                 float* partial, const float* S, int lds, int from_softmax, int row_offset, uint64_t* colpart,
                                 int nrows, int d, void* tail_ws, int64_t tail_ws_bytes, float* rs_part, int ldrs,
                                 vgan_stream_t stream) {
    VGAN_CHECK_ARG(Zh && Zl && sq && bw && tiles && partial && n > 0 && ntiles > 0 && kp > 0 && kp % 64 == 0);
    VGAN_CHECK_ARG((Wh == nullptr) == (Wl == nullptr) && (reinterpret_cast<uintptr_t>(partial) & 15) == 0);
    VGAN_CHECK_ARG(aligned16(Zh) && aligned16(Zl) && (Wh == nullptr || (aligned16(Wh) && aligned16(Wl))));
    ColmaxJob cj{};
    int extra = 0;
    if (S != nullptr) {
        VGAN_CHECK_ARG(colpart && nrows > 0 && d > 0 && lds >= d);
        cj = ColmaxJob{S, reinterpret_cast<unsigned long long*>(colpart), lds, row_offset, nrows, d, from_softmax, (d + 63) / 64};
        extra = cj.nbx * ((nrows + kColChunkRows - 1) / kColChunkRows);
    }
    VGAN_CHECK_ARG(tile == 64 || tile == 128 || tile == 256);
    VGAN_CHECK_ARG(rs_part == nullptr || (tile == 256 && Wh != nullptr && ldrs > 0 && wrow0 >= 0 && n % 128 == 0));
    if (tile == 256) {  // 256 x 128 tiles: 768-thread workgroups holding all but 16 KB of a CU's LDS
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_gram_bf3_wide_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, GemmBF3Wide::kLdsBytes);
        VGAN_CHECK_ARG(attr == hipSuccess);
        // a short last round is split over K (TailSplit, mmd_common.hpp) when the caller lends the workspace
        TailSplit ts{nullptr, nullptr, ntiles, 1, kp};
        if (tail_ws != nullptr) {
            VGAN_CHECK_ARG(tail_ws_bytes >= vgan_mmd_gram_bf3_tail_ws_bytes() && aligned16(tail_ws));
            static const int cus = [] {
                int dev = 0, v = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
                return v;
            }();
            const int slots = cus < kTailSlabs ? cus : kTailSlabs;
            const int r = slots > 0 ? ntiles % slots : 0, nst = kp / GemmBF3Wide::BK;
            int parts = 1;  // the largest of 2, 4 that keeps the tail in one round and a part at least 8 stages long (8 parts measured
                            // slower at c4: 32 tiles x 8 slabs of 128 KB written and read back inside ~20 us)
            while (r > 0 && parts < VGAN_TAIL_MAX_PARTS && 2 * parts * r <= slots && nst / (2 * parts) >= 8) parts *= 2;
            if (parts > 1) {
                const int kchunk = ((nst + parts - 1) / parts) * GemmBF3Wide::BK;
                ts = TailSplit{static_cast<float*>(tail_ws), reinterpret_cast<int*>(static_cast<char*>(tail_ws) + kTailSlabs * kTailSlabBytes),
                               ntiles - r, (kp + kchunk - 1) / kchunk, kchunk};
            }
        }
        const int nblk = ts.first + (ntiles - ts.first) * ts.parts;
        hipLaunchKernelGGL(mmd_gram_bf3_wide_kernel, dim3(nblk + extra), dim3(GemmBF3Wide::NTH), GemmBF3Wide::kLdsBytes, (hipStream_t)stream, Zh,
                           Zl, kp, sq, n, bw, reinterpret_cast<const TileDesc*>(tiles), ntiles, Wh, Wl, ldw, wrow0, partial, cj, ts, rs_part, ldrs);
    } else if (tile == 128)
        hipLaunchKernelGGL(mmd_gram_bf3_big_kernel, dim3(ntiles + extra), dim3(512), 0, (hipStream_t)stream, Zh, Zl, kp, sq, n, bw,
                           reinterpret_cast<const TileDesc*>(tiles), ntiles, Wh, Wl, ldw, wrow0, partial, cj);
    else
        hipLaunchKernelGGL(mmd_gram_bf3_kernel<64>, dim3(ntiles + extra), dim3(kBlock), 0, (hipStream_t)stream, Zh, Zl, kp, sq, n, bw,
                           reinterpret_cast<const TileDesc*>(tiles), ntiles, Wh, Wl, ldw, wrow0, partial, cj);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

// the tile edge vgan_mmd_backward_bf3 runs for this shape: 128-wide tiles (half the L2 -> LDS bytes per flop) once they
// fill the chip at least twice over, unless the caller forces one
extern "C" int vgan_mmd_backward_bf3_tile(int nr, int p, int splits, int tile) {
    if (tile == 64 || tile == 128 || tile == 256) return tile;
    // 256 x 128 tiles (GemmBF3Wide, row-major B only: the caller falls back to 128 for transposed images) once they fill the chip
    const int wide_tiles = ((p + 127) / 128) * ((nr + 255) / 256);
    if (nr >= 256 && wide_tiles * splits >= 256) return 256;
    const int big_tiles = ((p + 127) / 128) * ((nr + 127) / 128);
    return big_tiles * splits >= 512 ? 128 : 64;
}

// common launcher of the two operand forms: rm = 0: B = (ZTh, ZTl) [kp, kn] transposed images; rm = 1: B = (Zh, Zl) [zrows, kp]
static int launch_backward_bf3(int rm, const uint16_t* Wh, const uint16_t* Wl, int ldw, const uint16_t* Bh, const uint16_t* Bl, int kn, int kp,
                               int zrows, const float* Z, int ldz, int wrow0, int nr, int p, const float* mul, int ldmul,
                               const float* mul_shift, float* out, int ldo, int splits, int64_t slab_stride, int tile,
                               const vgan_finalize_job* finalize, vgan_stream_t stream, const vgan_xx_job* xxjob = nullptr,
                               const float* rs_part = nullptr, int ldrs = 0) {
    VGAN_CHECK_ARG(Wh && Wl && Bh && Bl && Z && out && nr > 0 && p > 0 && kn > 0 && kn % 64 == 0 && kp >= p && kp % 64 == 0);
    VGAN_CHECK_ARG(ldw >= kn && ldz >= p && ldo >= p && (mul == nullptr || ldmul >= p) && wrow0 >= 0);
    VGAN_CHECK_ARG(aligned16(Wh) && aligned16(Wl) && aligned16(Bh) && aligned16(Bl) && ldw % 8 == 0);
    VGAN_CHECK_ARG(splits >= 1 && splits <= 64 && (splits == 1 || slab_stride >= (int64_t)nr * ldo));
    VGAN_CHECK_ARG((tile == 0 || tile == 64 || tile == 128 || tile == 256) && (mul_shift == nullptr || mul != nullptr) && (!rm || zrows > 0));
    const int kchunk = ((kn / 64 + splits - 1) / splits) * 64;
    const int ldb = rm ? kp : kn;
    vgan_finalize_job job{};
    if (finalize != nullptr) {
        VGAN_CHECK_ARG(finalize_job_ok(*finalize));
        job = *finalize;
    }
    hipStream_t st = (hipStream_t)stream;
    XXJob xx{};
    if (xxjob != nullptr) {
        const vgan_xx_job& j = *xxjob;
        VGAN_CHECK_ARG(rm && tile == 64 && j.Dh && j.Dl && j.dsq && j.tiles && j.bw && j.partial && j.ntiles > 0 && j.ldd % 64 == 0 &&
                       aligned16(j.Dh) && aligned16(j.Dl) && (reinterpret_cast<uintptr_t>(j.partial) & 15) == 0);
        xx = XXJob{j.Dh, j.Dl, j.dsq, nullptr, nullptr, reinterpret_cast<const TileDesc*>(j.tiles), j.bw, j.partial, j.ldd, 1, 0, j.ntiles, 0};
    }
    const int nfin = finalize != nullptr ? 1 : 0;
    int edge = vgan_mmd_backward_bf3_tile(nr, p, splits, tile);
    if (edge == 256 && !rm) edge = 128;
    if (edge == 256) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_backward_bf3_wide_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, GemmBF3Wide::kLdsBytes);
        VGAN_CHECK_ARG(attr == hipSuccess);
        const int pt = (p + 127) / 128;
        dim3 grid(pt * ((nr + 255) / 256) + nfin, splits);
        // the Gram's per-slot row sums serve when every K range is a whole number of 128-column slots
        if (rs_part != nullptr && !(ldrs >= nr && (splits == 1 || kchunk % 128 == 0))) rs_part = nullptr;
        hipLaunchKernelGGL(mmd_backward_bf3_wide_kernel, grid, dim3(GemmBF3Wide::NTH), GemmBF3Wide::kLdsBytes, st, Wh, Wl, ldw, Bh, Bl, kn, ldb,
                           zrows, Z, ldz, wrow0, nr, p, pt, mul, ldmul, mul_shift, out, ldo, kchunk, (long)slab_stride, kp, job, rs_part, ldrs);
        VGAN_CHECK_LAUNCH();
        return VGAN_OK;
    }
    if (edge == 128) {
        const int pt = (p + 127) / 128;
        dim3 grid(pt * ((nr + 127) / 128) + (finalize != nullptr ? 1 : 0), splits);
        if (rm)
            hipLaunchKernelGGL(mmd_backward_bf3_big_kernel<true>, grid, dim3(512), 0, st, Wh, Wl, ldw, Bh, Bl, kn, ldb, zrows, Z, ldz, wrow0, nr,
                               p, pt, mul, ldmul, mul_shift, out, ldo, kchunk, (long)slab_stride, kp, job);
        else
            hipLaunchKernelGGL(mmd_backward_bf3_big_kernel<false>, grid, dim3(512), 0, st, Wh, Wl, ldw, Bh, Bl, kn, ldb, 0, Z, ldz, wrow0, nr, p,
                               pt, mul, ldmul, mul_shift, out, ldo, kchunk, (long)slab_stride, kp, job);
        VGAN_CHECK_LAUNCH();
        return VGAN_OK;
    }
    const int ptiles = (p + 63) / 64;
    dim3 grid(ptiles * ((nr + 63) / 64) + nfin + xx.ntiles, splits);
    if (rm)
        hipLaunchKernelGGL((mmd_backward_bf3_kernel<64, true>), grid, dim3(kBlock), 0, st, Wh, Wl, ldw, Bh, Bl, kn, ldb, zrows, Z, ldz, wrow0, nr,
                           p, ptiles, mul, ldmul, mul_shift, out, ldo, kchunk, (long)slab_stride, job, nfin, xx);
    else
        hipLaunchKernelGGL((mmd_backward_bf3_kernel<64, false>), grid, dim3(kBlock), 0, st, Wh, Wl, ldw, Bh, Bl, kn, ldb, 0, Z, ldz, wrow0, nr, p,
                           ptiles, mul, ldmul, mul_shift, out, ldo, kchunk, (long)slab_stride, job, nfin, xx);
    VGAN_CHECK_LAUNCH();
    return VGAN_OK;
}

extern "C" int vgan_mmd_backward_bf3(const uint16_t* Wh, const uint16_t* Wl, int ldw, const uint16_t* ZTh, const uint16_t* ZTl, int kn,
                                     int kp, const float* Z, int ldz, int wrow0, int nr, int p, const float* mul, int ldmul,
                                     const float* mul_shift, float* out, int ldo, int splits, int64_t slab_stride, int tile,
                                     const vgan_finalize_job* finalize, vgan_stream_t stream) {
    return launch_backward_bf3(0, Wh, Wl, ldw, ZTh, ZTl, kn, kp, 0, Z, ldz, wrow0, nr, p, mul, ldmul, mul_shift, out, ldo, splits, slab_stride,
                               tile, finalize, stream);
}

extern "C" int vgan_mmd_backward_bf3_rm_xx(const uint16_t* Wh, const uint16_t* Wl, int ldw, int kn, const uint16_t* Zh, const uint16_t* Zl,
                                           int kp, int zrows, const float* Z, int ldz, int wrow0, int nr, int p, const float* mul, int ldmul,
                                           const float* mul_shift, float* out, int ldo, int splits, int64_t slab_stride,
                                           const vgan_finalize_job* finalize, const vgan_xx_job* xx, vgan_stream_t stream) {
    VGAN_CHECK_ARG(xx != nullptr);
    return launch_backward_bf3(1, Wh, Wl, ldw, Zh, Zl, kn, kp, zrows, Z, ldz, wrow0, nr, p, mul, ldmul, mul_shift, out, ldo, splits,
                               slab_stride, 64, finalize, stream, xx);
}

extern "C" int vgan_mmd_backward_bf3_rm(const uint16_t* Wh, const uint16_t* Wl, int ldw, int kn, const uint16_t* Zh, const uint16_t* Zl,
                                        int kp, int zrows, const float* Z, int ldz, int wrow0, int nr, int p, const float* mul, int ldmul,
                                        const float* mul_shift, float* out, int ldo, int splits, int64_t slab_stride, int tile,
                                        const vgan_finalize_job* finalize, const float* rs_part, int ldrs, vgan_stream_t stream) {
    return launch_backward_bf3(1, Wh, Wl, ldw, Zh, Zl, kn, kp, zrows, Z, ldz, wrow0, nr, p, mul, ldmul, mul_shift, out, ldo, splits,
                               slab_stride, tile, finalize, stream, nullptr, rs_part, ldrs);
}
