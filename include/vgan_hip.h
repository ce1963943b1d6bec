/*
 * vgan_hip.h -- C ABI of libvgan_hip.so: the MI355X (gfx950) kernels behind the V-GAN training
 * hot path (reference: jcribeiro98/V-GAN, src/vgan.py:597-621 and src/models/ *.py).
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; each entry point below replaces
 * the ATen op sequence of one reference module call, cited as file:line of the reference checkout.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 (unless typed otherwise), row-major, with an
 *     explicit leading dimension (elements);  the caller owns and allocates every buffer,
 *     including workspaces;  the library keeps no state between calls (the frozen RBF bandwidth
 *     lives in a caller-owned device scalar);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never synchronises
 *     the host, never allocates, and is therefore HIP-graph capturable;
 *   - return value: 0 = VGAN_OK, otherwise an error code; vgan_last_error() gives the text.
 */
#ifndef VGAN_HIP_H
#define VGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGAN_OK 0
#define VGAN_ERR_ARG 1  /* bad shape / null pointer / unsupported configuration */
#define VGAN_ERR_HIP 2  /* a HIP runtime call or launch failed */

#define VGAN_ABI_VERSION 6

typedef void* vgan_stream_t; /* hipStream_t */

int vgan_abi_version(void);
const char* vgan_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Generator_big / Encoder / Decoder Linear layers  (src/models/Generator.py:61-66,
 * src/models/Detector.py:8-13,24-29: nn.Linear == addmm; autograd: two mm per layer)
 *   W is PyTorch layout [out, in].
 * ------------------------------------------------------------------------------------------- */
/* y[n,out] = x[n,in] . W^T + b            (b may be NULL).
 * x may be given as x_nslabs partial slabs x_slab_stride elements apart (a split-K result that was
 * not reduced yet): they are summed, in ascending order, while the operand is staged. */
int vgan_linear_forward(const float* x, int ldx, int x_nslabs, int64_t x_slab_stride, const float* W,
                        int ldw, const float* b, float* y, int ldy, int n, int in, int out,
                        vgan_stream_t stream);
/* dx[n,in] = dy[n,out] . W */
int vgan_linear_backward_input(const float* dy, int lddy, const float* W, int ldw,
                               float* dx, int lddx, int n, int in, int out, vgan_stream_t stream);
/* dW[out,in] = dy^T . x ;  db[out] = column sums of dy   (db may be NULL).
 * splits > 1: the batch rows are cut into `splits` slices and slice s writes its PARTIAL result to
 * dW + s*slab_stride / db + s*slab_stride (elements); sum the slabs with vgan_reduce_slabs.  The
 * contraction runs over the batch while the outputs are small, so slicing is what fills the chip.
 * x may itself be given as x_nslabs unreduced slabs (see vgan_linear_forward). */
int vgan_linear_backward_params(const float* dy, int lddy, const float* x, int ldx, int x_nslabs,
                                int64_t x_slab_stride, float* dW, int lddw, float* db, int n, int in,
                                int out, int splits, int64_t slab_stride, vgan_stream_t stream);
/* vgan_linear_backward_params (db == NULL, no slabs) for the training step's M_4 = dlogits^T [z|1] with the X-X tiles of the
 * Gram (struct vgan_xx_job, declared below; identity row map: Dh / Dl / dsq are the step's own Zh / Zl / sq) riding in the
 * launch as surplus workgroups.  Shape contract: the 16-wave tall-skinny kernel's (out, in, leading dimensions % 4 == 0,
 * aligned bases, n >= 256, at most 32 64x64 output tiles). */
struct vgan_xx_job;
int vgan_linear_backward_params_xx_supported(int n, int in, int out); /* host-side query of that contract (1 / 0) */
int vgan_linear_backward_params_xx(const float* dy, int lddy, const float* x, int ldx, float* dW, int lddw, int n, int in,
                                   int out, const struct vgan_xx_job* xx, vgan_stream_t stream);
/* dst[i] = sum over s < nslabs of src[s*slab_stride + i], in ascending s (bitwise reproducible) */
int vgan_reduce_slabs(const float* src, int64_t slab_stride, int nslabs, float* dst, int64_t count,
                      vgan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * upper_softmax + projection  (src/models/Generator.py:18-22, src/vgan.py:616 `U * batch`)
 *   logits [n,d] -> S = softmax rows, U = (S < 1/d ? S : 1), and the two halves of the stacked
 *   MMD operand Z = [X_batch ; U * X_batch]:  Zx[i] = X_batch[i], Zy[i] = U[i] * X_batch[i]
 *   (row stride ldz; columns d..ldz-1 are not written: pre-zero them to pad the feature dimension)
 *   with squared row norms sqx[n], sqy[n].  S, U are dense [n,d].  Zx/sqx may be NULL.
 *   The shuffled-batch gather of the DataLoader (src/vgan.py:578-584, :599) is fused in:
 *   X_batch row i = data[ rows[(t % row_batches) * row_stride + row_offset + i] ], where `rows` is a
 *   whole epoch's table of shuffled indices and t = *row_cursor is the device-side step counter
 *   (row_cursor == NULL: t = 0;  rows == NULL: X_batch row i = data[row_offset + i]).  U may be NULL.
 *   center (may be NULL): a per-feature constant c[d] subtracted from EVERY row of Z, i.e. Zx[i] = X[i] - c and
 *   Zy[i] = fl(U[i] * X[i]) - c.  cdist(Z,Z)**2 (Mmd_loss_constrained.py:25) is translation invariant, so the loss and
 *   its gradient are unchanged in exact arithmetic, while a common offset of a feature no longer eats the mantissa of
 *   the fp32 (and above all the split-bf16) operands; the step engine passes the data-set mean (vgan_col_mean).
 *   norm_split != 0: sqx/sqy are the norms of the split values hi + lo that vgan_mmd_bf3_prepare will produce from
 *   Zx/Zy (what vgan_mmd_gram_bf3 needs: L = s_i + s_j - 2 g is then |zhat_i - zhat_j|^2 exactly).
 *   chain (may be NULL): Generator_big collapsed into one matrix (vgan_homogeneous_pack / vgan_gemm_grouped): the launch
 *   computes logits = za . At4^T itself (za [n, ldza] = [z | 1 | 0-pad], At4 [d, ldat], e0 = round4(L + 1) columns used), one
 *   wave per batch row, and `logits` is neither read nor written (may be NULL).  Needs the row-in-registers path: d % 4 == 0,
 *   d <= 1024, aligned bases.  Removes the logits launch and 2 x n d x 4 bytes of traffic from the step.
 * ------------------------------------------------------------------------------------------- */
typedef struct vgan_logits_chain {
    const float* za;
    const float* At4;
    int32_t ldza, ldat, e0, pad;
} vgan_logits_chain;
int vgan_mask_project_forward(const float* logits, int ldl, const float* data, int ldd,
                              const int32_t* rows, const uint64_t* row_cursor, int row_batches,
                              int row_stride, int row_offset, float* S, float* U, float* Zx, float* Zy,
                              int ldz, float* sqx, float* sqy, int n, int d, const float* center,
                              int norm_split, const vgan_logits_chain* chain, vgan_stream_t stream);
/* out[j] = mean over the rows of data[:, j] (float64 accumulation, fixed order): the `center` of the calls above. */
int vgan_col_mean(const float* data, int ldd, int rows, int d, float* out, vgan_stream_t stream);
/* out[i, :d] = data[rows[i], :d], sq[i] = |out[i]|^2 (sq may be NULL): batch rows a rank needs as
 * Gram columns but holds no mask for (data-parallel runs keep the data set replicated). */
int vgan_gather_rows(const float* data, int ldd, const int32_t* rows, const uint64_t* row_cursor,
                     int row_batches, int row_stride, int row_offset, float* out, int ldo,
                     float* sq, int n, int d, vgan_stream_t stream);
/* The X half of the (centred) MMD operand alone, for a batch whose mask does not exist yet: out[i] = data[rows[i]] - center
 * (out may be NULL), sq[i] its squared norm (of the split values when norm_split != 0), Zh/Zl[i] (may be NULL) its bf16
 * hi/lo images with row stride kp.  The data-parallel step runs it for the NEXT batch while the gradient all-reduce is in
 * flight; the X-X tiles of the next Gram (sums only) then run behind the collective as well. */
int vgan_gather_rows_split(const float* data, int ldd, const int32_t* rows, const uint64_t* row_cursor,
                           int row_batches, int row_stride, int row_offset, const float* center, float* out,
                           int ldo, float* sq, int norm_split, uint16_t* Zh, uint16_t* Zl, int kp, int n, int d,
                           vgan_stream_t stream);
/* dlogits = softmax-Jacobian( [S < 1/d] * (gU + penalty_grad) ), the autograd of Generator.py:19-21.
 * colkey (may be NULL): packed column arg-max keys from vgan_colmax; row r of column j gets
 * -pen_weight/d added when it holds column j's maximum (topk(U,1,0), Mmd_loss_constrained.py:50). */
/* gU may be given as `nslabs` partial slabs `slab_stride` elements apart (split-K output of
 * vgan_mmd_backward); they are summed in ascending order inside the kernel. */
int vgan_mask_backward(const float* gU, int ldg, int nslabs, int64_t slab_stride, const float* S,
                       int lds, const uint64_t* colkey, float pen_weight, int row_offset,
                       float* dlogits, int ldo, int n, int d, vgan_stream_t stream);
/* colkey[j] = max over rows of pack(U[i,j], row_offset + i)  (value in the high 32 bits, ~row in
 * the low 32 bits; lowest row wins ties).  from_softmax != 0: the input is S and U is derived from
 * it; otherwise the input is U itself (must be > 0).  part: workspace [chunks*d] u64 with
 * chunks = vgan_colmax_chunks(n). */
int vgan_colmax_chunks(int n);
int vgan_colmax(const float* S, int lds, int from_softmax, int row_offset, uint64_t* part,
                uint64_t* colkey, int n, int d, vgan_stream_t stream);
/* first half of vgan_colmax only: per-chunk keys into part[chunks*d] (finished by vgan_mmd_finalize) */
int vgan_colmax_partial(const float* S, int lds, int from_softmax, int row_offset, uint64_t* part,
                        int n, int d, vgan_stream_t stream);
/* dense U from S (for callers that need the mask tensor itself) */
int vgan_mask_from_softmax(const float* S, int lds, float* U, int ldu, int n, int d, vgan_stream_t stream);
/* plain row softmax -> upper_softmax for a dense generator output (no projection) */
int vgan_upper_softmax_forward(const float* logits, int ldl, float* S, float* U, int n, int d,
                               vgan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * RBF + MMDLossConstrained  (src/models/Mmd_loss_constrained.py:16-26, 42-50)
 *
 * The 2n x 2n kernel matrix is never materialised.  The Gram tiles g = Z_I . Z_J^T run on the
 * fp32 MFMA; the epilogue forms L = |z_i|^2 + |z_j|^2 - 2g (clamped at 0), t = exp(-L/(4 bw)) and
 * the five-bandwidth sum K = t + t^2 + t^4 + t^8 + t^16, accumulates the block sums, and writes
 *     Wg[i - wrow0, j] = sgn(i,j) * (2/n^2) * dK/dL        (sgn = +1 same half, -1 across halves)
 * for the rows whose gradient is needed, so that  dZ_i = 2 (rowsum(Wg_i) z_i - (Wg . Z)_i).
 *
 * Work is described by a tile table (host-built by vgan_mmd_build_tiles, uploaded by the caller):
 * each entry is 8 int32 {r0, c0, rlim, clim, flags, 0,0,0}.
 * ------------------------------------------------------------------------------------------- */
#define VGAN_TILE 64           /* Gram tile edge of the fp32 kernels; the split-bf16 Gram also has a 128 variant */
#define VGAN_TILE_INTS 8
#define VGAN_TF_SLOT_MASK 3    /* 0 = XX, 1 = XY, 2 = YY  block sum the tile contributes to */
#define VGAN_TF_TWICE 4        /* off-diagonal tile of a symmetric block: counted twice */
#define VGAN_TF_STORE 8        /* write Wg tile */
#define VGAN_TF_MIRROR 16      /* also write the transposed tile (symmetric block) */
#define VGAN_TF_NEG 32         /* sgn = -1 (rows and columns in different halves) */

/* grad_mode: 0 = no gradient (sums only), 1 = gradient for the Y rows only (Wg is [n, 2n],
 * wrow0 = n), 2 = gradient for all rows (Wg is [2n, 2n], wrow0 = 0).
 * Row-sharded data parallel: rank `rank` of `world` owns rows [rank*n/world, (rank+1)*n/world)
 * of each half; its table covers exactly the pairs (own row, any column), without symmetry.
 * tile: edge of the square tiles, 64 (every kernel) or 128 (vgan_mmd_gram_bf3 only), or 256 = tiles of 256 rows x 128 columns
 * (vgan_mmd_gram_bf3's loader-wave kernel; a symmetric block then keeps the two tiles of each 256 x 256 diagonal square whole and
 * mirrors / counts twice only the tiles outside it).
 * Row-sharded ranks (world > 1): (own Y rows) x (all columns) for the XY and YY blocks -- inside the rank's own diagonal YY block
 * the upper triangle with mirrored stores, when the block sits on the tile grid -- and, for the X-X block (sums only, no row
 * ownership), every world-th tile of the WHOLE block's upper triangle.
 * Returns the number of tiles (or -1 if cap is too small); out may be NULL to query the count. */
int vgan_mmd_build_tiles(int n, int grad_mode, int rank, int world, int tile, int32_t* out, int cap);
/* Re-applies the XCD-aware launch order (Morton curve, dealt to the 8 XCDs in contiguous chunks) to a table the caller
 * filtered or re-assembled on the host, e.g. the X-X tiles split off for the launch that overlaps the gradient all-reduce. */
int vgan_mmd_order_tiles(int32_t* tiles, int count, int tile);

/* partial[tiles*4] (float): per tile {sum K, sum L, 0, 0}.  calibrate != 0: only sum L is
 * produced (first-call bandwidth, Mmd_loss_constrained.py:16-20) and bw/Wg are not touched. */
int vgan_mmd_gram(const float* Z, int ldz, const float* sq, int n, int p, const float* bw,
                  const int32_t* tiles, int ntiles, int calibrate,
                  float* Wg, int ldw, int wrow0, float* partial, vgan_stream_t stream);
/* vgan_mmd_gram (calibrate = 0) for an RBF with other than the reference's defaults -- RBF(n_kernels, mul_factor),
 * Mmd_loss_constrained.py:7-13: K = sum_k exp(-L / (bw * multipliers[k])), one exp per kernel (the default
 * n_kernels = 5, mul_factor = 2 runs the one-exp squaring chain of vgan_mmd_gram).  multipliers is a HOST array of
 * n_kernels <= VGAN_RBF_MAX_KERNELS positive floats (mul_factor ** (k - n_kernels // 2)). */
#define VGAN_RBF_MAX_KERNELS 8
int vgan_mmd_gram_general(const float* Z, int ldz, const float* sq, int n, int p, const float* bw,
                          const int32_t* tiles, int ntiles, const float* multipliers, int n_kernels,
                          float* Wg, int ldw, int wrow0, float* partial, vgan_stream_t stream);
/* vgan_mmd_gram (calibrate = 0) and vgan_colmax_partial in ONE launch: the column arg-max cells run in surplus
 * workgroups behind the Gram tiles (they are independent of each other, and the Gram grid leaves CUs idle in its
 * tail).  Arguments as in the two separate calls; colpart has vgan_colmax_chunks(nrows)*d entries. */
int vgan_mmd_gram_colmax(const float* Z, int ldz, const float* sq, int n, int p, const float* bw,
                         const int32_t* tiles, int ntiles, float* Wg, int ldw, int wrow0, float* partial,
                         const float* S, int lds, int from_softmax, int row_offset, uint64_t* colpart,
                         int nrows, int d, vgan_stream_t stream);
/* stats[4] (double): {Sxx, Sxy, Syy, sumL} += reduction of partial[] by tile slot (deterministic).
 * zero_first != 0 clears stats before accumulating. */
int vgan_mmd_reduce(const float* partial, const int32_t* tiles, int ntiles, double* stats,
                    int zero_first, vgan_stream_t stream);
/* bw[0] = stats[3] / (N^2 - N), N = 2n   (Mmd_loss_constrained.py:18-19) */
int vgan_mmd_set_bandwidth(const double* stats, int n, float* bw, vgan_stream_t stream);
/* loss[0] = (Sxx - 2 Sxy + Syy)/n^2 + weight * mean_j(1 - colmax_j)   (Mmd_loss_constrained.py:47-50)
 * colkey may be NULL (weight term skipped).  loss_accum (may be NULL) += loss * accum_scale;
 * step_counter (may be NULL) += 1  (device-side step index for the Philox noise stream). */
int vgan_mmd_loss(const double* stats, const uint64_t* colkey, int n, int d, float weight,
                  float* loss, float* loss_accum, float accum_scale, uint64_t* step_counter,
                  vgan_stream_t stream);
/* Single-rank step tail in one launch = vgan_mmd_reduce (zero_first) + the second half of vgan_colmax
 * + vgan_mmd_loss: partial[] -> stats[4]; colpart[chunks*d] -> colkey[d] (colpart may be NULL: no
 * penalty term); loss / loss_accum / step_counter as in vgan_mmd_loss. */
int vgan_mmd_finalize(const float* partial, const int32_t* tiles, int ntiles, const uint64_t* colpart,
                      int chunks, uint64_t* colkey, int n, int d, float weight, double* stats,
                      float* loss, float* loss_accum, float accum_scale, uint64_t* step_counter,
                      vgan_stream_t stream);
/* The same step tail as a job that rides in a backward launch (vgan_mmd_backward / vgan_mmd_backward_bf3,
 * argument `finalize`, NULL = none): ONE extra workgroup of that launch does what vgan_mmd_finalize does, off
 * the critical path (its outputs -- colkey, the loss bookkeeping -- are first needed by the kernel AFTER the
 * backward product).  Field meaning as the arguments of vgan_mmd_finalize. */
typedef struct vgan_finalize_job {
    const float* partial;
    const int32_t* tiles;
    const uint64_t* colpart;
    uint64_t* colkey;
    double* stats;
    float* loss;
    float* loss_accum;
    uint64_t* step_counter;
    int32_t ntiles, chunks, n, d;
    float weight, accum_scale;
    /* mode 0: the whole tail.  Some X-X tiles of the Gram may be computed LATER in the step than the launch the tail rides in
     * (vgan_linear_backward_params_xx); the tail is then split: mode 1 = everything over the tiles [0, ntiles_main) (the Gram
     * launch's: all XY / YY tiles and any X-X tiles it had room for) -- column keys, block sums, the step counter; the loss so
     * far is parked in stats[3], the X-X sum so far in stats[0] -- and mode 2 = the X-X sums of the late tiles
     * [ntiles_main, ntiles), the loss and its accumulator (rides in vgan_gemm_grouped_ex, `fold`). */
    int32_t mode, ntiles_main;
} vgan_finalize_job;
/* dZ[i - wrow0, :] = 2 (rowsum(Wg_i) z_i - Wg_i . Z) for the nr rows starting at wrow0;
 * if mul != NULL the result is multiplied elementwise by mul[i - wrow0, :] (the `U * batch`
 * product rule: gU = dY * X).  Z is [ncols, p], Wg is [nr, ncols].
 * splits > 1: the contraction over the Z rows is cut into `splits` slices and slice s writes its
 * PARTIAL result to out + s*slab_stride (the result is linear in the slice sums); the consumer adds
 * the slabs (vgan_mask_backward does, or vgan_reduce_slabs).
 * mul_shift (may be NULL; needs mul): per-column constant added to mul, for callers whose Z (and with it the X rows
 * passed as `mul`) is stored centred: mul + mul_shift is then the batch itself.  Z may be centred freely: the
 * expression rowsum(Wg_i) z_i - Wg_i . Z = sum_j Wg_ij (z_i - z_j) is translation invariant. */
int vgan_mmd_backward(const float* Wg, int ldw, const float* Z, int ldz, int wrow0, int nr,
                      int ncols, int p, const float* mul, int ldmul, const float* mul_shift, float* out,
                      int ldo, int splits, int64_t slab_stride, const vgan_finalize_job* finalize,
                      vgan_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * Split-bf16 ("bf16x3") variants of the two dense contractions, for large problems (opt-in precision
 * mode): every operand z = hi + lo with hi = bf16(z), lo = bf16(z - hi), products hi.hi' + hi.lo' + lo.hi'
 * on the bf16 MFMA (16x the fp32 MFMA rate) with fp32 accumulation: ~3e-7 relative on a Gram entry at
 * K = 784.  The fused epilogues are those of the fp32 kernels.
 * ------------------------------------------------------------------------------------------- */
/* Z [rows, ldz] (first p columns) -> Zh, Zl [rows, kp] bf16 bit patterns (kp = p rounded up to 64, pad = 0)
 * and, unless NULL, the transposed images ZTh, ZTl [kp, kn] (kn = rows rounded up to 64, pad = 0). */
int vgan_mmd_bf3_prepare(const float* Z, int ldz, int rows, int p, uint16_t* Zh, uint16_t* Zl, int kp,
                         uint16_t* ZTh, uint16_t* ZTl, int kn, vgan_stream_t stream);
/* vgan_mmd_gram_colmax on the split operands; the gradient weights leave as a bf16 hi/lo pair Wh, Wl
 * [nr, ldw] (ldw >= 2n rounded up to 64; columns >= 2n must be pre-zeroed).  S may be NULL (no column job).
 * tile = the edge the table was built with: 64, 128 (512-thread workgroups, half the L2->LDS bytes per flop; pays when the
 * table still has >~ 128 tiles) or 256 (256 x 128 tiles, 768-thread workgroups of 8 consumer + 4 loader waves, three K stages
 * of 32 in LDS, v_mfma_f32_16x16x32_bf16: c4 / c5 sizes).
 * tail_ws (may be NULL; read only with tile = 256): device workspace of at least vgan_mmd_gram_bf3_tail_ws_bytes() bytes,
 * 16-byte aligned, whose last 4 096 bytes are ZERO before the first launch that uses it (the launches keep them zero) and
 * which no other launch uses concurrently.  With it, a table whose last round would leave at least half of the CUs idle (one
 * 768-thread workgroup holds a CU: ntiles mod CUs <= CUs / 2) has the tiles of that round computed by 2 or 4 workgroups
 * each, split over K; the partial products meet in the workspace and the last workgroup to arrive finishes the tile (sums in
 * part order: deterministic; nobody waits on anybody).  Results then differ from the unsplit launch by fp32 summation order
 * in those tiles only.
 * rs_part (may be NULL; tile = 256, n a multiple of 128, W stored): float [ceil(2n / 128), ldrs], ldrs >= rows of W.  Every
 * storing tile also leaves rs_part[c / 128][i - wrow0] = sum of the stored (hi + lo) weights of row i over the tile's columns
 * c .. c + 127 -- the row sums vgan_mmd_backward_bf3_rm needs, taken from the epilogue's registers.  Cells of slots no tile
 * covers are not written (zero them once). */
int64_t vgan_mmd_gram_bf3_tail_ws_bytes(void);
int vgan_mmd_gram_bf3(const uint16_t* Zh, const uint16_t* Zl, int kp, const float* sq, int n, const float* bw,
                      const int32_t* tiles, int ntiles, int tile, uint16_t* Wh, uint16_t* Wl, int ldw,
                      int wrow0, float* partial, const float* S, int lds, int from_softmax, int row_offset,
                      uint64_t* colpart, int nrows, int d, void* tail_ws, int64_t tail_ws_bytes,
                      float* rs_part, int ldrs, vgan_stream_t stream);
/* vgan_mmd_backward on the split operands: out = 2 (rowsum(W) z - W . Z) * mul with W = Wh + Wl [nr, kn]
 * and Z^T = ZTh + ZTl [kp, kn]; Z (fp32) is only read by the epilogue.  splits / slab_stride as in
 * vgan_mmd_backward (slabs of out, summed by the consumer in slab order); mul_shift as there.
 * tile: 0 = chosen by the library (256 x 128 loader-wave tiles once they fill the chip -- row-major B operand only --, else
 * 128-wide tiles once they fill it twice over, else 64), or 64 / 128 / 256 to force one. */
/* host-side query (no launch): the tile edge (64, 128 or 256) vgan_mmd_backward_bf3_rm uses for this shape and `tile` argument */
int vgan_mmd_backward_bf3_tile(int nr, int p, int splits, int tile);
int vgan_mmd_backward_bf3(const uint16_t* Wh, const uint16_t* Wl, int ldw, const uint16_t* ZTh,
                          const uint16_t* ZTl, int kn, int kp, const float* Z, int ldz, int wrow0, int nr,
                          int p, const float* mul, int ldmul, const float* mul_shift, float* out, int ldo,
                          int splits, int64_t slab_stride, int tile, const vgan_finalize_job* finalize,
                          vgan_stream_t stream);
/* The same backward product reading Z's ROW-MAJOR split images Zh, Zl [zrows, kp] -- the ones vgan_mmd_gram_bf3 reads -- so
 * that no transposed copy of Z has to be produced: the B fragments (8 consecutive contraction indices per lane) come out of a
 * row-major LDS image through ds_read_b64_tr_b16.  kn = the padded contraction length (columns of Wh / Wl, a multiple of 64,
 * >= zrows; columns >= zrows of W must be zero).  Everything else as vgan_mmd_backward_bf3.
 * rs_part (may be NULL): the per-slot row sums a tile-256 vgan_mmd_gram_bf3 launch left beside W (rs_part [ceil(kn / 128), ldrs],
 * ldrs >= nr).  The 256 x 128 kernel then folds them instead of having its loader waves sum the W rows from LDS (-5 % of the
 * launch at c5); ignored by the other tile sizes and when a K split is not a whole number of 128-column slots. */
int vgan_mmd_backward_bf3_rm(const uint16_t* Wh, const uint16_t* Wl, int ldw, int kn, const uint16_t* Zh,
                             const uint16_t* Zl, int kp, int zrows, const float* Z, int ldz, int wrow0, int nr,
                             int p, const float* mul, int ldmul, const float* mul_shift, float* out, int ldo,
                             int splits, int64_t slab_stride, int tile, const vgan_finalize_job* finalize,
                             const float* rs_part, int ldrs, vgan_stream_t stream);
/* vgan_mmd_backward_bf3_rm on 64-wide tiles with a few X-X tiles of the Gram (struct vgan_xx_job; identity row map) riding in
 * the launch as surplus workgroups: the backward launch of the training step fills 416 of the chip's 512 workgroup slots for
 * 25 us, so up to ~90 eight-microsecond tiles cost it nothing.  The tiles' partial sums are complete when the launch is;
 * a `finalize` job in the same launch must therefore not cover them (vgan_finalize_job.mode 1 / 2). */
int vgan_mmd_backward_bf3_rm_xx(const uint16_t* Wh, const uint16_t* Wl, int ldw, int kn, const uint16_t* Zh,
                                const uint16_t* Zl, int kp, int zrows, const float* Z, int ldz, int wrow0, int nr,
                                int p, const float* mul, int ldmul, const float* mul_shift, float* out, int ldo,
                                int splits, int64_t slab_stride, const vgan_finalize_job* finalize,
                                const struct vgan_xx_job* xx, vgan_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * Grouped small products: up to VGAN_GEMM_MAX_GROUP independent row-major GEMMs in one launch.
 * Generator_big (src/models/Generator.py:61-66) has no activation between its Linear layers, so its
 * forward/backward is a chain of small matrix products (see vgan_homogeneous_pack); products of one
 * dependency level share a launch.  kind: NN C[m,n] = A[m,k] . B[k,n];  NT C = A[m,k] . B[n,k]^T;
 * TN C = A[k,m]^T . B[k,n].  All operands row-major with leading dimensions lda / ldb / ldc.
 * splitk > 1: the contraction is cut into splitk slices run by different workgroups (for a long contraction over few
 * output tiles); slice s writes its PARTIAL product to slab s of C, slabs m * ldc floats apart, and the caller sums the
 * slabs in fixed order (vgan_reduce_slabs) -- deterministic, so data-parallel replicas stay bit-identical.  0 / 1: no split.
 * ------------------------------------------------------------------------------------------- */
#define VGAN_GEMM_MAX_GROUP 4
#define VGAN_GEMM_NN 0
#define VGAN_GEMM_NT 1
#define VGAN_GEMM_TN 2
#define VGAN_GEMM_NT_NT 3 /* two products in one tile: C[m,n] = (A[m,k] . B[k2,k]^T) . D[n,k2]^T -- a 64 x 64 tile of C first forms its 64
                           * rows of H = A . B^T ([64, k2], into `scratch`: ceil(m/64) * ceil(n/64) regions of 64 * round4(k2)
                           * floats, one per tile) and then multiplies them with D.  For a dependent pair of small products whose
                           * second would otherwise cost a launch of its own (the logits of the collapsed generator: T = ([z|1] .
                           * Wt_1^T) . Wt_2^T rides with the first level of chain products, src/models/Generator.py:61-66). */
typedef struct vgan_gemm_problem {
    const float* a;
    const float* b;
    float* c;
    int32_t kind, m, n, k, lda, ldb, ldc, splitk;
    const float* d;   /* VGAN_GEMM_NT_NT only */
    float* scratch;   /* VGAN_GEMM_NT_NT only */
    int32_t ldd, k2;  /* VGAN_GEMM_NT_NT only: D [n, k2] row-major */
} vgan_gemm_problem;
int vgan_gemm_grouped(const vgan_gemm_problem* problems, int count, vgan_stream_t stream);
/* The same launch with work riding in it (each part optional; a dependent launch costs ~5 us whatever its size, so the
 * tail of the step shares launches):
 *   copy      dst[i] = src[i], i < copy_count (a snapshot a later launch of the step reads while its source is updated);
 *   adadelta  != 0: problem i's output C_i is the packed gradient [dW | db] of layer[i] (rows < out, columns <= in, column
 *             `in` = bias) and the torch.optim.Adadelta update (src/vgan.py:567-568, :619; rule of vgan_adadelta_step) runs in
 *             the product's epilogue: the flat parameter p[off_w + row*in + col] / p[off_b + row], its state, and the packed
 *             weight w_packed[row*ldp + col] are updated in place (C_i is still written).  None of the launch's operands
 *             may alias an updated w_packed.  g_extra != NULL: one more layer, layer[count], whose packed gradient already
 *             sits in memory (row stride ld_extra) is updated element-wise by surplus workgroups;
 *   noise     next_noise != NULL: the next step's noise draw, as in vgan_adadelta_step_packed;
 *   fold      a vgan_finalize_job run by one surplus workgroup (the late half of a split step tail). */
typedef struct vgan_adadelta_layer {
    float* w_packed;
    int64_t off_w, off_b;
    int32_t ldp, out, in, pad;
} vgan_adadelta_layer;
typedef struct vgan_grouped_extras {
    const float* copy_src;
    float* copy_dst;
    int64_t copy_count;
    int32_t adadelta, pad;
    float* p;
    float* sq_avg;
    float* acc_delta;
    float lr, rho, eps, weight_decay, grad_scale;
    int32_t ld_extra;
    vgan_adadelta_layer layer[VGAN_GEMM_MAX_GROUP + 1];
    const float* g_extra;
    float* next_noise;
    int32_t noise_rows, noise_cols, noise_ld, noise_ones_col;
    uint64_t seed;
    const uint64_t* step_counter;
    const vgan_finalize_job* fold; /* NULL, or a step-tail job (normally mode 2) run by one surplus workgroup */
} vgan_grouped_extras;
int vgan_gemm_grouped_ex(const vgan_gemm_problem* problems, int count, const vgan_grouped_extras* extras,
                         vgan_stream_t stream);

/* vgan_mask_project_forward fused with vgan_mmd_bf3_prepare for the training step: from logits [n, d] and the
 * batch rows it writes S [n, d], Z = [X ; U*X] ([2n, ldz] fp32), sq [2n] and the split images Zh, Zl [2n, kp],
 * ZTh, ZTl [kp, kn] of Z (pad regions are not touched: pre-zeroed by the caller).  Shape contract: d % 4 == 0,
 * d <= 1024, n % 8 == 0, leading dimensions % 4 == 0, 16-byte aligned bases; otherwise use the two calls
 * (with norm_split = 1).  center as in vgan_mask_project_forward; sq holds the norms of the split values.
 * ZTh / ZTl may both be NULL (callers of vgan_mmd_backward_bf3_rm need no transposed images).
 * write_x == 0 (needs ZTh == NULL): the X half of Z, sq, Zh, Zl is left alone -- vgan_gather_rows_split has already produced
 * it for this batch.
 * xx (may be NULL; needs ZTh == NULL): the X-X tiles of the Gram ride in this launch as surplus workgroups.  They only
 * feed the block sum of the reported loss and depend on nothing the step computes: their operand is this batch's rows of
 * the data set, gathered by the same index table from split images prepared once per fit -- Dh, Dl [data rows, ldd] and dsq
 * (vgan_gather_rows_split over the whole data set with norm_split = 1).  tiles / ntiles: the X-X part of the tile table
 * (vgan_mmd_build_tiles, flags slot 0); partial: where their sums go (4 floats per tile, the layout vgan_mmd_finalize folds);
 * bw: the frozen bandwidth.  The Gram launch then covers the XY and YY tiles only.
 * chain (may be NULL; needs ZTh == NULL): as in vgan_mask_project_forward. */
typedef struct vgan_xx_job {
    const uint16_t* Dh;
    const uint16_t* Dl;
    const float* dsq;
    const int32_t* tiles;
    const float* bw;
    float* partial;
    int32_t ldd, ntiles;
} vgan_xx_job;
int vgan_mask_project_forward_bf3(const float* logits, int ldl, const float* data, int ldd, const int32_t* rows,
                                  const uint64_t* row_cursor, int row_batches, int row_stride, float* S, float* Z,
                                  int ldz, float* sq, uint16_t* Zh, uint16_t* Zl, int kp, uint16_t* ZTh,
                                  uint16_t* ZTl, int kn, int n, int d, const float* center, int write_x,
                                  const vgan_xx_job* xx, const vgan_logits_chain* chain, vgan_stream_t stream);
/* squared row norms sq[r] = |Z_r|^2 (for callers that assemble Z themselves) */
int vgan_row_sqnorm(const float* Z, int ldz, float* sq, int rows, int p, vgan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * torch.optim.Adadelta over one flat parameter buffer  (src/vgan.py:567-568, :619)
 *   g += wd*p; v = rho v + (1-rho) g^2; delta = sqrt(a+eps)/sqrt(v+eps) g; a = rho a + (1-rho) delta^2;
 *   p -= lr*delta.   grad_scale multiplies g first (1 for plain training).  g may be `nslabs` split-K
 *   slabs `slab_stride` elements apart (see vgan_linear_backward_params): summed in ascending order.
 * ------------------------------------------------------------------------------------------- */
int vgan_adadelta_step(float* p, const float* g, int nslabs, int64_t slab_stride, float* sq_avg,
                       float* acc_delta, int64_t count, float lr, float rho, float eps,
                       float weight_decay, float grad_scale, vgan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Noise feed  (src/vgan.py:610 `noise_tensor.normal_()`): standard normals from a counter-based
 * Philox4x32-10 stream + Box-Muller, keyed by (seed, *step_counter); replay-safe under HIP graphs.
 * ------------------------------------------------------------------------------------------- */
/* z is [rows, cols] with row stride ld; element (r,c) is draw number r*cols + c of the stream, so the
 * values do not depend on ld.  ones_col >= cols (or -1): column that is set to 1.0 in every row (the
 * homogeneous coordinate of the collapsed generator chain). */
int vgan_noise_normal(float* z, int rows, int cols, int ld, int ones_col, uint64_t seed,
                      const uint64_t* step_counter, uint64_t stream_id, vgan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Generator_big has NO activation between its four Linear layers (src/models/Generator.py:61-66),
 * so in homogeneous coordinates the chain is a product of matrices Wt_k = [[W_k, b_k],[0, 1]]:
 *   logits = [z|1] . (Wt_4 Wt_3 Wt_2 Wt_1)^T   and   dWt_k = (Wt_{k+1..4}^T dlogits^T [z|1]) . (Wt_{k-1..1})^T,
 * i.e. every product has an inner or outer dimension of L+1 instead of the batch: ~0.2 GFLOP instead
 * of 2.5 GFLOP per step at d=784.  The products run on vgan_linear_*; this entry point moves the
 * parameters/gradients between the PyTorch layout and the packed one in a single launch.
 * desc: device table, 8 int64 per layer {W ptr, b ptr, packed ptr, out, in, ldw, ldp, 0};
 * packed is [out+1, in+1] with row stride ldp.  unpack != 0: packed -> (W, b) (rows < out only).
 * max_elems: the largest (out+1)*(in+1) in the table (sizes the grid). */
int vgan_homogeneous_pack(const int64_t* desc, int count, int max_elems, int unpack, vgan_stream_t stream);
/* Adadelta for that chain without pack/unpack launches: the gradient of flat element i is
 * g_packed[pmap[i]] and the updated parameter is also stored to w_packed[pmap[i]] (pmap[i] < 0: layout
 * padding, skipped).  Same update rule as vgan_adadelta_step.
 * next_noise != NULL: the kernel also draws the noise of the NEXT step (vgan_noise_normal with
 * stream_id 0 and the current value of *step_counter, which the loss kernel has already advanced),
 * saving the separate noise launch at the head of every step. */
int vgan_adadelta_step_packed(float* p, const int32_t* pmap, const float* g_packed, float* w_packed,
                              float* sq_avg, float* acc_delta, int64_t count, float lr, float rho,
                              float eps, float weight_decay, float grad_scale, float* next_noise,
                              int noise_rows, int noise_cols, int noise_ld, int noise_ones_col,
                              uint64_t seed, const uint64_t* step_counter, vgan_stream_t stream);

/* sum of squared differences: out[0] (+)= scale * sum((a-b)^2)  -- `__distance(x,y,'L2')`,
 * src/vgan.py:58-59 (one workgroup; for reporting-sized inputs). */
int vgan_mse(const float* a, int lda, const float* b, int ldb, int n, int d, float scale,
             float* out, int accumulate, vgan_stream_t stream);

/* The same term with its gradient, for VGAN.fit's detector loss (src/vgan.py:276-277): one pass over [n, d],
 * part[b] (b < ceil(n/4)) = float64 partial sums of (pred - target)^2, g = gscale * (pred - target).
 * vgan_sum_f64 folds such partials: out[0] (+)= scale * sum(in[0..count)), fixed order. */
int vgan_mse_grad(const float* target, int ldt, const float* pred, int ldp, int n, int d, float gscale,
                  double* part, float* g, int ldg, vgan_stream_t stream);
int vgan_sum_f64(const double* in, int count, double scale, float* out, int accumulate, vgan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Data-parallel exchange  (the reference has no collective): thin wrappers over RCCL for callers that do not
 * use torch.distributed.  One process per GPU.  Rank 0 calls vgan_dp_unique_id and hands the 128 bytes to every rank by
 * its own means; each rank then creates its communicator and, once per step, all-reduces (SUM, in place, float32) the
 * generator gradient on the stream the step's kernels run on -- after vgan_linear_backward_params has produced M_4 (or the
 * flat gradient), before the products that consume it.  To overlap it, issue it on a second stream and run
 * vgan_gather_rows_split + the X-X tiles of the next batch meanwhile (what v-gan_amd/trainer.py does).
 * RCCL is dlopen'ed at first use: without librccl.so these calls return VGAN_ERR_HIP and everything else still works.
 * ------------------------------------------------------------------------------------------- */
#define VGAN_DP_ID_BYTES 128
typedef struct vgan_dp_comm vgan_dp_comm;
int vgan_dp_unique_id(uint8_t* id /* [VGAN_DP_ID_BYTES] */);
int vgan_dp_comm_create(vgan_dp_comm** comm, int nranks, const uint8_t* id, int rank);
int vgan_dp_allreduce_sum(vgan_dp_comm* comm, float* buf, int64_t count, vgan_stream_t stream);
/* In-place all-gather of raw bytes: rank r's bytes_per_rank bytes already sit at buf + r * bytes_per_rank; afterwards every rank
 * holds all nranks pieces.  The exchange of the SHARDED front of a data-parallel step (chosen for large
 * batches, v-gan_amd/trainer.py): each rank runs vgan_linear_forward, vgan_mask_project_forward(row_offset = its first row),
 * vgan_mmd_bf3_prepare and vgan_colmax_partial for ITS rows only, then the ranks gather the Y rows of the split images (or of
 * Z in fp32 mode), their norms and the column-key chunks -- three or four calls, or one over a packed record -- while the
 * Gram tiles that need no other rank's rows (XY, X-X) run on another stream.
 * STATUS of the four vgan_dp_* calls: exercised on MI355X with nranks = 1 only (a one-GPU box cannot form a larger
 * communicator); the Python engine carries the same exchanges through torch.distributed (backend "nccl" = RCCL). */
int vgan_dp_allgather(vgan_dp_comm* comm, void* buf, int64_t bytes_per_rank, vgan_stream_t stream);
int vgan_dp_comm_destroy(vgan_dp_comm* comm);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline / sampling post-processing on the device  
 * vgan_shuffle_epoch: perm[i] = pi_{seed,epoch}(i) for i < count, pi a pseudo-random permutation of [0, train_size)
 * evaluated per element (balanced Feistel network + cycle walking): the shuffled drop_last batches of one epoch
 * (DataLoader(shuffle=True, drop_last=True), src/vgan.py:578-584) without a host draw, a sort or a copy.  It is NOT torch's
 * randperm stream: parity runs keep the host draw.  vgan_shuffle_index evaluates the same permutation on the host.
 * vgan_mask_unique: np.unique(masks, axis=0, return_counts=True) of approx_subspace_dist (src/vgan.py:372-382) for a
 * boolean (uint8) mask matrix [n, d]: out_row[r] = index of the first sampled row holding the r-th distinct mask in
 * numpy's lexicographic order, out_count[r] = its multiplicity; entries r >= #distinct are left untouched (pre-zero
 * out_count).  keys: workspace [n * ceil(d/64)] u64, work: [2n] i32.
 * ------------------------------------------------------------------------------------------- */
int vgan_shuffle_epoch(int32_t* perm, int64_t count, int64_t train_size, uint64_t seed, uint64_t epoch,
                       vgan_stream_t stream);
int64_t vgan_shuffle_index(int64_t i, int64_t train_size, uint64_t seed, uint64_t epoch);
int vgan_mask_unique(const uint8_t* masks, int ldm, int n, int d, uint64_t* keys, int32_t* work,
                     int32_t* out_row, int32_t* out_count, vgan_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Myopicity two-sample test  (check_if_myopic, src/vgan.py:384-431 -> torch-two-sample's MMDStatistic
 * with ret_matrix=True and its permutation p-value; that dependency is absent and unpinned, the algorithm
 * is restated in oracle/vgan_oracle.py: PARITY UNPINNED against the dependency).
 * vgan_rbf_kernel_matrix: K[i,j] = exp(-alpha |z_i - z_j|^2) for the m rows of Z ([m, p], sq = squared row
 * norms), K[i,i] = 1.   vgan_rows_dot: out[r] = sum_c A[r,c] * B[r*ldb + c] in float64 (ldb = 0 broadcasts
 * one row of B) -- with T = Ut . K (vgan_gemm_grouped) this gives u^T K u and u^T K 1 for every 0/1
 * assignment row u of Ut, from which the host forms the permutation statistics.
 * ------------------------------------------------------------------------------------------- */
int vgan_rbf_kernel_matrix(const float* Z, int ldz, int m, int p, const float* sq, float alpha, float* K, int ldk,
                           vgan_stream_t stream);
/* RBF.forward(Z) -> K [m, m]  (src/models/Mmd_loss_constrained.py:24-26) for callers of the stand-alone module:
 * K[i,j] = sum_k exp(-|z_i - z_j|^2 / (bw[0] * multipliers[k])); bw is a DEVICE scalar (the frozen bandwidth),
 * multipliers a HOST array.  dK (may be NULL) receives dK/dL = -sum_k exp(..)/(bw multipliers[k]), what the module's
 * autograd multiplies the upstream gradient with before vgan_mmd_backward. */
int vgan_rbf_multi_kernel_matrix(const float* Z, int ldz, int m, int p, const float* sq, const float* bw,
                                 const float* multipliers, int n_kernels, float* K, int ldk, float* dK, int lddk,
                                 vgan_stream_t stream);
int vgan_rows_dot(const float* A, int lda, const float* B, int ldb, double* out, int rows, int cols,
                  vgan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VGAN_HIP_H */
