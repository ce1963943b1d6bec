"""Kernel provider: the one place that turns torch tensors into C-ABI calls (include/vgan_hip.h).

``HipOps`` is the product's only provider.  Every method launches asynchronously on torch's
current HIP stream (so calls can be captured into a HIP graph) and writes into caller-owned
tensors.  A provider with the same method set over CPU tensors exists only under ``tests/`` to
exercise the host logic (fit loop, sharding, collectives) without a GPU.
"""
import ctypes

import torch

from . import lib as _lib

_f32 = torch.float32


def _round4(v):
    return (int(v) + 3) // 4 * 4


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _mat(t, name):
    if not (t.is_cuda and t.dtype == _f32 and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError(f"{name}: need a float32 HIP matrix with unit inner stride, got "
                         f"{t.dtype} {tuple(t.shape)} strides {t.stride()} on {t.device}")
    return t


def _vec(t, name, dtype=_f32):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name}: need a contiguous {dtype} HIP tensor, got {t.dtype} on {t.device}")
    return t


class HipOps:
    """libvgan_hip.so on the current device/stream.  Raises if the library or a GPU is missing."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.VganHipError("vgan_amd needs a HIP device (torch.cuda.is_available() is False); "
                                    "there is no CPU fallback")

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- host helpers -----------------------------------------------------------------------
    def build_tiles(self, n, grad_mode, rank=0, world=1, device=None, tile=64, split_xx=False, split=None):
        """split_xx (= split "xx_last"): the table is returned as [XY and YY tiles ... | XX tiles ...] (each part in XCD order)
        together with the length of the first part, for callers that launch the sum-only XX tiles separately; split "yy_last":
        [XY and XX tiles ... | YY tiles ...] (the sharded data-parallel front: the first part needs no other rank's rows)."""
        flat, cnt = _lib.build_tiles(n, grad_mode, rank, world, tile)
        table = torch.tensor(flat, dtype=torch.int32).view(cnt, 8)
        if split_xx or split:
            first, second = _lib.split_tiles(table, tile, yy_last=(split == "yy_last"))
            return torch.cat([first, second]).to(device or "cuda"), first.shape[0]
        return table.to(device or "cuda")

    def colmax_chunks(self, n):
        return self.lib.vgan_colmax_chunks(n)

    # ---- Linear ------------------------------------------------------------------------------
    def linear_forward(self, x, W, b, y, x_nslabs=1, x_slab_stride=0):
        """x_nslabs > 1: `x` is slab 0 of unreduced split-K slabs `x_slab_stride` elements apart (summed while staged)."""
        _mat(x, "x"), _mat(W, "W"), _mat(y, "y")
        n, kin = x.shape
        out = W.shape[0]
        assert W.shape[1] == kin and y.shape == (n, out)
        _lib.check(self.lib.vgan_linear_forward(_ptr(x), x.stride(0), int(x_nslabs), int(x_slab_stride), _ptr(W), W.stride(0),
                                                _ptr(b), _ptr(y), y.stride(0), n, kin, out, self._stream()), "vgan_linear_forward")

    def linear_backward_input(self, dy, W, dx):
        _mat(dy, "dy"), _mat(W, "W"), _mat(dx, "dx")
        n, out = dy.shape
        kin = W.shape[1]
        assert W.shape[0] == out and dx.shape == (n, kin)
        _lib.check(self.lib.vgan_linear_backward_input(_ptr(dy), dy.stride(0), _ptr(W), W.stride(0), _ptr(dx), dx.stride(0),
                                                       n, kin, out, self._stream()), "vgan_linear_backward_input")

    def linear_backward_params(self, dy, x, dW, db, splits=1, slab_stride=0, x_nslabs=1, x_slab_stride=0):
        """splits > 1: dW/db are slab 0 of `splits` slabs `slab_stride` elements apart (partial sums).
        x_nslabs > 1: `x` itself is slab 0 of unreduced slabs (summed while staged)."""
        _mat(dy, "dy"), _mat(x, "x"), _mat(dW, "dW")
        n, out = dy.shape
        kin = x.shape[1]
        assert x.shape[0] == n and dW.shape == (out, kin)
        _lib.check(self.lib.vgan_linear_backward_params(_ptr(dy), dy.stride(0), _ptr(x), x.stride(0), int(x_nslabs), int(x_slab_stride),
                                                        _ptr(dW), dW.stride(0), _ptr(db), n, kin, out, int(splits), int(slab_stride),
                                                        self._stream()), "vgan_linear_backward_params")

    def linear_backward_params_xx_supported(self, n, kin, out):
        return bool(self.lib.vgan_linear_backward_params_xx_supported(int(n), int(kin), int(out)))

    def linear_backward_params_xx(self, dy, x, dW, xx):
        """linear_backward_params (no bias, no slabs) with an xx_job() riding in the launch (the step's M_4 product)."""
        _mat(dy, "dy"), _mat(x, "x"), _mat(dW, "dW")
        n, out = dy.shape
        kin = x.shape[1]
        assert x.shape[0] == n and dW.shape == (out, kin)
        _lib.check(self.lib.vgan_linear_backward_params_xx(_ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(dW), dW.stride(0), n, kin, out,
                                                           ctypes.byref(xx), self._stream()), "vgan_linear_backward_params_xx")

    def reduce_slabs(self, src, slab_stride, nslabs, dst):
        _vec(dst, "dst")
        _lib.check(self.lib.vgan_reduce_slabs(_ptr(src), int(slab_stride), int(nslabs), _ptr(dst), dst.numel(), self._stream()),
                   "vgan_reduce_slabs")

    # ---- upper_softmax / projection ------------------------------------------------------------
    def col_mean(self, data, out):
        """out[d] = per-feature mean of data [rows, d]: the centre the step engine subtracts from the MMD operand."""
        _mat(data, "data"), _vec(out, "out")
        rows, d = data.shape
        assert out.numel() >= d
        _lib.check(self.lib.vgan_col_mean(_ptr(data), data.stride(0), rows, d, _ptr(out), self._stream()), "vgan_col_mean")

    @staticmethod
    def logits_chain(za, At4):
        """The collapsed generator as a job for mask_project_forward(_bf3) (`chain=`): logits = za . At4^T are formed inside that
        launch.  za [n, e0] = [z | 1 | 0-pad], At4 [d, e0].  Raw pointers: the tensors must outlive every launch using the job."""
        _mat(za, "za"), _mat(At4, "At4")
        assert za.shape[1] == At4.shape[1]
        return _lib.LogitsChain(_ptr(za), _ptr(At4), za.stride(0), At4.stride(0), za.shape[1], 0)

    @staticmethod
    def chain_fusable(n, d, *lds):
        return d % 4 == 0 and d <= 1024 and all(int(v) % 4 == 0 for v in lds)

    def mask_project_forward(self, logits, data, rows, S, U, Zx, Zy, sqx, sqy, row_cursor=None, row_batches=1, row_stride=0,
                             row_offset=0, center=None, norm_split=False, chain=None):
        """center [d]: subtracted from every row written to Zx / Zy; norm_split: sqx / sqy are the norms of the bf16 hi + lo
        split of those rows (what mmd_gram_bf3 needs); chain: a logits_chain() -- `logits` is then not read (may be None)."""
        _mat(data, "data"), _mat(Zy, "Zy"), _mat(S, "S")
        n, d = S.shape
        if logits is not None:
            _mat(logits, "logits")
            assert tuple(logits.shape) == (n, d)
        assert S.is_contiguous() and S.shape == (n, d) and (U is None or (U.is_contiguous() and U.shape == (n, d)))
        assert Zx is None or Zx.stride(0) == Zy.stride(0)
        if rows is not None:
            _vec(rows, "rows", torch.int32)
        _lib.check(self.lib.vgan_mask_project_forward(_ptr(logits), logits.stride(0) if logits is not None else 0, _ptr(data), data.stride(0), _ptr(rows),
                                                      _ptr(row_cursor), int(row_batches), int(row_stride), int(row_offset), _ptr(S), _ptr(U), _ptr(Zx), _ptr(Zy), Zy.stride(0), _ptr(sqx), _ptr(sqy),
                                                      n, d, _ptr(center), int(bool(norm_split)),
                                                      ctypes.byref(chain) if chain is not None else None, self._stream()), "vgan_mask_project_forward")

    def xx_job(self, Dh, Dl, dsq, tiles, bw, partial):
        """The X-X Gram tiles as a job for mask_project_forward_bf3 (`xx=`): Dh, Dl, dsq = split images / norms of the whole
        (centred) data set, tiles / partial = the X-X part of the tile table and of the partial buffer.  Raw pointers: the
        tensors must outlive every launch (and graph replay) that uses the job."""
        assert tiles.is_contiguous() and partial.is_contiguous() and Dh.stride(0) == Dl.stride(0)
        return _lib.XXJob(_ptr(Dh), _ptr(Dl), _ptr(dsq), _ptr(tiles), _ptr(bw), _ptr(partial), Dh.stride(0), tiles.shape[0])

    def mask_project_forward_bf3(self, logits, data, rows, S, Z, sq, Zh, Zl, ZTh, ZTl, row_cursor=None, row_batches=1, row_stride=0,
                                 center=None, write_x=True, xx=None, chain=None):
        """mask_project_forward + mmd_bf3_prepare in one launch (shape contract in include/vgan_hip.h; see bf3_fusable)."""
        _mat(data, "data"), _mat(Z, "Z"), _mat(S, "S")
        n, d = S.shape
        if logits is not None:
            _mat(logits, "logits")
        _lib.check(self.lib.vgan_mask_project_forward_bf3(_ptr(logits), logits.stride(0) if logits is not None else 0, _ptr(data), data.stride(0), _ptr(rows),
                                                          _ptr(row_cursor), int(row_batches), int(row_stride), _ptr(S), _ptr(Z), Z.stride(0),
                                                          _ptr(sq), _ptr(Zh), _ptr(Zl), Zh.stride(0), _ptr(ZTh), _ptr(ZTl),
                                                          ZTh.stride(0) if ZTh is not None else 0,
                                                          n, d, _ptr(center), int(bool(write_x)),
                                                          ctypes.byref(xx) if xx is not None else None,
                                                          ctypes.byref(chain) if chain is not None else None, self._stream()),
                   "vgan_mask_project_forward_bf3")

    @staticmethod
    def bf3_fusable(n, d, *lds):
        return d % 4 == 0 and d <= 1024 and n % 8 == 0 and all(int(v) % 4 == 0 for v in lds)

    def gather_rows(self, data, rows, out, sq, row_cursor=None, row_batches=1, row_stride=0, row_offset=0):
        _mat(data, "data"), _mat(out, "out")
        n, d = out.shape[0], data.shape[1]
        _lib.check(self.lib.vgan_gather_rows(_ptr(data), data.stride(0), _ptr(rows), _ptr(row_cursor), int(row_batches),
                                             int(row_stride), int(row_offset), _ptr(out), out.stride(0), _ptr(sq), n, d,
                                             self._stream()), "vgan_gather_rows")

    def gather_rows_split(self, data, rows, center, out, sq, norm_split=False, Zh=None, Zl=None, row_cursor=None, row_batches=1,
                          row_stride=0, row_offset=0, n=None):
        """X half of the centred MMD operand for the batch the cursor points at: out / sq / split images (each optional)."""
        _mat(data, "data")
        d = data.shape[1]
        n = int(n if n is not None else (out.shape[0] if out is not None else sq.numel()))
        _lib.check(self.lib.vgan_gather_rows_split(_ptr(data), data.stride(0), _ptr(rows), _ptr(row_cursor), int(row_batches),
                                                   int(row_stride), int(row_offset), _ptr(center), _ptr(out),
                                                   out.stride(0) if out is not None else 0, _ptr(sq), int(bool(norm_split)), _ptr(Zh),
                                                   _ptr(Zl), Zh.stride(0) if Zh is not None else 0, n, d, self._stream()),
                   "vgan_gather_rows_split")

    def upper_softmax_forward(self, logits, S, U):
        _mat(logits, "logits")
        n, d = logits.shape
        assert S.is_contiguous() and (U is None or U.is_contiguous())
        _lib.check(self.lib.vgan_upper_softmax_forward(_ptr(logits), logits.stride(0), _ptr(S), _ptr(U), n, d, self._stream()),
                   "vgan_upper_softmax_forward")

    def mask_backward(self, gU, S, colkey, pen_weight, row_offset, dlogits, nslabs=1, slab_stride=0):
        _mat(gU, "gU"), _mat(S, "S"), _mat(dlogits, "dlogits")
        n, d = S.shape
        _lib.check(self.lib.vgan_mask_backward(_ptr(gU), gU.stride(0), int(nslabs), int(slab_stride), _ptr(S), S.stride(0),
                                               _ptr(colkey), float(pen_weight),
                                               int(row_offset), _ptr(dlogits), dlogits.stride(0), n, d, self._stream()),
                   "vgan_mask_backward")

    def colmax(self, S, row_offset, part, colkey, from_softmax=True):
        _mat(S, "S")
        n, d = S.shape
        assert part.dtype == torch.int64 and colkey.dtype == torch.int64 and part.numel() >= self.colmax_chunks(n) * d
        _lib.check(self.lib.vgan_colmax(_ptr(S), S.stride(0), int(bool(from_softmax)), int(row_offset), _ptr(part), _ptr(colkey), n, d, self._stream()),
                   "vgan_colmax")

    def colmax_partial(self, S, row_offset, part, from_softmax=True):
        _mat(S, "S")
        n, d = S.shape
        assert part.dtype == torch.int64 and part.numel() >= self.colmax_chunks(n) * d
        _lib.check(self.lib.vgan_colmax_partial(_ptr(S), S.stride(0), int(bool(from_softmax)), int(row_offset), _ptr(part), n, d,
                                                self._stream()), "vgan_colmax_partial")

    def mmd_finalize(self, partial, tiles, colpart, chunks, colkey, n, d, weight, stats, loss, loss_accum=None, accum_scale=1.0,
                     step_counter=None):
        _lib.check(self.lib.vgan_mmd_finalize(_ptr(partial), _ptr(tiles), tiles.shape[0], _ptr(colpart), int(chunks), _ptr(colkey),
                                              int(n), int(d), float(weight), _ptr(stats), _ptr(loss), _ptr(loss_accum),
                                              float(accum_scale), _ptr(step_counter), self._stream()), "vgan_mmd_finalize")

    def finalize_job(self, partial, tiles, colpart, chunks, colkey, n, d, weight, stats, loss, loss_accum=None, accum_scale=1.0,
                     step_counter=None, mode=0, ntiles_main=0):
        """The arguments of mmd_finalize as a job for mmd_backward / mmd_backward_bf3 (`finalize=`) or gemm_grouped (`fold=`).
        mode 1 / 2: the two halves of a split tail (include/vgan_hip.h: the X-X block sum arrives later in the step).  The job
        holds raw device pointers: the tensors must outlive every launch (and graph replay) that uses it."""
        return _lib.FinalizeJob(_ptr(partial), _ptr(tiles), _ptr(colpart), _ptr(colkey), _ptr(stats), _ptr(loss), _ptr(loss_accum),
                                _ptr(step_counter), tiles.shape[0], int(chunks), int(n), int(d), float(weight), float(accum_scale),
                                int(mode), int(ntiles_main))

    def mask_from_softmax(self, S, U):
        _mat(S, "S"), _mat(U, "U")
        n, d = S.shape
        _lib.check(self.lib.vgan_mask_from_softmax(_ptr(S), S.stride(0), _ptr(U), U.stride(0), n, d, self._stream()),
                   "vgan_mask_from_softmax")

    # ---- MMD -------------------------------------------------------------------------------------
    def row_sqnorm(self, Z, sq, p):
        _mat(Z, "Z")
        _lib.check(self.lib.vgan_row_sqnorm(_ptr(Z), Z.stride(0), _ptr(sq), Z.shape[0], int(p), self._stream()), "vgan_row_sqnorm")

    def mmd_gram(self, Z, sq, n, p, bw, tiles, calibrate, Wg, wrow0, partial):
        _mat(Z, "Z")
        assert (calibrate or Z.shape[0] >= 2 * n) and tiles.dtype == torch.int32 and tiles.is_contiguous()  # (calibration: the table bounds the rows)
        ntiles = tiles.shape[0]
        assert partial.numel() >= 4 * ntiles
        ldw = Wg.stride(0) if Wg is not None else 0
        _lib.check(self.lib.vgan_mmd_gram(_ptr(Z), Z.stride(0), _ptr(sq), int(n), int(p), _ptr(bw), _ptr(tiles), ntiles,
                                          int(bool(calibrate)), _ptr(Wg), ldw, int(wrow0), _ptr(partial), self._stream()),
                   "vgan_mmd_gram")

    @staticmethod
    def _mults(multipliers):
        vals = [float(v) for v in multipliers]
        return (ctypes.c_float * len(vals))(*vals), len(vals)

    def mmd_gram_general(self, Z, sq, n, p, bw, tiles, multipliers, Wg, wrow0, partial):
        """mmd_gram (no calibration) for RBF(n_kernels, mul_factor) other than the reference's defaults: `multipliers` is the
        host list mul_factor ** (k - n_kernels // 2)."""
        _mat(Z, "Z")
        ntiles = tiles.shape[0]
        assert partial.numel() >= 4 * ntiles
        arr, nk = self._mults(multipliers)
        ldw = Wg.stride(0) if Wg is not None else 0
        _lib.check(self.lib.vgan_mmd_gram_general(_ptr(Z), Z.stride(0), _ptr(sq), int(n), int(p), _ptr(bw), _ptr(tiles), ntiles, arr, nk,
                                                  _ptr(Wg), ldw, int(wrow0), _ptr(partial), self._stream()), "vgan_mmd_gram_general")

    def rbf_multi_kernel_matrix(self, Z, sq, bw, multipliers, K, dK=None):
        """RBF.forward: K [m, m] = sum_k exp(-|z_i - z_j|^2 / (bw * multipliers[k])); dK (optional) = dK/dL."""
        _mat(Z, "Z"), _mat(K, "K")
        m, p = Z.shape
        arr, nk = self._mults(multipliers)
        _lib.check(self.lib.vgan_rbf_multi_kernel_matrix(_ptr(Z), Z.stride(0), m, p, _ptr(sq), _ptr(bw), arr, nk, _ptr(K), K.stride(0),
                                                         _ptr(dK), dK.stride(0) if dK is not None else 0, self._stream()),
                   "vgan_rbf_multi_kernel_matrix")

    def mmd_gram_colmax(self, Z, sq, n, p, bw, tiles, Wg, wrow0, partial, S, row_offset, colpart, from_softmax=True):
        """mmd_gram (no calibration) + colmax_partial(S) in one launch."""
        _mat(Z, "Z"), _mat(S, "S")
        ntiles = tiles.shape[0]
        nrows, d = S.shape
        assert partial.numel() >= 4 * ntiles and colpart.dtype == torch.int64 and colpart.numel() >= self.colmax_chunks(nrows) * d
        ldw = Wg.stride(0) if Wg is not None else 0
        _lib.check(self.lib.vgan_mmd_gram_colmax(_ptr(Z), Z.stride(0), _ptr(sq), int(n), int(p), _ptr(bw), _ptr(tiles), ntiles,
                                                 _ptr(Wg), ldw, int(wrow0), _ptr(partial), _ptr(S), S.stride(0),
                                                 int(bool(from_softmax)), int(row_offset), _ptr(colpart), nrows, d, self._stream()),
                   "vgan_mmd_gram_colmax")

    def mmd_reduce(self, partial, tiles, stats, zero_first=True):
        assert stats.dtype == torch.float64 and stats.numel() >= 4
        _lib.check(self.lib.vgan_mmd_reduce(_ptr(partial), _ptr(tiles), tiles.shape[0], _ptr(stats), int(bool(zero_first)),
                                            self._stream()), "vgan_mmd_reduce")

    def mmd_set_bandwidth(self, stats, n, bw):
        _lib.check(self.lib.vgan_mmd_set_bandwidth(_ptr(stats), int(n), _ptr(bw), self._stream()), "vgan_mmd_set_bandwidth")

    def mmd_loss(self, stats, colkey, n, d, weight, loss, loss_accum=None, accum_scale=1.0, step_counter=None):
        _lib.check(self.lib.vgan_mmd_loss(_ptr(stats), _ptr(colkey), int(n), int(d), float(weight), _ptr(loss), _ptr(loss_accum),
                                          float(accum_scale), _ptr(step_counter), self._stream()), "vgan_mmd_loss")

    def mmd_backward(self, Wg, Z, wrow0, nr, ncols, p, mul, out, splits=1, slab_stride=0, finalize=None, mul_shift=None):
        """splits > 1: `out` is slab 0 of `splits` partial slabs `slab_stride` elements apart.  finalize: a finalize_job()
        that one extra workgroup of the launch executes.  mul_shift [p]: added to `mul` (which is stored centred)."""
        _mat(Wg, "Wg"), _mat(Z, "Z"), _mat(out, "out")
        ldmul = mul.stride(0) if mul is not None else 0
        _lib.check(self.lib.vgan_mmd_backward(_ptr(Wg), Wg.stride(0), _ptr(Z), Z.stride(0), int(wrow0), int(nr), int(ncols), int(p),
                                              _ptr(mul), ldmul, _ptr(mul_shift), _ptr(out), out.stride(0), int(splits), int(slab_stride),
                                              ctypes.byref(finalize) if finalize is not None else None, self._stream()),
                   "vgan_mmd_backward")

    # ---- split-bf16 MMD (opt-in precision mode) ------------------------------------------------------
    def mmd_bf3_prepare(self, Z, rows, p, Zh, Zl, ZTh=None, ZTl=None):
        _mat(Z, "Z")
        for t in (Zh, Zl, ZTh, ZTl):
            assert t is None or (t.dtype == torch.int16 and t.is_cuda and t.stride(1) == 1)
        kn = ZTh.stride(0) if ZTh is not None else 0
        _lib.check(self.lib.vgan_mmd_bf3_prepare(_ptr(Z), Z.stride(0), int(rows), int(p), _ptr(Zh), _ptr(Zl), Zh.stride(0),
                                                 _ptr(ZTh), _ptr(ZTl), kn, self._stream()), "vgan_mmd_bf3_prepare")

    def gram_tail_workspace(self, device):
        """Workspace a tile-256 mmd_gram_bf3 launch splits its last round in (include/vgan_hip.h: tail_ws); zeroed once here."""
        return torch.zeros(int(self.lib.vgan_mmd_gram_bf3_tail_ws_bytes()) // 4, dtype=torch.int32, device=device)

    def mmd_gram_bf3(self, Zh, Zl, sq, n, bw, tiles, Wh, Wl, wrow0, partial, S=None, row_offset=0, colpart=None, from_softmax=True,
                     tile=64, tail_ws=None, rs_part=None):
        ntiles = tiles.shape[0]
        nrows, d = (S.shape if S is not None else (0, 0))
        ldw = Wh.stride(0) if Wh is not None else 0
        _lib.check(self.lib.vgan_mmd_gram_bf3(_ptr(Zh), _ptr(Zl), Zh.stride(0), _ptr(sq), int(n), _ptr(bw), _ptr(tiles), ntiles,
                                              int(tile), _ptr(Wh), _ptr(Wl), ldw, int(wrow0), _ptr(partial), _ptr(S),
                                              S.stride(0) if S is not None else 0, int(bool(from_softmax)), int(row_offset),
                                              _ptr(colpart), nrows, d, _ptr(tail_ws), tail_ws.numel() * 4 if tail_ws is not None else 0,
                                              _ptr(rs_part), rs_part.stride(0) if rs_part is not None else 0, self._stream()),
                   "vgan_mmd_gram_bf3")

    def mmd_backward_bf3(self, Wh, Wl, ZTh, ZTl, Z, wrow0, nr, p, mul, out, splits=1, slab_stride=0, finalize=None, mul_shift=None,
                         tile=0):
        """tile: 0 = the library's choice between its 64- and 128-wide tiles, or 64 / 128 to force one."""
        _mat(Z, "Z"), _mat(out, "out")
        ldmul = mul.stride(0) if mul is not None else 0
        _lib.check(self.lib.vgan_mmd_backward_bf3(_ptr(Wh), _ptr(Wl), Wh.stride(0), _ptr(ZTh), _ptr(ZTl), ZTh.stride(0),
                                                  ZTh.shape[0], _ptr(Z), Z.stride(0), int(wrow0), int(nr), int(p), _ptr(mul), ldmul,
                                                  _ptr(mul_shift), _ptr(out), out.stride(0), int(splits), int(slab_stride), int(tile),
                                                  ctypes.byref(finalize) if finalize is not None else None, self._stream()),
                   "vgan_mmd_backward_bf3")

    def mmd_backward_bf3_rm(self, Wh, Wl, Zh, Zl, zrows, Z, wrow0, nr, p, mul, out, splits=1, slab_stride=0, finalize=None, mul_shift=None,
                            tile=0, xx=None, rs_part=None):
        """mmd_backward_bf3 on the ROW-MAJOR split images Zh, Zl [>= zrows, kp] (no transposed copies of Z).  xx: an xx_job() whose
        X-X Gram tiles ride in the launch as surplus workgroups (64-wide tiles only).  rs_part: the per-slot row sums of W a
        tile-256 mmd_gram_bf3 launch left (include/vgan_hip.h)."""
        _mat(Z, "Z"), _mat(out, "out")
        ldmul = mul.stride(0) if mul is not None else 0
        kn = (int(zrows) + 63) // 64 * 64
        assert Wh.stride(0) >= kn and Zh.shape[0] >= zrows
        if xx is not None:
            assert tile in (0, 64)
            _lib.check(self.lib.vgan_mmd_backward_bf3_rm_xx(_ptr(Wh), _ptr(Wl), Wh.stride(0), kn, _ptr(Zh), _ptr(Zl), Zh.stride(0), int(zrows),
                                                            _ptr(Z), Z.stride(0), int(wrow0), int(nr), int(p), _ptr(mul), ldmul, _ptr(mul_shift),
                                                            _ptr(out), out.stride(0), int(splits), int(slab_stride),
                                                            ctypes.byref(finalize) if finalize is not None else None, ctypes.byref(xx),
                                                            self._stream()), "vgan_mmd_backward_bf3_rm_xx")
            return
        _lib.check(self.lib.vgan_mmd_backward_bf3_rm(_ptr(Wh), _ptr(Wl), Wh.stride(0), kn, _ptr(Zh), _ptr(Zl), Zh.stride(0), int(zrows),
                                                     _ptr(Z), Z.stride(0), int(wrow0), int(nr), int(p), _ptr(mul), ldmul, _ptr(mul_shift),
                                                     _ptr(out), out.stride(0), int(splits), int(slab_stride), int(tile),
                                                     ctypes.byref(finalize) if finalize is not None else None, _ptr(rs_part),
                                                     rs_part.stride(0) if rs_part is not None else 0, self._stream()),
                   "vgan_mmd_backward_bf3_rm")

    def mmd_backward_bf3_tile(self, nr, p, splits=1, tile=0):
        """Tile edge (64 / 128) mmd_backward_bf3 runs for this shape (host-side query of the library's rule)."""
        return int(self.lib.vgan_mmd_backward_bf3_tile(int(nr), int(p), int(splits), int(tile)))

    def gemm_grouped(self, problems, copy=None, adadelta=None, noise=None, fold=None):
        """problems: up to 4 tuples (kind, A, B, C) with kind in "NN" (C = A.B), "NT" (C = A.B^T), "TN" (C = A^T.B); 2-D float32
        tensors with unit inner stride.  One launch; the products must not depend on each other.  A fifth tuple element
        `splits` > 1 cuts the contraction into that many slices run by different workgroups: C is then a contiguous
        [splits, m, n] tensor of partial products for the caller to sum (reduce_slabs).  Jobs that may ride in the launch
        (vgan_gemm_grouped_ex):
          copy = (src, dst)             contiguous float32 tensors of equal size, dst <- src;
          adadelta = dict(p, sq, acc, lr, rho, eps, weight_decay, grad_scale, layers=[(w_packed, off_w, off_b, out, in) per
                     problem (+ one more with extra_grad)], extra_grad=None): the optimiser update in the products' epilogue;
          noise = dict(next_noise, noise_cols, noise_ones_col, seed, step_counter): the next step's noise draw;
          fold = a finalize_job() run by one surplus workgroup (the late half of a split step tail)."""
        assert 1 <= len(problems) <= _lib.GEMM_MAX_GROUP
        arr = (_lib.GemmProblem * len(problems))()
        for q, (kind, A, B, C, *rest) in zip(arr, problems):
            if kind == "NT2":  # C = (A . B^T) . D^T in one tile pass: (kind, A, B, C, D, scratch)
                D, scratch = rest
                _mat(A, "A"), _mat(B, "B"), _mat(C, "C"), _mat(D, "D")
                (m, k), (k2, kb), (n, k2d) = A.shape, B.shape, D.shape
                assert k == kb and k2 == k2d and tuple(C.shape) == (m, n)
                tiles = ((m + 63) // 64) * ((n + 63) // 64)
                assert scratch.is_cuda and scratch.dtype == _f32 and scratch.is_contiguous() and scratch.numel() >= tiles * 64 * _round4(k2)
                q.a, q.b, q.c, q.d, q.scratch = A.data_ptr(), B.data_ptr(), C.data_ptr(), D.data_ptr(), scratch.data_ptr()
                q.kind, q.m, q.n, q.k, q.k2, q.splitk = _lib.GEMM_NT_NT, m, n, k, k2, 1
                q.lda, q.ldb, q.ldc, q.ldd = A.stride(0), B.stride(0), C.stride(0), D.stride(0)
                continue
            q.splitk = int(rest[0]) if rest else 1
            if q.splitk > 1:
                assert C.dim() == 3 and C.shape[0] == q.splitk and C.is_contiguous()
                C = C[0]
            _mat(A, "A"), _mat(B, "B"), _mat(C, "C")
            if kind == "NN":
                (m, k), (k2, n), code = A.shape, B.shape, _lib.GEMM_NN
            elif kind == "NT":
                (m, k), (n, k2), code = A.shape, B.shape, _lib.GEMM_NT
            elif kind == "TN":
                (k, m), (k2, n), code = A.shape, B.shape, _lib.GEMM_TN
            else:
                raise ValueError(kind)
            assert k == k2 and tuple(C.shape) == (m, n), (kind, tuple(A.shape), tuple(B.shape), tuple(C.shape))
            q.a, q.b, q.c, q.kind, q.m, q.n, q.k = A.data_ptr(), B.data_ptr(), C.data_ptr(), code, m, n, k
            q.lda, q.ldb, q.ldc = A.stride(0), B.stride(0), C.stride(0)
        if copy is None and adadelta is None and noise is None and fold is None:
            _lib.check(self.lib.vgan_gemm_grouped(arr, len(problems), self._stream()), "vgan_gemm_grouped")
            return
        x = _lib.GroupedExtras()
        if copy is not None:
            src, dst = copy
            _vec(src, "copy src"), _vec(dst, "copy dst")
            assert src.numel() == dst.numel()
            x.copy_src, x.copy_dst, x.copy_count = src.data_ptr(), dst.data_ptr(), src.numel()
        if adadelta is not None:
            a = adadelta
            for nm in ("p", "sq", "acc"):
                _vec(a[nm], nm)
            x.adadelta, x.p, x.sq_avg, x.acc_delta = 1, a["p"].data_ptr(), a["sq"].data_ptr(), a["acc"].data_ptr()
            x.lr, x.rho, x.eps, x.weight_decay, x.grad_scale = (float(a["lr"]), float(a.get("rho", 0.9)), float(a.get("eps", 1e-6)),
                                                                float(a.get("weight_decay", 0.0)), float(a.get("grad_scale", 1.0)))
            extra = a.get("extra_grad")
            assert len(a["layers"]) == len(problems) + (1 if extra is not None else 0)
            for L, (w, off_w, off_b, out, inp) in zip(x.layer, a["layers"]):
                _mat(w, "w_packed")
                L.w_packed, L.off_w, L.off_b, L.ldp, L.out, L.inp = w.data_ptr(), int(off_w), int(off_b), w.stride(0), int(out), int(inp)
            if extra is not None:
                _mat(extra, "extra_grad")
                x.g_extra, x.ld_extra = extra.data_ptr(), extra.stride(0)
        if noise is not None:
            z = noise["next_noise"]
            x.next_noise, x.noise_rows, x.noise_ld = z.data_ptr(), z.shape[0], z.stride(0)
            x.noise_cols, x.noise_ones_col = int(noise["noise_cols"]), int(noise["noise_ones_col"])
            x.seed, x.step_counter = int(noise["seed"]) & 0xFFFFFFFFFFFFFFFF, noise["step_counter"].data_ptr()
        if fold is not None:
            x.fold = ctypes.addressof(fold)
        _lib.check(self.lib.vgan_gemm_grouped_ex(arr, len(problems), ctypes.byref(x), self._stream()), "vgan_gemm_grouped_ex")

    def mse_grad(self, target, pred, gscale, part, g):
        """part[ceil(n/4)] (float64) = partial sums of (pred - target)^2; g = gscale * (pred - target)."""
        _mat(target, "target"), _mat(pred, "pred"), _mat(g, "g")
        n, d = pred.shape
        assert part.dtype == torch.float64 and part.numel() >= (n + 3) // 4
        _lib.check(self.lib.vgan_mse_grad(_ptr(target), target.stride(0), _ptr(pred), pred.stride(0), n, d, float(gscale), _ptr(part),
                                          _ptr(g), g.stride(0), self._stream()), "vgan_mse_grad")

    def sum_f64(self, src, count, scale, out, accumulate=False):
        assert src.dtype == torch.float64
        _lib.check(self.lib.vgan_sum_f64(_ptr(src), int(count), float(scale), _ptr(out), int(bool(accumulate)), self._stream()),
                   "vgan_sum_f64")

    # ---- myopicity two-sample test -----------------------------------------------------------------
    def rbf_kernel_matrix(self, Z, sq, alpha, K):
        _mat(Z, "Z"), _mat(K, "K")
        m, p = Z.shape
        _lib.check(self.lib.vgan_rbf_kernel_matrix(_ptr(Z), Z.stride(0), m, p, _ptr(sq), float(alpha), _ptr(K), K.stride(0),
                                                   self._stream()), "vgan_rbf_kernel_matrix")

    def rows_dot(self, A, B, out, broadcast_b=False):
        """out[r] = <A[r], B[r]> (float64); broadcast_b: B is one row used for every r."""
        _mat(A, "A")
        rows, cols = A.shape
        assert out.dtype == torch.float64 and out.numel() >= rows
        ldb = 0 if broadcast_b else B.stride(0)
        _lib.check(self.lib.vgan_rows_dot(_ptr(A), A.stride(0), _ptr(B), ldb, _ptr(out), rows, cols, self._stream()), "vgan_rows_dot")

    # ---- data-parallel exchange through the C ABI (RCCL; the step engine itself uses torch.distributed) ------------
    def dp_unique_id(self):
        buf = (ctypes.c_uint8 * 128)()
        _lib.check(self.lib.vgan_dp_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "vgan_dp_unique_id")
        return bytes(buf)

    def dp_comm_create(self, nranks, unique_id, rank):
        comm = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self.lib.vgan_dp_comm_create(ctypes.byref(comm), int(nranks), ctypes.cast(buf, ctypes.c_void_p), int(rank)),
                   "vgan_dp_comm_create")
        return comm

    def dp_allreduce_sum(self, comm, t):
        _vec(t, "t")
        _lib.check(self.lib.vgan_dp_allreduce_sum(comm, _ptr(t), t.numel(), self._stream()), "vgan_dp_allreduce_sum")

    def dp_allgather(self, comm, t, nranks):
        """In-place all-gather over dim 0 of a contiguous tensor: rank r's slice is t[r * len(t) // nranks ...]."""
        assert t.is_cuda and t.is_contiguous() and t.shape[0] % nranks == 0
        _lib.check(self.lib.vgan_dp_allgather(comm, _ptr(t), t.numel() * t.element_size() // nranks, self._stream()), "vgan_dp_allgather")

    def dp_comm_destroy(self, comm):
        _lib.check(self.lib.vgan_dp_comm_destroy(comm), "vgan_dp_comm_destroy")

    # ---- input pipeline / sampling post-processing on the device ------------------------------------
    def shuffle_epoch(self, perm, train_size, seed, epoch):
        """perm (int32, any shape, contiguous): the first perm.numel() entries of a pseudo-random permutation of
        [0, train_size) keyed by (seed, epoch) -- one epoch of shuffled drop_last batches, produced on the device."""
        _vec(perm, "perm", torch.int32)
        _lib.check(self.lib.vgan_shuffle_epoch(_ptr(perm), perm.numel(), int(train_size), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                               int(epoch) & 0xFFFFFFFFFFFFFFFF, self._stream()), "vgan_shuffle_epoch")

    def shuffle_index(self, i, train_size, seed, epoch):
        return int(self.lib.vgan_shuffle_index(int(i), int(train_size), int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch) & 0xFFFFFFFFFFFFFFFF))

    def mask_unique(self, masks):
        """masks: bool [n, d] on the device -> (unique rows [m, d] bool in numpy's np.unique(axis=0) order, counts [m] int64)."""
        assert masks.is_cuda and masks.dtype == torch.bool and masks.dim() == 2
        m8 = masks.contiguous().view(torch.uint8)
        n, d = m8.shape
        dev = masks.device
        keys = torch.empty(n * ((d + 63) // 64), dtype=torch.int64, device=dev)
        work = torch.empty(2 * n, dtype=torch.int32, device=dev)
        out_row = torch.zeros(n, dtype=torch.int32, device=dev)
        out_count = torch.zeros(n, dtype=torch.int32, device=dev)
        _lib.check(self.lib.vgan_mask_unique(_ptr(m8), m8.stride(0), n, d, _ptr(keys), _ptr(work), _ptr(out_row), _ptr(out_count),
                                             self._stream()), "vgan_mask_unique")
        cnt = out_count.cpu()
        m = int((cnt > 0).sum())
        return masks[out_row[:m].long()], cnt[:m].to(torch.int64)

    # ---- optimiser / noise / misc ----------------------------------------------------------------
    def adadelta_step(self, p, g, sq, acc, lr, rho=0.9, eps=1e-6, weight_decay=0.0, grad_scale=1.0, nslabs=1, slab_stride=0):
        """nslabs > 1: `g` is slab 0 of split-K gradient slabs `slab_stride` apart, summed inside the kernel."""
        for t, nm in ((p, "p"), (sq, "sq"), (acc, "acc")):
            _vec(t, nm)
        _lib.check(self.lib.vgan_adadelta_step(_ptr(p), _ptr(g), int(nslabs), int(slab_stride), _ptr(sq), _ptr(acc), p.numel(),
                                               float(lr), float(rho), float(eps),
                                               float(weight_decay), float(grad_scale), self._stream()), "vgan_adadelta_step")

    def adadelta_step_packed(self, p, pmap, g_packed, w_packed, sq, acc, lr, rho=0.9, eps=1e-6, weight_decay=0.0, grad_scale=1.0,
                             next_noise=None, noise_cols=0, noise_ones_col=-1, seed=0, step_counter=None):
        """next_noise [rows, ld]: also draw the next step's noise (first `noise_cols` columns, optional ones column)."""
        for t, nm in ((p, "p"), (sq, "sq"), (acc, "acc"), (g_packed, "g_packed"), (w_packed, "w_packed")):
            _vec(t, nm)
        _vec(pmap, "pmap", torch.int32)
        assert pmap.numel() == p.numel()
        zr, zld = (next_noise.shape[0], next_noise.stride(0)) if next_noise is not None else (0, 0)
        _lib.check(self.lib.vgan_adadelta_step_packed(_ptr(p), _ptr(pmap), _ptr(g_packed), _ptr(w_packed), _ptr(sq), _ptr(acc),
                                                      p.numel(), float(lr), float(rho), float(eps), float(weight_decay),
                                                      float(grad_scale), _ptr(next_noise), zr, int(noise_cols), zld,
                                                      int(noise_ones_col), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(step_counter),
                                                      self._stream()), "vgan_adadelta_step_packed")

    def noise_normal(self, z, seed, step_counter, stream_id=0, cols=None, ones_col=-1):
        """z [rows, ld]: standard normals in the first `cols` columns (default all); optional column of ones."""
        _mat(z, "z")
        rows, ld = z.shape[0], z.stride(0)
        cols = z.shape[1] if cols is None else int(cols)
        _lib.check(self.lib.vgan_noise_normal(_ptr(z), rows, cols, ld, int(ones_col), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                              _ptr(step_counter), int(stream_id), self._stream()), "vgan_noise_normal")

    def homogeneous_pack(self, layers, unpack=False):
        """layers: [(W [out,in], b [out], P [>=out+1, >=in+1]) ...].  pack: P = [[W, b],[0, 1]]; unpack: (W, b) <- P.
        One launch for all layers; the device-side pointer table is built once per distinct layer set."""
        key = tuple((W.data_ptr(), tuple(W.shape), W.stride(0), b.data_ptr(), P.data_ptr(), tuple(P.shape), P.stride(0)) for W, b, P in layers)
        cache = self.__dict__.setdefault("_pack_tables", {})
        if key not in cache:
            if len(cache) >= 16:  # a handful of live engines at most: drop the oldest table instead of growing without bound
                cache.pop(next(iter(cache)))
            rows = []
            for W, b, P in layers:
                _mat(W, "W"), _vec(b, "b"), _mat(P, "P")
                out, kin = W.shape
                assert b.numel() == out and P.shape[0] >= out + 1 and P.shape[1] >= kin + 1
                rows.append([W.data_ptr(), b.data_ptr(), P.data_ptr(), out, kin, W.stride(0), P.stride(0), 0])
            cache[key] = (torch.tensor(rows, dtype=torch.int64, device=layers[0][0].device),
                          max((r[3] + 1) * (r[4] + 1) for r in rows))
        desc, max_elems = cache[key]
        _lib.check(self.lib.vgan_homogeneous_pack(_ptr(desc), desc.shape[0], int(max_elems), int(bool(unpack)), self._stream()),
                   "vgan_homogeneous_pack")

    def mse(self, a, b, scale, out, accumulate=False):
        _mat(a, "a"), _mat(b, "b")
        n, d = a.shape
        _lib.check(self.lib.vgan_mse(_ptr(a), a.stride(0), _ptr(b), b.stride(0), n, d, float(scale), _ptr(out),
                                     int(bool(accumulate)), self._stream()), "vgan_mse")


_default = None


def default_ops():
    global _default
    if _default is None:
        _default = HipOps()
    return _default
