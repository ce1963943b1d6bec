"""TEST INFRASTRUCTURE ONLY: a kernel provider with HipOps' method set over CPU tensors, built on the
numpy oracle.  It lets the host logic of the product (step engine, fit loop, row-sharded data parallel
exchange over gloo) run in the CPU-only test tier.  The product never imports this file.
"""
import numpy as np
import torch

from oracle import vgan_oracle as orc
from vgan_amd import lib as _lib

TF_SLOT, TF_TWICE, TF_STORE, TF_MIRROR, TF_NEG = 3, 4, 8, 16, 32


def _np(t):
    return t.numpy() if t is not None else None


def pack_key(u, row):
    return (np.asarray(u, dtype=np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.uint64(0xFFFFFFFF - row)


class CpuOps:
    name = "cpu-oracle (tests only)"

    def __init__(self):
        self.lib = _lib.load()  # host-side helpers of the .so (tile tables) work without a GPU

    def build_tiles(self, n, grad_mode, rank=0, world=1, device=None, tile=64, split_xx=False, split=None):
        flat, cnt = _lib.build_tiles(n, grad_mode, rank, world, tile)
        table = torch.tensor(flat, dtype=torch.int32).view(cnt, 8)
        if split_xx or split:
            first, second = _lib.split_tiles(table, tile, yy_last=(split == "yy_last"))
            return torch.cat([first, second]), first.shape[0]
        return table

    def colmax_chunks(self, n):
        return self.lib.vgan_colmax_chunks(n)

    # ---- Linear
    @staticmethod
    def _sum_slabs(x, nslabs, stride):
        if nslabs == 1:
            return x
        return sum(torch.as_strided(x, x.shape, x.stride(), x.storage_offset() + q * stride) for q in range(nslabs))

    def linear_forward(self, x, W, b, y, x_nslabs=1, x_slab_stride=0):
        x = self._sum_slabs(x, x_nslabs, x_slab_stride)
        y.copy_(torch.as_tensor(_np(x).astype(np.float64) @ _np(W).astype(np.float64).T + (_np(b) if b is not None else 0)))

    def linear_backward_input(self, dy, W, dx):
        dx.copy_(torch.as_tensor(_np(dy).astype(np.float64) @ _np(W).astype(np.float64)))

    def linear_backward_params(self, dy, x, dW, db, splits=1, slab_stride=0, x_nslabs=1, x_slab_stride=0):
        x = self._sum_slabs(x, x_nslabs, x_slab_stride)
        n = dy.shape[0]
        if splits == 1:
            dW.copy_(torch.as_tensor(_np(dy).astype(np.float64).T @ _np(x).astype(np.float64)))
            if db is not None:
                db.copy_(torch.as_tensor(_np(dy).astype(np.float64).sum(0)))
            return
        # slabs: dW/db are views into slab 0 of a [splits, slab_stride] buffer
        kchunk = ((n + splits - 1) // splits + 3) // 4 * 4
        for s in range(splits):
            lo, hi = min(s * kchunk, n), min((s + 1) * kchunk, n)
            part_w = _np(dy)[lo:hi].astype(np.float64).T @ _np(x)[lo:hi].astype(np.float64)
            wv = torch.as_strided(dW, dW.shape, dW.stride(), dW.storage_offset() + s * slab_stride)
            wv.copy_(torch.as_tensor(part_w))
            if db is not None:
                bv = torch.as_strided(db, db.shape, db.stride(), db.storage_offset() + s * slab_stride)
                bv.copy_(torch.as_tensor(_np(dy)[lo:hi].astype(np.float64).sum(0)))

    def reduce_slabs(self, src, slab_stride, nslabs, dst):
        flat = src.reshape(-1)
        acc = torch.zeros_like(dst)
        for s in range(nslabs):
            acc += flat[s * slab_stride:s * slab_stride + dst.numel()]
        dst.copy_(acc)

    # ---- mask / projection
    @staticmethod
    def _rows(rows, row_cursor, row_batches, row_stride, row_offset, n):
        if rows is None:
            return np.arange(row_offset, row_offset + n)
        b = int(row_cursor.item()) % row_batches if row_cursor is not None else 0
        flat = rows.reshape(-1).numpy()
        return flat[b * row_stride + row_offset: b * row_stride + row_offset + n].astype(np.int64)

    def col_mean(self, data, out):
        out[:data.shape[1]].copy_(torch.as_tensor(_np(data).astype(np.float64).mean(0)))

    @staticmethod
    def logits_chain(za, At4):
        return (za, At4)

    @staticmethod
    def chain_fusable(n, d, *lds):
        return d % 4 == 0 and d <= 1024 and all(int(v) % 4 == 0 for v in lds)

    def mask_project_forward(self, logits, data, rows, S, U, Zx, Zy, sqx, sqy, row_cursor=None, row_batches=1, row_stride=0,
                             row_offset=0, center=None, norm_split=False, chain=None):
        n, d = S.shape
        if chain is not None:  # logits = [z|1] . At_4^T formed inside the launch
            logits = torch.as_tensor((_np(chain[0]).astype(np.float64) @ _np(chain[1])[:d].astype(np.float64).T).astype(np.float32))
        u, s = orc.upper_softmax_forward(_np(logits).astype(np.float32))
        S.copy_(torch.as_tensor(s))
        if U is not None:
            U.copy_(torch.as_tensor(u))
        X = _np(data)[self._rows(rows, row_cursor, row_batches, row_stride, row_offset, n)]
        Y = u * X
        if center is not None:  # the MMD operand is centred: Z = [X - c ; U*X - c]
            X, Y = X - _np(center)[:d], Y - _np(center)[:d]
        if Zx is not None:
            Zx[:, :d].copy_(torch.as_tensor(X))
        Zy[:, :d].copy_(torch.as_tensor(Y))
        if norm_split:  # norms of the bf16 hi + lo values
            def rounded(a):
                hi, lo = self._split(a)
                return (hi.float() + lo.float()).numpy()
            X, Y = rounded(X), rounded(Y)
        if sqx is not None:
            sqx.copy_(torch.as_tensor((X.astype(np.float64) ** 2).sum(1)))
        sqy.copy_(torch.as_tensor((Y.astype(np.float64) ** 2).sum(1)))

    def xx_job(self, Dh, Dl, dsq, tiles, bw, partial):
        return dict(Dh=Dh, Dl=Dl, dsq=dsq, tiles=tiles, bw=bw, partial=partial)

    def mask_project_forward_bf3(self, logits, data, rows, S, Z, sq, Zh, Zl, ZTh, ZTl, row_cursor=None, row_batches=1, row_stride=0,
                                 center=None, write_x=True, xx=None, chain=None):
        n, d = S.shape
        if xx is not None:  # the X-X tiles of this batch from the data set's split images, gathered by the batch indices
            idx = torch.as_tensor(self._rows(rows, row_cursor, row_batches, row_stride, 0, n))
            Xh = torch.zeros(2 * n, xx["Dh"].shape[1], dtype=torch.int16)
            Xl = torch.zeros_like(Xh)
            Xh[:n], Xl[:n] = xx["Dh"][idx], xx["Dl"][idx]
            sqx = torch.zeros(2 * n)
            sqx[:n] = xx["dsq"][idx]
            self.mmd_gram_bf3(Xh, Xl, sqx, n, xx["bw"], xx["tiles"], None, None, 0, xx["partial"])
        if write_x:
            self.mask_project_forward(logits, data, rows, S, None, Z[:n], Z[n:], sq[:n], sq[n:], row_cursor, row_batches, row_stride,
                                      center=center, norm_split=True, chain=chain)
            self.mmd_bf3_prepare(Z, 2 * n, d, Zh, Zl, ZTh, ZTl)
        else:  # the X half is already in place (gather_rows_split ran ahead): only the Y half is produced
            assert ZTh is None
            self.mask_project_forward(logits, data, rows, S, None, None, Z[n:], None, sq[n:], row_cursor, row_batches, row_stride,
                                      center=center, norm_split=True, chain=chain)
            self.mmd_bf3_prepare(Z[n:], n, d, Zh[n:], Zl[n:])

    @staticmethod
    def bf3_fusable(n, d, *lds):
        return d % 4 == 0 and d <= 1024 and n % 8 == 0 and all(int(v) % 4 == 0 for v in lds)

    def gather_rows(self, data, rows, out, sq, row_cursor=None, row_batches=1, row_stride=0, row_offset=0):
        n, d = out.shape[0], data.shape[1]
        X = _np(data)[self._rows(rows, row_cursor, row_batches, row_stride, row_offset, n)]
        out[:, :d].copy_(torch.as_tensor(X))
        if sq is not None:
            sq.copy_(torch.as_tensor((X.astype(np.float64) ** 2).sum(1)))

    def gather_rows_split(self, data, rows, center, out, sq, norm_split=False, Zh=None, Zl=None, row_cursor=None, row_batches=1,
                          row_stride=0, row_offset=0, n=None):
        d = data.shape[1]
        n = int(n if n is not None else (out.shape[0] if out is not None else sq.numel()))
        X = _np(data)[self._rows(rows, row_cursor, row_batches, row_stride, row_offset, n)]
        if center is not None:
            X = X - _np(center)[:d]
        if out is not None:
            out[:, :d].copy_(torch.as_tensor(X))
        hi, lo = self._split(X)
        if Zh is not None:
            Zh[:n, :d] = hi.view(torch.int16)
            Zl[:n, :d] = lo.view(torch.int16)
        if sq is not None:
            Xn = (hi.float() + lo.float()).numpy() if norm_split else X
            sq[:n].copy_(torch.as_tensor((Xn.astype(np.float64) ** 2).sum(1)))

    def upper_softmax_forward(self, logits, S, U):
        u, s = orc.upper_softmax_forward(_np(logits).astype(np.float32))
        S.copy_(torch.as_tensor(s))
        if U is not None:
            U.copy_(torch.as_tensor(u))

    def mask_from_softmax(self, S, U):
        s = _np(S)
        U.copy_(torch.as_tensor(np.where(s < np.float32(1.0 / s.shape[1]), s, np.float32(1.0))))

    def colmax(self, S, row_offset, part, colkey, from_softmax=True):  # `part` unused by the CPU stand-in
        s = _np(S).astype(np.float32)
        n, d = s.shape
        u = np.where(s < np.float32(1.0 / d), s, np.float32(1.0)) if from_softmax else s
        keys = pack_key(u, (row_offset + np.arange(n, dtype=np.uint64))[:, None])
        colkey.copy_(torch.as_tensor(keys.max(axis=0).view(np.int64)))

    def colmax_partial(self, S, row_offset, part, from_softmax=True):
        # the CPU stand-in keeps the finished keys in chunk 0 and zeros elsewhere (max-reduction neutral)
        n, d = S.shape
        part.zero_()
        self.colmax(S, row_offset, None, part[:d], from_softmax)

    def mmd_finalize(self, partial, tiles, colpart, chunks, colkey, n, d, weight, stats, loss, loss_accum=None, accum_scale=1.0,
                     step_counter=None, mode=0, ntiles_main=0):
        if mode == 0:
            self.mmd_reduce(partial, tiles, stats, True)
            if colpart is not None:
                colkey.copy_(torch.as_tensor(_np(colpart).view(np.uint64).reshape(chunks, d).max(axis=0).view(np.int64)))
            self.mmd_loss(stats, colkey if colpart is not None else None, n, d, weight, loss, loss_accum, accum_scale, step_counter)
            return
        # split tail (include/vgan_hip.h): mode 1 = everything over the Gram launch's tiles, mode 2 = the late X-X sums and the loss
        st = torch.zeros(4, dtype=torch.float64)
        if mode == 1:
            self.mmd_reduce(partial[:ntiles_main], tiles[:ntiles_main], st, True)
            v = (float(st[0]) - 2.0 * float(st[1]) + float(st[2])) / (float(n) * n)
            if colpart is not None:
                colkey.copy_(torch.as_tensor(_np(colpart).view(np.uint64).reshape(chunks, d).max(axis=0).view(np.int64)))
                vals = (_np(colkey).view(np.uint64) >> np.uint64(32)).astype(np.uint32).view(np.float32)
                v += weight * float(np.mean(1.0 - vals.astype(np.float64)))
            stats[0], stats[1], stats[2], stats[3] = float(st[0]), float(st[1]), float(st[2]), v
            if step_counter is not None:
                step_counter += 1
        else:
            self.mmd_reduce(partial[ntiles_main:], tiles[ntiles_main:], st, True)
            stats[0] = float(stats[0]) + float(st[0])
            v = float(stats[3]) + float(st[0]) / (float(n) * n)
            loss.fill_(v)
            if loss_accum is not None:
                loss_accum += v * accum_scale

    def finalize_job(self, *args, **kw):
        return (args, kw)

    def mask_backward(self, gU, S, colkey, pen_weight, row_offset, dlogits, nslabs=1, slab_stride=0):
        s = _np(S).astype(np.float32)
        n, d = s.shape
        g = _np(gU)[:, :d].astype(np.float32).copy()
        for q in range(1, nslabs):
            g += torch.as_strided(gU, gU.shape, gU.stride(), gU.storage_offset() + q * slab_stride).numpy()[:, :d]
        if colkey is not None:
            rows = 0xFFFFFFFF - (_np(colkey).view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.int64) - row_offset
            for j in range(d):
                if 0 <= rows[j] < n:
                    g[rows[j], j] += np.float32(-pen_weight / d)
        dlogits.copy_(torch.as_tensor(orc.upper_softmax_backward(g, s)))

    # ---- MMD (tile-table driven, like the kernel)
    def row_sqnorm(self, Z, sq, p):
        sq.copy_(torch.as_tensor((_np(Z)[:, :p].astype(np.float64) ** 2).sum(1)))

    def mmd_gram_general(self, Z, sq, n, p, bw, tiles, multipliers, Wg, wrow0, partial):
        self.mmd_gram(Z, sq, n, p, bw, tiles, False, Wg, wrow0, partial, multipliers=multipliers)

    def rbf_multi_kernel_matrix(self, Z, sq, bw, multipliers, K, dK=None):
        z = _np(Z).astype(np.float64)
        s = (z * z).sum(1)
        L = np.maximum(s[:, None] + s[None, :] - 2.0 * z @ z.T, 0.0)
        scales = (np.float32(float(bw.reshape(-1)[0])) * np.asarray(multipliers, dtype=np.float32)).astype(np.float64)
        K[:, :z.shape[0]].copy_(torch.as_tensor(sum(np.exp(-L / sc) for sc in scales)))
        if dK is not None:
            dK[:, :z.shape[0]].copy_(torch.as_tensor(sum(-np.exp(-L / sc) / sc for sc in scales)))

    def mmd_gram(self, Z, sq, n, p, bw, tiles, calibrate, Wg, wrow0, partial, multipliers=None):
        z = _np(Z)[:, :p].astype(np.float64)
        s = _np(sq).astype(np.float64)
        T = 64
        bwv = float(bw.reshape(-1)[0]) if not calibrate else None
        scales = None
        if not calibrate:
            scales = (orc.rbf_scales(bwv, np.float64) if multipliers is None else
                      (np.float32(bwv) * np.asarray(multipliers, dtype=np.float32)).astype(np.float64))
        part = np.zeros((tiles.shape[0], 4), dtype=np.float32)
        for t, (r0, c0, rlim, clim, fl, *_rest) in enumerate(tiles.tolist()):
            ri, cj = np.arange(r0, min(r0 + T, rlim)), np.arange(c0, min(c0 + T, clim))
            L = np.maximum(s[ri][:, None] + s[cj][None, :] - 2 * z[ri] @ z[cj].T, 0.0)
            if calibrate:
                part[t, 1] = L.sum()
                continue
            K = np.zeros_like(L)
            dK = np.zeros_like(L)
            for sc in scales:
                e = np.exp(-L / sc)
                K += e
                dK -= e / sc
            part[t, 0] = K.sum()
            if (fl & TF_STORE) and Wg is not None:
                sgn = -1.0 if fl & TF_NEG else 1.0
                w = sgn * 2.0 / (n * n) * dK
                wg = _np(Wg)
                wg[np.ix_(ri - wrow0, cj)] = w
                if fl & TF_MIRROR:
                    wg[np.ix_(cj - wrow0, ri)] = w.T
        partial.reshape(-1, 4)[:tiles.shape[0]].copy_(torch.as_tensor(part))

    def mmd_gram_colmax(self, Z, sq, n, p, bw, tiles, Wg, wrow0, partial, S, row_offset, colpart, from_softmax=True):
        self.mmd_gram(Z, sq, n, p, bw, tiles, False, Wg, wrow0, partial)
        self.colmax_partial(S, row_offset, colpart, from_softmax)

    def mmd_reduce(self, partial, tiles, stats, zero_first=True):
        st = np.zeros(4)
        p = _np(partial).reshape(-1, 4).astype(np.float64)
        for t, row in enumerate(tiles.tolist()):
            fl = row[4]
            w = 2.0 if fl & TF_TWICE else 1.0
            st[fl & TF_SLOT] += w * p[t, 0]
            st[3] += (2.0 if (fl & TF_SLOT) == 1 else w) * p[t, 1]
        if zero_first:
            stats.copy_(torch.as_tensor(st))
        else:
            stats.add_(torch.as_tensor(st))

    def mmd_set_bandwidth(self, stats, n, bw):
        N = 2.0 * n
        bw.fill_(float(stats[3]) / (N * N - N))

    def mmd_loss(self, stats, colkey, n, d, weight, loss, loss_accum=None, accum_scale=1.0, step_counter=None):
        st = _np(stats)
        v = (st[0] - 2 * st[1] + st[2]) / (float(n) * n)
        if colkey is not None:
            vals = (_np(colkey).view(np.uint64) >> np.uint64(32)).astype(np.uint32).view(np.float32)
            v += weight * float(np.mean(1.0 - vals.astype(np.float64)))
        loss.fill_(v)
        if loss_accum is not None:
            loss_accum += v * accum_scale
        if step_counter is not None:
            step_counter += 1

    def mmd_backward(self, Wg, Z, wrow0, nr, ncols, p, mul, out, splits=1, slab_stride=0, finalize=None, mul_shift=None):
        if finalize is not None:
            self.mmd_finalize(*finalize[0], **finalize[1])
        kchunk = ((ncols + splits - 1) // splits + 31) // 32 * 32
        for sl in range(splits):
            lo, hi = min(sl * kchunk, ncols), min((sl + 1) * kchunk, ncols)
            w = _np(Wg)[:nr, lo:hi].astype(np.float64)
            z = _np(Z)[:ncols, :p].astype(np.float64)
            r = 2.0 * (w.sum(1, keepdims=True) * z[wrow0:wrow0 + nr] - w @ z[lo:hi])
            if mul is not None:
                r = r * (_np(mul)[:nr, :p] + (_np(mul_shift)[:p] if mul_shift is not None else 0.0))
            o = torch.as_strided(out, out.shape, out.stride(), out.storage_offset() + sl * slab_stride)
            o[:nr, :p].copy_(torch.as_tensor(r))

    # ---- split-bf16 MMD: emulated with torch.bfloat16 roundings; products hi.hi + hi.lo + lo.hi in float64
    @staticmethod
    def _split(x):
        hi = torch.as_tensor(x, dtype=torch.float32).to(torch.bfloat16)
        lo = (torch.as_tensor(x, dtype=torch.float32) - hi.float()).to(torch.bfloat16)
        return hi, lo

    def mmd_bf3_prepare(self, Z, rows, p, Zh, Zl, ZTh=None, ZTl=None):
        hi, lo = self._split(Z[:rows, :p])
        Zh.zero_(), Zl.zero_()
        Zh[:rows, :p] = hi.view(torch.int16)
        Zl[:rows, :p] = lo.view(torch.int16)
        if ZTh is not None:
            ZTh.zero_(), ZTl.zero_()
            ZTh[:p, :rows] = hi.t().contiguous().view(torch.int16)
            ZTl[:p, :rows] = lo.t().contiguous().view(torch.int16)

    @staticmethod
    def _bf(t):
        return t.view(torch.bfloat16).double().numpy()

    def gram_tail_workspace(self, device):
        return torch.zeros(4, dtype=torch.int32)  # (a scheduling aid of the HIP launch: nothing to mirror)

    def mmd_gram_bf3(self, Zh, Zl, sq, n, bw, tiles, Wh, Wl, wrow0, partial, S=None, row_offset=0, colpart=None, from_softmax=True,
                     tile=64, tail_ws=None, rs_part=None):
        assert rs_part is None or (tile == 256 and n % 128 == 0 and Wh is not None)
        zh, zl = self._bf(Zh), self._bf(Zl)
        s = _np(sq).astype(np.float64)
        bwv = float(bw.reshape(-1)[0])
        part = np.zeros((tiles.shape[0], 4), dtype=np.float32)
        tr, tc = (256, 128) if tile == 256 else (tile, tile)  # tile 256 = 256 rows x 128 columns
        for t, (r0, c0, rlim, clim, fl, *_rest) in enumerate(tiles.tolist()):
            ri, cj = np.arange(r0, min(r0 + tr, rlim)), np.arange(c0, min(c0 + tc, clim))
            g = zh[ri] @ zh[cj].T + zh[ri] @ zl[cj].T + zl[ri] @ zh[cj].T
            L = np.maximum(s[ri][:, None] + s[cj][None, :] - 2 * g, 0.0)
            K = np.zeros_like(L)
            dK = np.zeros_like(L)
            for sc in orc.rbf_scales(bwv, np.float64):
                e = np.exp(-L / sc)
                K += e
                dK -= e / sc
            part[t, 0] = K.sum()
            if (fl & TF_STORE) and Wh is not None:
                w = (-1.0 if fl & TF_NEG else 1.0) * 2.0 / (n * n) * dK
                hi, lo = self._split(w)
                Wh[np.ix_(ri - wrow0, cj)] = hi.view(torch.int16)
                Wl[np.ix_(ri - wrow0, cj)] = lo.view(torch.int16)
                if fl & TF_MIRROR:
                    Wh[np.ix_(cj - wrow0, ri)] = hi.t().contiguous().view(torch.int16)
                    Wl[np.ix_(cj - wrow0, ri)] = lo.t().contiguous().view(torch.int16)
                if rs_part is not None:  # per 128-column slot: the sums of the stored hi + lo values
                    v = hi.double().numpy() + lo.double().numpy()
                    rs_part[c0 // 128, torch.as_tensor(ri - wrow0)] = torch.as_tensor(v.sum(1).astype(np.float32))
                    if fl & TF_MIRROR:
                        for h in range(0, len(ri), 128):
                            rs_part[(r0 + h) // 128, torch.as_tensor(cj - wrow0)] = torch.as_tensor(v[h:h + 128].sum(0).astype(np.float32))
        partial.reshape(-1, 4)[:tiles.shape[0]].copy_(torch.as_tensor(part))
        if S is not None:
            self.colmax_partial(S, row_offset, colpart, from_softmax)

    def mmd_backward_bf3(self, Wh, Wl, ZTh, ZTl, Z, wrow0, nr, p, mul, out, splits=1, slab_stride=0, finalize=None, mul_shift=None,
                         tile=0, rs_part=None):
        if finalize is not None:
            self.mmd_finalize(*finalize[0], **finalize[1])
        for q in range(1, splits):  # the whole product goes to slab 0
            torch.as_strided(out, out.shape, out.stride(), out.storage_offset() + q * slab_stride).zero_()
        wh, wl, th, tl = self._bf(Wh)[:nr], self._bf(Wl)[:nr], self._bf(ZTh)[:p], self._bf(ZTl)[:p]
        prod = wh @ th.T + wh @ tl.T + wl @ th.T
        z = _np(Z)[wrow0:wrow0 + nr, :p].astype(np.float64)
        rs = (wh + wl).sum(1, keepdims=True)
        if rs_part is not None and self.mmd_backward_bf3_tile(nr, p, splits, tile) == 256:  # the Gram's per-slot sums, folded in slot order
            rs = _np(rs_part)[:, :nr].astype(np.float32).astype(np.float64).sum(0)[:, None]
        r = 2.0 * (rs * z - prod)
        if mul is not None:
            r = r * (_np(mul)[:nr, :p] + (_np(mul_shift)[:p] if mul_shift is not None else 0.0))
        out[:nr, :p].copy_(torch.as_tensor(r))

    def mmd_backward_bf3_rm(self, Wh, Wl, Zh, Zl, zrows, Z, wrow0, nr, p, mul, out, splits=1, slab_stride=0, finalize=None, mul_shift=None,
                            tile=0, xx=None, rs_part=None):
        kn = (int(zrows) + 63) // 64 * 64
        ZTh = torch.zeros(Zh.shape[1], kn, dtype=torch.int16)
        ZTl = torch.zeros(Zh.shape[1], kn, dtype=torch.int16)
        ZTh[:, :zrows], ZTl[:, :zrows] = Zh[:zrows].t(), Zl[:zrows].t()
        self.mmd_backward_bf3(Wh, Wl, ZTh, ZTl, Z, wrow0, nr, p, mul, out, splits, slab_stride, finalize, mul_shift, tile, rs_part)
        if xx is not None:  # X-X tiles riding in the launch: their sums are NOT seen by the tail of the same launch (it ran above)
            n = xx["Dh"].shape[0] // 2
            self.mmd_gram_bf3(xx["Dh"], xx["Dl"], xx["dsq"], n, xx["bw"], xx["tiles"], None, None, 0, xx["partial"])

    def mmd_backward_bf3_tile(self, nr, p, splits=1, tile=0):
        return int(self.lib.vgan_mmd_backward_bf3_tile(int(nr), int(p), int(splits), int(tile)))

    def linear_backward_params_xx_supported(self, n, kin, out):
        return bool(self.lib.vgan_linear_backward_params_xx_supported(int(n), int(kin), int(out)))

    def linear_backward_params_xx(self, dy, x, dW, xx):
        self.linear_backward_params(dy, x, dW, None)
        n = xx["Dh"].shape[0] // 2
        self.mmd_gram_bf3(xx["Dh"], xx["Dl"], xx["dsq"], n, xx["bw"], xx["tiles"], None, None, 0, xx["partial"])

    def gemm_grouped(self, problems, copy=None, adadelta=None, noise=None, fold=None):
        outs = []
        problems = [q if q[0] == "NT2" else (tuple(q[:4]) if len(q) == 4 or q[4] <= 1 else (q[0], q[1], q[2], q[3], q[4])) for q in problems]
        for kind, A, B, *rest in problems:  # all reads before any write: the products are independent by contract
            a, b = _np(A).astype(np.float64), _np(B).astype(np.float64)
            if kind == "NT2":  # (A . B^T) . D^T, the intermediate rounded to float32 like the kernel's scratch
                outs.append((a @ b.T).astype(np.float32).astype(np.float64) @ _np(rest[1]).astype(np.float64).T)
                continue
            outs.append(a @ b if kind == "NN" else a @ b.T if kind == "NT" else a.T @ b)
        if noise is not None:  # reads the step counter as the launch finds it
            self.noise_normal(noise["next_noise"], noise["seed"], noise["step_counter"], 0, cols=noise["noise_cols"],
                              ones_col=noise["noise_ones_col"])
        if copy is not None:
            copy[1].copy_(copy[0])
        for q, r in zip(problems, outs):
            C = q[3]
            if q[0] != "NT2" and len(q) > 4:  # split-K slabs: the CPU stand-in puts the whole product in slab 0
                C.zero_()
                C = C[0]
            C.copy_(torch.as_tensor(r))
        if fold is not None:
            self.mmd_finalize(*fold[0], **fold[1])
        if adadelta is not None:
            a = adadelta
            grads = [q[3] for q in problems] + ([a["extra_grad"]] if a.get("extra_grad") is not None else [])
            for (w, off_w, off_b, out, inp), G in zip(a["layers"], grads):
                for off, cnt, g in ((off_w, out * inp, G[:out, :inp].reshape(-1)), (off_b, out, G[:out, inp].reshape(-1))):
                    sl = slice(off, off + cnt)
                    pn, sn, an = orc.adadelta_step(_np(a["p"])[sl].astype(np.float64), _np(g).astype(np.float64) * a.get("grad_scale", 1.0),
                                                   _np(a["sq"])[sl].astype(np.float64), _np(a["acc"])[sl].astype(np.float64), a["lr"],
                                                   a.get("weight_decay", 0.0), a.get("rho", 0.9), a.get("eps", 1e-6))
                    a["p"][sl].copy_(torch.as_tensor(pn))
                    a["sq"][sl].copy_(torch.as_tensor(sn))
                    a["acc"][sl].copy_(torch.as_tensor(an))
                w[:out, :inp].copy_(a["p"][off_w:off_w + out * inp].view(out, inp))
                w[:out, inp].copy_(a["p"][off_b:off_b + out])

    def mse_grad(self, target, pred, gscale, part, g):
        n, d = pred.shape
        df = _np(pred)[:, :d].astype(np.float64) - _np(target)[:, :d].astype(np.float64)
        rows = (df * df).sum(1)
        pad = np.zeros((n + 3) // 4 * 4)
        pad[:n] = rows
        part[:(n + 3) // 4].copy_(torch.as_tensor(pad.reshape(-1, 4).sum(1)))
        g[:, :d].copy_(torch.as_tensor(float(gscale) * df))

    def sum_f64(self, src, count, scale, out, accumulate=False):
        v = float(_np(src)[:count].sum() * scale)
        out[0] = (float(out[0]) if accumulate else 0.0) + v

    def rbf_kernel_matrix(self, Z, sq, alpha, K):
        z = _np(Z).astype(np.float64)
        s = (z * z).sum(1)
        L = np.maximum(s[:, None] + s[None, :] - 2.0 * z @ z.T, 0.0)
        k = np.exp(-float(alpha) * L)
        np.fill_diagonal(k, 1.0)
        K[:, :z.shape[0]].copy_(torch.as_tensor(k))

    def rows_dot(self, A, B, out, broadcast_b=False):
        a, b = _np(A).astype(np.float64), _np(B).astype(np.float64)
        out[:a.shape[0]].copy_(torch.as_tensor((a * (b.reshape(1, -1) if broadcast_b else b)).sum(1)))

    # ---- input pipeline (host evaluation of the library's own permutation: same numbers as the device kernel)
    def shuffle_epoch(self, perm, train_size, seed, epoch):
        flat = perm.view(-1)
        for i in range(flat.numel()):
            flat[i] = self.shuffle_index(i, train_size, seed, epoch)

    def shuffle_index(self, i, train_size, seed, epoch):
        return int(self.lib.vgan_shuffle_index(int(i), int(train_size), int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch) & 0xFFFFFFFFFFFFFFFF))

    def mask_unique(self, masks):
        u, c = np.unique(masks.numpy(), axis=0, return_counts=True)
        return torch.as_tensor(u), torch.as_tensor(c)

    # ---- optimiser / noise
    def adadelta_step(self, p, g, sq, acc, lr, rho=0.9, eps=1e-6, weight_decay=0.0, grad_scale=1.0, nslabs=1, slab_stride=0):
        if nslabs > 1:
            g = sum(torch.as_strided(g, g.shape, g.stride(), g.storage_offset() + q * slab_stride) for q in range(nslabs))
        pn, sn, an = orc.adadelta_step(_np(p).astype(np.float64), _np(g).astype(np.float64) * grad_scale, _np(sq).astype(np.float64),
                                       _np(acc).astype(np.float64), lr, weight_decay, rho, eps)
        p.copy_(torch.as_tensor(pn))
        sq.copy_(torch.as_tensor(sn))
        acc.copy_(torch.as_tensor(an))

    def adadelta_step_packed(self, p, pmap, g_packed, w_packed, sq, acc, lr, rho=0.9, eps=1e-6, weight_decay=0.0, grad_scale=1.0,
                             next_noise=None, noise_cols=0, noise_ones_col=-1, seed=0, step_counter=None):
        if next_noise is not None:
            self.noise_normal(next_noise, seed, step_counter, 0, cols=noise_cols, ones_col=noise_ones_col)
        m = pmap.long()
        live = m >= 0
        g = torch.zeros_like(p)
        g[live] = g_packed[m[live]]
        self.adadelta_step(p, g, sq, acc, lr, rho, eps, weight_decay, grad_scale)
        w_packed[m[live]] = p[live]

    def noise_normal(self, z, seed, step_counter, stream_id=0, cols=None, ones_col=-1):
        g = torch.Generator()
        g.manual_seed((int(seed) * 1000003 + int(step_counter.item()) * 7919 + int(stream_id)) % (2 ** 63))
        cols = z.shape[1] if cols is None else cols
        z[:, :cols].copy_(torch.randn(z.shape[0], cols, generator=g))
        if ones_col >= 0:
            z[:, ones_col] = 1.0

    def homogeneous_pack(self, layers, unpack=False):
        for W, b, P in layers:
            out, kin = W.shape
            if unpack:
                W.copy_(P[:out, :kin])
                b.copy_(P[:out, kin])
            else:
                P[:out, :kin].copy_(W)
                P[:out, kin].copy_(b)
                P[out, :kin + 1] = 0.0
                P[out, kin] = 1.0
