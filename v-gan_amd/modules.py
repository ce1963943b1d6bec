"""Operator boundary: the reference's ``nn.Module`` classes with the same names, constructor
arguments, attributes and state_dict keys, whose forward/backward run on libvgan_hip.so.

  Generator_big, upper_softmax   <- src/models/Generator.py:6-22, 58-70
  Encoder, Decoder, Detector     <- src/models/Detector.py:5-48
  RBF, MMDLossConstrained        <- src/models/Mmd_loss_constrained.py:5-50

Tensors must live on a HIP device; there is no CPU path (HipOps raises).
"""
import torch
from torch import nn

from .ops import default_ops


def _round4(v):
    return (v + 3) // 4 * 4


class _LinearFn(torch.autograd.Function):
    """F.linear on the fp32 MFMA (vgan_linear_forward / _backward_input / _backward_params)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ops = default_ops()
        x = x.contiguous()
        y = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32, device=x.device)
        ops.linear_forward(x, W, b, y)
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = default_ops()
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ops.linear_backward_input(dy, W, dx)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.empty_like(W)
            db = torch.empty(W.shape[0], dtype=torch.float32, device=W.device) if ctx.has_bias else None
            ops.linear_backward_params(dy, x, dW, db)
        return dx, dW, db


def hip_linear(x, layer):
    return _LinearFn.apply(x, layer.weight, layer.bias)


class _UpperSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ops = default_ops()
        x = x.contiguous()
        S = torch.empty_like(x)
        U = torch.empty_like(x)
        ops.upper_softmax_forward(x, S, U)
        ctx.save_for_backward(S)
        return U

    @staticmethod
    def backward(ctx, gU):
        ops = default_ops()
        (S,) = ctx.saved_tensors
        dl = torch.empty_like(S)
        ops.mask_backward(gU.contiguous(), S, None, 0.0, 0, dl)
        return dl


class upper_softmax(nn.Module):
    """Upper softmax activation (src/models/Generator.py:6-22): softmax entries >= 1/d snap to 1."""

    def __init__(self):
        super().__init__()

    def forward(self, x):
        return _UpperSoftmaxFn.apply(x)


class _Chain(nn.Module):
    """``self.main = nn.Sequential(Linear...)`` with the reference's state_dict keys; forward runs
    each Linear through the HIP GEMM instead of ATen."""

    def _run(self, x):
        for m in self.main:
            x = hip_linear(x, m) if isinstance(m, nn.Linear) else m(x)
        return x


class Generator_big(_Chain):
    """src/models/Generator.py:58-70: Linear(L,2L) -> (2L,4L) -> (4L,8L) -> (8L,d) -> upper_softmax."""

    def __init__(self, latent_size, img_size):
        super().__init__()
        self.main = nn.Sequential(
            nn.Linear(latent_size, 2 * latent_size),
            nn.Linear(2 * latent_size, 4 * latent_size),
            nn.Linear(4 * latent_size, 8 * latent_size),
            nn.Linear(8 * latent_size, img_size),
            upper_softmax(),
        )

    def forward(self, input):
        return self._run(input)


class Encoder(_Chain):
    """src/models/Detector.py:5-18."""

    def __init__(self, latent_size, img_size):
        super().__init__()
        self.main = nn.Sequential(
            nn.Linear(img_size, 8 * latent_size),
            nn.Linear(8 * latent_size, 4 * latent_size),
            nn.Linear(4 * latent_size, 2 * latent_size),
            nn.Linear(2 * latent_size, latent_size),
        )

    def forward(self, input):
        return self._run(input)


class Decoder(_Chain):
    """src/models/Detector.py:21-34."""

    def __init__(self, latent_size, img_size):
        super().__init__()
        self.main = nn.Sequential(
            nn.Linear(latent_size, 2 * latent_size),
            nn.Linear(2 * latent_size, 4 * latent_size),
            nn.Linear(4 * latent_size, 8 * latent_size),
            nn.Linear(8 * latent_size, img_size),
        )

    def forward(self, input):
        return self._run(input)


class Detector(nn.Module):
    """src/models/Detector.py:37-48: forward(x) -> (enc [n,L], dec [n,d])."""

    def __init__(self, latent_size, img_size, encoder, decoder):
        super().__init__()
        self.encoder = encoder(latent_size, img_size)
        self.decoder = decoder(latent_size, img_size)

    def forward(self, input):
        enc_X = self.encoder(input)
        dec_X = self.decoder(enc_X)
        return enc_X.view(input.size(0), -1), dec_X.view(input.size(0), -1)


class RBF(nn.Module):
    """src/models/Mmd_loss_constrained.py:5-26.  Holds the multipliers and the bandwidth, which is
    computed on the first call and frozen.  The HIP epilogue evaluates the five kernels with one
    exp and a squaring chain, which requires the reference's defaults (n_kernels=5, mul_factor=2)."""

    def __init__(self, n_kernels=5, mul_factor=2.0, bandwidth=None):
        super().__init__()
        self.n_kernels, self.mul_factor = n_kernels, mul_factor
        self.bandwidth_multipliers = mul_factor ** (torch.arange(n_kernels) - n_kernels // 2)
        self.bandwidth = bandwidth

    def _check_supported(self):
        if self.n_kernels != 5 or float(self.mul_factor) != 2.0:
            raise NotImplementedError("the HIP MMD kernel implements the reference's default RBF(n_kernels=5, mul_factor=2.0)")

    def forward(self, X):
        raise NotImplementedError(
            "RBF.forward would materialise the N x N kernel matrix; use MMDLossConstrained (fused, never materialised)")


class _MMDLossFn(torch.autograd.Function):
    """loss = MMD^2(X, Y) + weight * mean_j(1 - max_i U_ij), fused Gram/kernel-sum tiles on the MFMA."""

    @staticmethod
    def forward(ctx, X, Y, U, weight, kernel):
        ops = default_ops()
        dev = X.device
        n, p = X.shape
        d = U.shape[1]
        pp = _round4(p)
        need_x, need_y = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_mode = 2 if need_x else (1 if need_y else 0)
        Z = torch.zeros(2 * n, pp, dtype=torch.float32, device=dev)
        Z[:n, :p].copy_(X)
        Z[n:, :p].copy_(Y)
        sq = torch.empty(2 * n, dtype=torch.float32, device=dev)
        ops.row_sqnorm(Z, sq, pp)
        tiles = ops.build_tiles(n, grad_mode, device=dev)
        partial = torch.empty(tiles.shape[0], 4, dtype=torch.float32, device=dev)
        stats = torch.empty(4, dtype=torch.float64, device=dev)
        if kernel.bandwidth is None:  # first call calibrates and freezes (Mmd_loss_constrained.py:16-20)
            bw = torch.empty(1, dtype=torch.float32, device=dev)
            ops.mmd_gram(Z, sq, n, pp, None, tiles, True, None, 0, partial)
            ops.mmd_reduce(partial, tiles, stats, True)
            ops.mmd_set_bandwidth(stats, n, bw)
            kernel.bandwidth = bw.view(())
        bw = kernel.bandwidth
        if not (torch.is_tensor(bw) and bw.is_cuda and bw.dtype == torch.float32):
            bw = torch.as_tensor(float(bw), dtype=torch.float32, device=dev)
        bw = bw.reshape(1)
        Wg = wrow0 = None
        if grad_mode == 1:
            Wg, wrow0 = torch.empty(n, 2 * n, dtype=torch.float32, device=dev), n
        elif grad_mode == 2:
            Wg, wrow0 = torch.empty(2 * n, 2 * n, dtype=torch.float32, device=dev), 0
        ops.mmd_gram(Z, sq, n, pp, bw, tiles, False, Wg, wrow0 or 0, partial)
        ops.mmd_reduce(partial, tiles, stats, True)
        Uc = U.detach().contiguous()
        colpart = torch.empty(ops.colmax_chunks(n) * d, dtype=torch.int64, device=dev)
        colkey = torch.empty(d, dtype=torch.int64, device=dev)
        ops.colmax(Uc, 0, colpart, colkey, False)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ops.mmd_loss(stats, colkey, n, d, float(weight), loss)
        ctx.save_for_backward(Z, colkey)
        ctx.Wg, ctx.wrow0, ctx.dims, ctx.weight, ctx.grad_mode = Wg, wrow0, (n, p, pp, d), float(weight), grad_mode
        return loss.view(())

    @staticmethod
    def backward(ctx, gl):
        ops = default_ops()
        Z, colkey = ctx.saved_tensors
        n, p, pp, d = ctx.dims
        dX = dY = dU = None
        if ctx.grad_mode:
            nr = n if ctx.grad_mode == 1 else 2 * n
            out = torch.empty(nr, pp, dtype=torch.float32, device=Z.device)
            ops.mmd_backward(ctx.Wg, Z, ctx.wrow0, nr, 2 * n, pp, None, out)
            out = out[:, :p] * gl
            if ctx.grad_mode == 1:
                dY = out
            else:
                dX, dY = out[:n], out[n:]
        if ctx.needs_input_grad[2]:
            rows = 0xFFFFFFFF - (colkey & 0xFFFFFFFF)  # arg-max row per column (lowest row on ties)
            dU = torch.zeros(n, d, dtype=torch.float32, device=Z.device)
            dU[rows, torch.arange(d, device=Z.device)] = -ctx.weight / d
            dU = dU * gl
        return dX, dY, dU, None, None


class MMDLossConstrained(nn.Module):
    """src/models/Mmd_loss_constrained.py:29-50.  As in the reference, the default ``kernel=RBF()`` is
    evaluated once at class-definition time: every instance created without an explicit kernel
    shares ONE RBF and therefore one frozen bandwidth per process (reference quirk, kept)."""

    def __init__(self, weight, kernel=RBF()):
        super().__init__()
        self.kernel = kernel
        self.weight = weight

    def forward(self, X, Y, U):
        self.kernel._check_supported()
        out = _MMDLossFn.apply(X.contiguous().float(), Y.contiguous().float(), U, self.weight, self.kernel)
        self.bandwidth = self.kernel.bandwidth
        self.bandwidth_multipliers = self.kernel.bandwidth_multipliers
        return out
