"""Step engine of ``VGAN.fit`` (kernel-learning variant, reference src/vgan.py:234-337) over a kernel provider.

The reference alternates a detector epoch with ``iternum_g`` generator epochs.  Because of the ``Variable(...)``
quirk (src/vgan.py:308-310 detaches the generator output) the generator never receives a gradient, so there are two
step kinds, both launched here kernel by kernel on preallocated buffers, without autograd:

  detector step (src/vgan.py:253-289)
      noise -> Generator_big -> upper_softmax -> U * batch                        (no gradient)
      detector on the 2n stacked rows [batch ; U * batch]: Encoder (4 Linear) -> enc [2n, L], Decoder (4 Linear) -> dec [2n, d]
      loss_D = -( MMD(enc_X, enc_P, U)  - 0.1 mse(batch, dec_X) - 0.1 mse(U * batch, dec_P) )
      backward through decoder and encoder, Adadelta on the detector parameters that still require a gradient
  generator-phase step (src/vgan.py:295-329): the same forward, loss_G = MMD(enc_X, enc_P, U); nothing is updated.

The stacked rows are exactly the [X ; Y] operand layout of the MMD kernels (p = L), and both detector passes share
one set of launches.  Gradients are formed for  -loss_D = MMD - 0.1 mse - 0.1 mse  (the sign the kernels produce) and
Adadelta is called with grad_scale = -1.  Reference quirks kept: the encoder is frozen for good by the first generator
phase (src/vgan.py:319-320) while the decoder is re-enabled by every detector step (:257-258).
"""
import torch

from .trainer import ADADELTA_EPS, ADADELTA_RHO, FlatParams, _round4


class KLStepEngine:
    def __init__(self, ops, generator, detector, data, batch_size, lr_D, weight_decay, penalty_weight, use_graph=True):
        self.ops = ops
        # each step kind is captured into a HIP graph at its second use (the very first step calibrates the bandwidth
        # eagerly); the batch indices and the noise are copied into fixed device buffers before every replay
        self.use_graph = bool(use_graph) and data.is_cuda
        self.graphs = {}
        self.dev = data.device
        self.data = data
        n = self.n = int(batch_size)
        d = self.d = data.shape[1]
        dp = self.dp = _round4(d)
        self.lr, self.wd, self.pen = float(lr_D), float(weight_decay), float(penalty_weight)
        f32 = dict(dtype=torch.float32, device=self.dev)

        self.gen = [m for m in generator.main if isinstance(m, torch.nn.Linear)]
        self.enc = [m for m in detector.encoder.main if isinstance(m, torch.nn.Linear)]
        self.dec = [m for m in detector.decoder.main if isinstance(m, torch.nn.Linear)]
        assert len(self.gen) == 4 and len(self.enc) == 4 and len(self.dec) == 4
        L = self.L = self.gen[0].in_features
        self.Lp = _round4(L)
        det_params = [q for m in self.enc + self.dec for q in (m.weight, m.bias)]
        self.fp = FlatParams(det_params, self.dev)          # detector parameters become views of one flat buffer
        self.W = [self.fp.view(self.fp.flat, 2 * k) for k in range(8)]
        self.b = [self.fp.view(self.fp.flat, 2 * k + 1) for k in range(8)]
        # the weight gradients contract over the 2n stacked rows while their outputs are small: the row range is cut into
        # slabs (partial sums, fixed order) so that a launch fills the chip; Adadelta sums the slabs itself
        self.splits = max(1, min(8, (2 * n) // 256))
        self.gslab = torch.zeros(self.splits, self.fp.total, dtype=torch.float32, device=self.dev)
        self.dW = [self.fp.view(self.gslab[0], 2 * k) for k in range(8)]
        self.db = [self.fp.view(self.gslab[0], 2 * k + 1) for k in range(8)]
        self.enc_end = self.fp.offsets[8]                   # flat range [0, enc_end) = encoder, [enc_end, total) = decoder

        # generator forward (no gradient)
        self.z = torch.zeros(n, L, **f32)
        self.gact = [self.z] + [torch.zeros(n, m.out_features, **f32) for m in self.gen]
        self.S = torch.zeros(n, d, **f32)
        self.U = torch.zeros(n, d, **f32)
        self.perm = torch.zeros(1, n, dtype=torch.int32, device=self.dev)
        self.sqxp = torch.zeros(2 * n, **f32)               # row norms of [batch ; U*batch] (a by-product nobody reads here)
        # detector on the stacked rows: act[0] = [batch ; U*batch] (pad columns zero), act[4] = enc, act[8] = dec
        widths = [m.out_features for m in self.enc + self.dec]
        self.XP = torch.zeros(2 * n, dp, **f32)
        self.act = [self.XP[:, :d]] + [torch.zeros(2 * n, _round4(w), **f32)[:, :w] for w in widths]
        self.encZ = self.act[4]                             # [2n, L] view of a [2n, Lp] buffer (pad columns stay zero)
        self.encZp = self.encZ.as_strided((2 * n, self.Lp), (self.Lp, 1))
        # gradients of the activations; at the encoder output two contributions meet (MMD and decoder): two slabs
        self.dact = [None] + [torch.zeros(2 * n, w, **f32) for w in widths]
        self.msplits = max(1, min(8, (2 * n) // 256))           # split-K slabs of the MMD backward (32 output tiles only)
        self.denc = torch.zeros(1 + self.msplits, 2 * n, self.Lp, **f32)
        self.dact[4] = self.denc[0][:, :L]
        self.mse_part = torch.zeros(2, (n + 3) // 4, dtype=torch.float64, device=self.dev)
        # MMD on the encodings (gradient for all 2n rows)
        self.sq = torch.zeros(2 * n, **f32)
        self.tiles = ops.build_tiles(n, 2, device=self.dev)
        self.tiles0 = ops.build_tiles(n, 0, device=self.dev)
        self.partial = torch.zeros(max(self.tiles.shape[0], self.tiles0.shape[0]), 4, **f32)
        self.Wg = torch.zeros(2 * n, 2 * n, **f32)
        self.stats = torch.zeros(4, dtype=torch.float64, device=self.dev)
        self.bw = torch.zeros(1, **f32)
        self.has_bw = False
        self.colpart = torch.zeros(ops.colmax_chunks(n) * d, dtype=torch.int64, device=self.dev)
        self.colkey = torch.zeros(d, dtype=torch.int64, device=self.dev)
        self.mmd = torch.zeros(1, **f32)                    # MMD^2 + penalty of the last step
        self.mse = torch.zeros(2, **f32)                    # mse(batch, dec_X), mse(U*batch, dec_P) of the last step
        self.acc_mmd = torch.zeros(1, **f32)                # epoch accumulators (sum over steps)
        self.acc_mse = torch.zeros(2, **f32)

    # ---- host-side controls -------------------------------------------------------------------------
    def set_bandwidth(self, value):
        self.bw.fill_(float(value))
        self.has_bw = True

    def epoch_sums(self):
        """(sum of MMD terms, sum of mse_X, sum of mse_P) since the last call -- one host sync."""
        out = (float(self.acc_mmd.item()), float(self.acc_mse[0].item()), float(self.acc_mse[1].item()))
        self.acc_mmd.zero_()
        self.acc_mse.zero_()
        return out

    # ---- pieces -------------------------------------------------------------------------------------------
    def _feed(self, idx, noise):
        self.perm.copy_(idx.to(dtype=torch.int32).view(1, self.n), non_blocking=True)
        self.z.copy_(noise.to(dtype=torch.float32), non_blocking=True)

    def _run(self, key, body):
        """Eager until the bandwidth exists, then one captured graph per step kind."""
        if not self.use_graph or not self.has_bw:
            body()
            return
        g = self.graphs.get(key)
        if g is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            self.graphs[key] = g
        g.replay()

    def _forward(self, want_grad):
        ops, n, d = self.ops, self.n, self.d
        for k, m in enumerate(self.gen):
            ops.linear_forward(self.gact[k], m.weight.detach(), m.bias.detach(), self.gact[k + 1])
        ops.mask_project_forward(self.gact[4], self.data, self.perm, self.S, self.U, self.XP[:n], self.XP[n:], self.sqxp[:n], self.sqxp[n:])
        for k in range(8):
            ops.linear_forward(self.act[k], self.W[k], self.b[k], self.act[k + 1])
        Z, p = self.encZp, self.Lp
        ops.row_sqnorm(Z, self.sq, p)
        if not self.has_bw:  # first call of the (process-wide) RBF calibrates its bandwidth (Mmd_loss_constrained.py:16-20)
            ops.mmd_gram(Z, self.sq, n, p, None, self.tiles0, True, None, 0, self.partial)
            ops.mmd_reduce(self.partial, self.tiles0, self.stats, True)
            ops.mmd_set_bandwidth(self.stats, n, self.bw)
            self.has_bw = True
        tiles = self.tiles if want_grad else self.tiles0
        ops.mmd_gram(Z, self.sq, n, p, self.bw, tiles, False, self.Wg if want_grad else None, 0, self.partial)
        ops.mmd_reduce(self.partial, tiles, self.stats, True)
        ops.colmax(self.S, 0, self.colpart, self.colkey, True)
        ops.mmd_loss(self.stats, self.colkey, n, d, self.pen, self.mmd, self.acc_mmd, 1.0, None)

    def generator_phase_step(self, idx, noise):
        """Loss evaluation of the generator phase: accumulates MMD(enc_X, enc_P, U) (src/vgan.py:295-329)."""
        self._feed(idx, noise)
        self._run("g", lambda: self._forward(want_grad=False))

    def detector_step(self, idx, noise, train_encoder):
        self._feed(idx, noise)
        self._run(("d", bool(train_encoder)), lambda: self._detector_body(bool(train_encoder)))

    def _detector_body(self, train_encoder):
        ops, n, d, L = self.ops, self.n, self.d, self.L
        # the MMD term reaches only the encoder: with the encoder frozen neither its gradient weights nor its backward run
        self._forward(want_grad=train_encoder)
        # gradients of G = MMD - 0.1 mse_X - 0.1 mse_P  (= -loss_D)
        gs = -0.1 * 2.0 / (float(n) * d)
        for h in range(2):
            rows = slice(h * n, (h + 1) * n)
            ops.mse_grad(self.XP[rows, :d], self.act[8][rows], gs, self.mse_part[h], self.dact[8][rows])
            ops.sum_f64(self.mse_part[h], (n + 3) // 4, 1.0 / (float(n) * d), self.mse[h:h + 1])
            ops.sum_f64(self.mse_part[h], (n + 3) // 4, 1.0 / (float(n) * d), self.acc_mse[h:h + 1], accumulate=True)
        if train_encoder:
            ops.mmd_backward(self.Wg, self.encZp, 0, 2 * n, 2 * n, self.Lp, None, self.denc[1], self.msplits, 2 * n * self.Lp)
        # decoder: layers 7..4 of the stacked chain
        g = self.dact[8]
        for k in (7, 6, 5, 4):
            ops.linear_backward_params(g, self.act[k], self.dW[k], self.db[k], self.splits, self.fp.total)
            if k > 4:
                ops.linear_backward_input(g, self.W[k], self.dact[k])
                g = self.dact[k]
            elif train_encoder:
                ops.linear_backward_input(g, self.W[k], self.denc[0][:, :L])
        adadelta = dict(lr=self.lr, rho=ADADELTA_RHO, eps=ADADELTA_EPS, weight_decay=self.wd, grad_scale=-1.0, nslabs=self.splits,
                        slab_stride=self.fp.total)
        lo = self.enc_end
        ops.adadelta_step(self.fp.flat[lo:], self.gslab[0][lo:], self.fp.sq[lo:], self.fp.acc[lo:], **adadelta)
        if train_encoder:
            ops.reduce_slabs(self.denc, 2 * n * self.Lp, 1 + self.msplits, self.denc[0].view(-1))  # d enc = decoder path + MMD slabs
            g = self.dact[4]
            for k in (3, 2, 1, 0):
                ops.linear_backward_params(g, self.act[k], self.dW[k], self.db[k], self.splits, self.fp.total)
                if k:
                    ops.linear_backward_input(g, self.W[k], self.dact[k])
                    g = self.dact[k]
            ops.adadelta_step(self.fp.flat[:lo], self.gslab[0][:lo], self.fp.sq[:lo], self.fp.acc[:lo], **adadelta)
