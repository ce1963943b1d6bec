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
one set of launches.  Encoder and decoder have no activation between their Linear layers either (src/models/Detector.py:8-13,
24-29), so each is run as ONE matrix chain in homogeneous coordinates, exactly like the generator of the no-kl engine
(trainer.py): enc = [x|1] . (Et_4 Et_3 Et_2 Et_1)^T, dec = [enc|1] . (Dt_4 .. Dt_1)^T, weight gradients from
M_4 = dy^T [x|1] through M_{k-1} = Wt_k^T M_k and [dW_k | db_k] = M_k At_{k-1}^T -- ~1 GFLOP per step instead of the ~9
of eight layered GEMMs over the 2n rows.  Gradients are formed for  -loss_D = MMD - 0.1 mse - 0.1 mse  (the sign the kernels produce) and
Adadelta is called with grad_scale = -1.  Reference quirks kept: the encoder is frozen for good by the first generator
phase (src/vgan.py:319-320) while the decoder is re-enabled by every detector step (:257-258).
"""
import torch

from .trainer import ADADELTA_EPS, ADADELTA_RHO, FlatParams, _round4


class CollapsedChain:
    """Four bias-Linear layers (parameters 2*k0 .. 2*k0+7 of a FlatParams, layer order input -> output) as one matrix chain
    in homogeneous coordinates: Wt_k = [[W_k, b_k],[0, 1]] zero-padded to multiples of 4, prefix products
    At_k = Wt_k .. Wt_1, suffix products B_3 = Wt_4 Wt_3, B_2 = B_3 Wt_2 (so that every product of the forward and of the
    backward is at most two dependent launches deep), packed gradients Gt_k = [dW_k | db_k], and the index map through
    which Adadelta reads the packed gradient and keeps the packed weight current."""

    def __init__(self, ops, fp, k0, dev):
        self.ops, self.fp = ops, fp
        f32 = dict(dtype=torch.float32, device=dev)
        shapes = [fp.shapes[2 * (k0 + k)] for k in range(4)]
        w = self.widths = [shapes[0][1]] + [sh[0] for sh in shapes]
        e = self.e = [_round4(v + 1) for v in w]
        poff = [0]
        for k in range(1, 5):
            poff.append(poff[-1] + e[k] * e[k - 1])
        self.Wt_all, self.Gt_all = torch.zeros(poff[-1], **f32), torch.zeros(poff[-1], **f32)
        self.Wt = [None] + [self.Wt_all[poff[k - 1]:poff[k]].view(e[k], e[k - 1]) for k in range(1, 5)]
        self.Gt = [None] + [self.Gt_all[poff[k - 1]:poff[k]].view(e[k], e[k - 1]) for k in range(1, 5)]
        self.At = [None, self.Wt[1]] + [torch.zeros(e[k], e[0], **f32) for k in range(2, 5)]
        self.M = [None, self.Gt[1]] + [torch.zeros(e[k], e[0], **f32) for k in range(2, 5)]
        self.B3, self.B2 = torch.zeros(e[4], e[2], **f32), torch.zeros(e[4], e[1], **f32)
        first = 2 * k0
        self.lo = fp.offsets[first]
        self.hi = fp.offsets[first + 8] if first + 8 < len(fp.offsets) else fp.total
        pmap = torch.full((self.hi - self.lo,), -1, dtype=torch.int32)
        for k in range(1, 5):
            wk, wk1 = w[k], w[k - 1]
            r = torch.arange(wk, dtype=torch.int32)[:, None] * e[k - 1]
            ow, ob = fp.offsets[first + 2 * (k - 1)] - self.lo, fp.offsets[first + 2 * (k - 1) + 1] - self.lo
            pmap[ow:ow + wk * wk1] = (poff[k - 1] + r + torch.arange(wk1, dtype=torch.int32)[None, :]).reshape(-1)
            pmap[ob:ob + wk] = (poff[k - 1] + r + wk1).reshape(-1)
        self.pmap = pmap.to(dev)
        layers = [(fp.view(fp.flat, first + 2 * (k - 1)), fp.view(fp.flat, first + 2 * (k - 1) + 1), self.Wt[k]) for k in range(1, 5)]
        ops.homogeneous_pack(layers, unpack=False)
        self.refresh()

    def refresh(self):
        """Prefix and suffix products of the current packed weights."""
        Wt, At = self.Wt, self.At
        self.ops.gemm_grouped([("NN", Wt[2], At[1], At[2]), ("NN", Wt[4], Wt[3], self.B3)])
        self.ops.gemm_grouped([("NN", Wt[3], At[2], At[3]), ("NN", self.B3, At[2], At[4]), ("NN", self.B3, Wt[2], self.B2)])

    def forward(self, xh, y):
        """y [rows, w4] = [x|1] . At_4^T  (xh [rows, e0] carries the ones column at index w0)."""
        self.ops.linear_forward(xh, self.At[4][:self.widths[4]], None, y)

    def input_grad(self, dy, dx):
        """dx [rows, w0] = dy . At_4[:, :w0]: the gradient of the chain's real inputs (not of the homogeneous coordinate)."""
        self.ops.linear_backward_input(dy, self.At[4][:dy.shape[1], :dx.shape[1]], dx)

    def backward(self, dy, xh):
        """Packed gradients Gt_k from dy [rows, out] (out = w4, or w4 zero-padded to a multiple of 4) and xh [rows, e0]."""
        ops, M, Gt, At = self.ops, self.M, self.Gt, self.At
        ops.linear_backward_params(dy, xh, M[4][:dy.shape[1]], None)
        ops.gemm_grouped([("TN", self.Wt[4], M[4], M[3]), ("TN", self.B3, M[4], M[2]), ("TN", self.B2, M[4], M[1])])
        ops.gemm_grouped([("NT", M[4], At[3], Gt[4]), ("NT", M[3], At[2], Gt[3]), ("NT", M[2], At[1], Gt[2])])

    def update(self, **adadelta):
        fp, lo, hi = self.fp, self.lo, self.hi
        self.ops.adadelta_step_packed(fp.flat[lo:hi], self.pmap, self.Gt_all, self.Wt_all, fp.sq[lo:hi], fp.acc[lo:hi], **adadelta)
        self.refresh()


class KLStepEngine:
    def __init__(self, ops, generator, detector, data, batch_size, lr_D, weight_decay, penalty_weight, use_graph=True,
                 batches_per_epoch=1, noise="host", seed=777):
        """batches_per_epoch > 1: the resident feed of the no-kl engine -- the epoch's shuffled indices live in a device table
        [batches_per_epoch, n] (set_epoch_batches / shuffle_epoch) and a device-side step counter, advanced by every step's
        tail, picks the row, so a step needs no host copy.  noise = "device": the Philox draw keyed by (seed, step counter) is
        the first launch of every step; "host": the caller provides it (set_noise, or the `noise` argument of a step)."""
        self.ops = ops
        # each step kind is captured into a HIP graph at its second use (the very first step calibrates the bandwidth
        # eagerly); nothing but the graph replay happens per step when the feed is resident
        self.use_graph = bool(use_graph) and data.is_cuda
        assert noise in ("host", "device")
        self.noise_mode = noise
        self.seed = int(seed)
        self.nb = int(batches_per_epoch)
        self.steps_per_graph = max(1, min(16, self.nb))
        self.graphs = {}
        self.dev = data.device
        self.data = data
        n = self.n = int(batch_size)
        d = self.d = data.shape[1]
        dp = self.dp = _round4(d)
        self.lr, self.wd, self.pen = float(lr_D), float(weight_decay), float(penalty_weight)
        f32 = dict(dtype=torch.float32, device=self.dev)

        self.gen = [m for m in generator.main if isinstance(m, torch.nn.Linear)]
        # The generator never trains in VGAN.fit (module docstring): its four bias-Linear layers are collapsed ONCE into
        # logits = [z|1] . Gt_4..Gt_1^T -- one product per step instead of four.
        self.gfp = FlatParams([q for m in self.gen for q in (m.weight, m.bias)], self.dev)
        self.G = CollapsedChain(ops, self.gfp, 0, self.dev)
        enc = [m for m in detector.encoder.main if isinstance(m, torch.nn.Linear)]
        dec = [m for m in detector.decoder.main if isinstance(m, torch.nn.Linear)]
        assert len(self.gen) == 4 and len(enc) == 4 and len(dec) == 4
        L = self.L = self.gen[0].in_features
        det_params = [q for m in enc + dec for q in (m.weight, m.bias)]
        self.fp = FlatParams(det_params, self.dev)          # detector parameters become views of one flat buffer
        self.E = CollapsedChain(ops, self.fp, 0, self.dev)  # encoder: d -> 8L -> 4L -> 2L -> L
        self.D = CollapsedChain(ops, self.fp, 4, self.dev)  # decoder: L -> 2L -> 4L -> 8L -> d
        eE, eD = self.E.e[0], self.D.e[0]                   # padded homogeneous input widths: round4(d + 1), round4(L + 1)
        # MMD operand: the L encoding columns only.  (Including the ones column would be exact on paper -- a constant column
        # shifts no distance -- but in fp32 it adds 1 to every squared norm and Gram entry, and L = s_i + s_j - 2 g then
        # cancels catastrophically against distances of ~1e-6.)
        self.p = L
        self.eD = eD

        # generator forward (no gradient)
        self.zh = torch.zeros(n, self.G.e[0], **f32)         # [z | 1 | 0-pad]
        self.zh[:, L] = 1.0
        self.z = self.zh[:, :L]
        self.logits = torch.zeros(n, d, **f32)
        self.S = torch.zeros(n, d, **f32)
        self.U = torch.zeros(n, d, **f32)
        self.perm = torch.zeros(self.nb, n, dtype=torch.int32, device=self.dev)
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.sqxp = torch.zeros(2 * n, **f32)               # row norms of [batch ; U*batch] (a by-product nobody reads here)
        # detector on the 2n stacked rows, homogeneous layouts: XPh = [batch ; U*batch | 1], encH = [enc | 1], dec
        self.XPh = torch.zeros(2 * n, eE, **f32)
        self.XPh[:, d] = 1.0
        self.encH = torch.zeros(2 * n, eD, **f32)
        self.encH[:, L] = 1.0
        self.dec = torch.zeros(2 * n, dp, **f32)            # pad columns stay zero
        self.ddec = torch.zeros(2 * n, dp, **f32)           # gradient of dec (pad columns zero: the vector path of M_4)
        # at the encoder output the decoder path (slab 0) and the split-K slabs of the MMD backward meet
        self.msplits = max(1, min(8, (2 * n) // 256))
        self.denc = torch.zeros(1 + self.msplits, 2 * n, eD, **f32)
        self.mse_part = torch.zeros((2 * n + 3) // 4, dtype=torch.float64, device=self.dev)
        # MMD on the encodings (gradient for all 2n rows)
        # the MMD operand: the L encoding columns copied into a zero-padded [2n, round4(L)] image (norms in the same pass), so
        # that p = L = 49 runs the 16-byte staging path of the Gram / backward kernels
        self.pz = _round4(L)
        self.encZ = torch.zeros(2 * n, self.pz, **f32)
        self.sq = torch.zeros(2 * n, **f32)
        self._fin = None
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
        self.acc_mmd = torch.zeros(1, **f32)                # epoch accumulators (sums over steps)
        self.acc_mse = torch.zeros(1, **f32)                # mse(batch, dec_X) + mse(U*batch, dec_P)

    # ---- host-side controls -------------------------------------------------------------------------
    def set_bandwidth(self, value):
        self.bw.fill_(float(value))
        self.has_bw = True

    def set_epoch_batches(self, idx):
        """idx: [batches_per_epoch, n] integer tensor of shuffled row indices (DataLoader order)."""
        self.perm.copy_(idx.to(dtype=torch.int32).view(self.nb, self.n), non_blocking=True)

    def shuffle_epoch(self, epoch):
        """This epoch's drop_last batches from the counter-based device permutation (vgan_shuffle_epoch): no host draw, no copy."""
        self.ops.shuffle_epoch(self.perm, self.data.shape[0], self.seed, int(epoch))

    def set_noise(self, z):
        """Host-provided noise [n, L] for the next step (parity runs: the reference draws it on the CPU)."""
        self.z.copy_(z.to(dtype=torch.float32), non_blocking=True)

    def epoch_sums(self):
        """(sum of MMD terms, sum of mse_X + mse_P) since the last call -- one host sync."""
        out = (float(self.acc_mmd.item()), float(self.acc_mse.item()))
        self.acc_mmd.zero_()
        self.acc_mse.zero_()
        return out

    # ---- pieces -------------------------------------------------------------------------------------------
    def _feed(self, idx, noise):
        if idx is not None:
            if self.nb != 1:
                raise ValueError("per-step indices need batches_per_epoch == 1 (use set_epoch_batches / shuffle_epoch otherwise)")
            self.set_epoch_batches(idx)
        if noise is not None:
            if self.noise_mode != "host":
                raise ValueError("the engine draws its own noise (noise='device'); build it with noise='host' to provide it")
            self.set_noise(noise)

    def _run(self, key, body, count=1):
        """`count` steps of one kind.  Eager until the bandwidth exists, then one captured graph per step kind -- and, with the
        resident feed and device noise (nothing on the host changes between steps), a second graph holding `steps_per_graph`
        steps, so that an epoch is a handful of graph launches: between two launches the GPU idles ~8 us, a sixth of the
        generator-phase step."""
        m = self.steps_per_graph
        while count > 0:
            if not self.use_graph or not self.has_bw or self.graphs.get(key) is False:
                body()
                count -= 1
                continue
            block = m if (count >= m and m > 1 and self.noise_mode == "device" and key in self.graphs and
                          self.graphs.get((key, m)) is not False) else 1
            gkey = (key, block) if block > 1 else key
            g = self.graphs.get(gkey)
            if g is None:
                g = self.graphs[gkey] = self._capture(body, block)
                if g is False:  # no graph of this size: the one-step graph (block > 1) or eager launches (block == 1) take over
                    continue
            g.replay()
            count -= block

    @staticmethod
    def _capture(body, steps):
        """`steps` step bodies captured into one HIP graph, or False if the capture fails: like NoKLStepEngine._capture_graph,
        a failed capture means "no graph of this size", never "no training" -- the caller falls back in the same process."""
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                for _ in range(steps):
                    body()
        except Exception as e:  # noqa: BLE001
            import warnings
            warnings.warn(f"vgan_amd: HIP-graph capture of {steps} VGAN.fit step(s) failed ({type(e).__name__}: {e}); "
                          f"falling back to {'the one-step graph' if steps > 1 else 'eager launches'}")
            torch.cuda.synchronize()
            return False
        return g  # capture does not execute: the caller replays it

    def _finalize_job(self, tiles):
        return self.ops.finalize_job(self.partial, tiles, self.colpart, self.ops.colmax_chunks(self.n), self.colkey, self.n, self.d,
                                     self.pen, self.stats, self.mmd, self.acc_mmd, 1.0, self.step_counter)

    def _forward(self, want_grad, want_decoder=True):
        """Returns True when the step tail (block sums -> loss) is still to be done by the caller (it rides in the MMD backward
        launch when there is one)."""
        ops, n, d, L = self.ops, self.n, self.d, self.L
        if self.noise_mode == "device":
            ops.noise_normal(self.zh, self.seed, self.step_counter, 0, cols=L, ones_col=L)
        self.G.forward(self.zh, self.logits)
        ops.mask_project_forward(self.logits, self.data, self.perm, self.S, self.U, self.XPh[:n], self.XPh[n:], self.sqxp[:n],
                                 self.sqxp[n:], row_cursor=self.step_counter, row_batches=self.nb, row_stride=n)
        self.E.forward(self.XPh, self.encH[:, :L])
        if want_decoder:
            self.D.forward(self.encH, self.dec[:, :d])
        Z, p = self.encZ, self.pz
        ops.gather_rows_split(self.encH[:, :L], None, None, Z, self.sq, n=2 * n)  # copy + norms; pad columns of Z stay zero
        if not self.has_bw:  # first call of the (process-wide) RBF calibrates its bandwidth (Mmd_loss_constrained.py:16-20)
            ops.mmd_gram(Z, self.sq, n, p, None, self.tiles0, True, None, 0, self.partial)
            ops.mmd_reduce(self.partial, self.tiles0, self.stats, True)
            ops.mmd_set_bandwidth(self.stats, n, self.bw)
            self.has_bw = True
        tiles = self.tiles if want_grad else self.tiles0
        # Gram + the column arg-max cells of topk(U, 1, 0) in one launch; the tail (block sums, arg-max keys, loss) in one more,
        # or inside the MMD backward launch when the encoder trains
        ops.mmd_gram_colmax(Z, self.sq, n, p, self.bw, tiles, self.Wg if want_grad else None, 0, self.partial, self.S, 0, self.colpart, True)
        if want_grad:
            return True
        ops.mmd_finalize(self.partial, tiles, self.colpart, ops.colmax_chunks(n), self.colkey, n, d, self.pen, self.stats, self.mmd,
                         self.acc_mmd, 1.0, self.step_counter)
        return False

    def generator_phase_step(self, idx=None, noise=None, count=1):
        """Loss evaluation of the generator phase: accumulates MMD(enc_X, enc_P, U) (src/vgan.py:295-329).  count > 1: that many
        consecutive steps of the epoch table (resident feed only)."""
        self._feed(idx, noise)
        self._run("g", lambda: self._forward(want_grad=False, want_decoder=False), self._count(count, idx, noise))  # loss_G needs no decoder pass

    def detector_step(self, idx=None, noise=None, train_encoder=True, count=1):
        self._feed(idx, noise)
        self._run(("d", bool(train_encoder)), lambda: self._detector_body(bool(train_encoder)), self._count(count, idx, noise))

    def _count(self, count, idx, noise):
        if count != 1 and (idx is not None or noise is not None or self.noise_mode != "device"):
            raise ValueError("several steps per call need the resident feed and the device noise stream: no per-step indices or noise")
        return int(count)

    def _detector_body(self, train_encoder):
        ops, n, d, L = self.ops, self.n, self.d, self.L
        # the MMD term reaches only the encoder: with the encoder frozen neither its gradient weights nor its backward run
        tail_pending = self._forward(want_grad=train_encoder)
        # gradients of G = MMD - 0.1 mse_X - 0.1 mse_P  (= -loss_D)
        gs = -0.1 * 2.0 / (float(n) * d)
        # both squared-error terms have the same weight and the same 1/(n d): one pass over the 2n stacked rows, one fold
        ops.mse_grad(self.XPh[:, :d], self.dec[:, :d], gs, self.mse_part, self.ddec[:, :d])
        ops.sum_f64(self.mse_part, self.mse_part.numel(), 1.0 / (float(n) * d), self.acc_mse, accumulate=True)
        adadelta = dict(lr=self.lr, rho=ADADELTA_RHO, eps=ADADELTA_EPS, weight_decay=self.wd, grad_scale=-1.0)
        if train_encoder:  # both read the CURRENT decoder products: before the decoder update
            self.D.input_grad(self.ddec, self.denc[0][:, :L])  # columns >= L of every slab stay zero
            if self._fin is None:
                self._fin = self._finalize_job(self.tiles)
            ops.mmd_backward(self.Wg, self.encZ, 0, 2 * n, 2 * n, self.pz, None, self.denc[1], self.msplits, 2 * n * self.eD,
                             self._fin if tail_pending else None)
        self.D.backward(self.ddec, self.encH)
        self.D.update(**adadelta)
        if train_encoder:
            ops.reduce_slabs(self.denc, 2 * n * self.eD, 1 + self.msplits, self.denc[0].view(-1))  # d enc = decoder path + MMD slabs
            self.E.backward(self.denc[0][:, :_round4(L)], self.XPh)  # zero pad columns: the vector path of M_4
            self.E.update(**adadelta)
