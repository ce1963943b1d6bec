"""Step engine of ``VGAN_no_kl.fit`` (reference: src/vgan.py:597-621) over a kernel provider.

One step = noise -> Generator_big (4 Linear) -> upper_softmax -> U*X -> MMDLossConstrained ->
backward -> Adadelta, launched as ~20 asynchronous kernels on caller-owned, preallocated HBM
buffers and (on a GPU) replayed from one captured HIP graph per step.  Nothing in a step
synchronises the host: the loss is accumulated on the device and read once per epoch.

HBM layout (all float32 unless noted; n = global batch, nl = rows owned by this rank, d features,
dp = d rounded up to 4, L = latent size):
  flat params / grads / Adadelta state   one contiguous buffer each, parameter k at a 16-byte
                                          aligned offset, weights dense [out, in]
  data [Ntrain, d]                        whole training set, resident; batches are gathered by
                                          index inside the mask/projection kernel
  perm [batches_per_epoch, n] int32       this epoch's shuffled indices
  Z [2n, dp]  sq [2n]                     MMD operand [X_batch ; U*X_batch] and its row norms
  Wg [nl, 2n]                             gradient weights of this rank's Y rows (never the 5x2nx2n K)
  acts: z [n, L], h1, h2, h3, logits [nl, *];  S [nl, d];  gU [nl, dp];  dlogits, dh3, dh2, dh1

Data parallel (exact): rank r owns rows [r*n/G, (r+1)*n/G) of the batch.  Exchange per
step: all-gather of the Y rows (+ their norms), all-reduce(MAX) of the packed column arg-max keys,
all-reduce(SUM) of the four block statistics, all-reduce(SUM) of the flat gradient.  Every rank
applies the identical Adadelta update.  The data set and the noise stream are replicated, so
results do not depend on the number of ranks.
"""
import torch

ADADELTA_RHO = 0.9   # torch.optim.Adadelta defaults used by the reference (src/vgan.py:567-568)
ADADELTA_EPS = 1e-6


def _round4(v):
    return (v + 3) // 4 * 4


class FlatParams:
    """Re-homes a module's parameters into one flat buffer (16-byte aligned offsets) so that Adadelta
    and the gradient all-reduce are single streaming passes.  ``module`` keeps working: each
    parameter's ``.data`` becomes a view of the flat buffer."""

    def __init__(self, params, device):
        self.shapes = [tuple(p.shape) for p in params]
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += _round4(p.numel())
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.flat)
        self.sq = torch.zeros_like(self.flat)
        self.acc = torch.zeros_like(self.flat)
        for k, (p, o) in enumerate(zip(params, self.offsets)):
            self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
            p.data = self.view(self.flat, k)

    def view(self, buf, k):
        o, shp = self.offsets[k], self.shapes[k]
        numel = 1
        for s in shp:
            numel *= s
        return buf[o:o + numel].view(shp)


class NoKLStepEngine:
    def __init__(self, ops, generator, data, batch_size, batches_per_epoch, lr=0.007, weight_decay=0.04, penalty_weight=10.0,
                 seed=777, noise="device", rank=0, world=1, group=None, use_graph=True, loss_accum_scale=None):
        self.ops = ops
        self.dev = data.device
        self.rank, self.world, self.group = rank, world, group
        n = self.n = int(batch_size)
        if n % world != 0:
            raise ValueError(f"global batch {n} must be divisible by the number of ranks {world}")
        self.nl = n // world
        self.lo = rank * self.nl
        self.data = data
        self.d = d = data.shape[1]
        self.dp = _round4(d)
        self.nb = int(batches_per_epoch)
        self.lr, self.wd, self.pen = float(lr), float(weight_decay), float(penalty_weight)
        self.seed = int(seed)
        self.noise_mode = noise
        self.use_graph = bool(use_graph) and data.is_cuda
        self.graph = None
        self.steps_done = 0

        lin = [m for m in generator.main if isinstance(m, torch.nn.Linear)]
        assert len(lin) == 4
        self.L = lin[0].in_features
        params = [q for m in lin for q in (m.weight, m.bias)]
        self.fp = FlatParams(params, self.dev)
        self.W = [self.fp.view(self.fp.flat, 2 * k) for k in range(4)]
        self.b = [self.fp.view(self.fp.flat, 2 * k + 1) for k in range(4)]
        # weight gradients contract over the batch rows: split that contraction into slabs so the launch fills
        # the chip, then sum the slabs (fixed order) into the flat gradient
        self.splits = max(1, min(8, (n // world) // 128))
        self.gslab = torch.zeros(self.splits, self.fp.total, dtype=torch.float32, device=self.dev) if self.splits > 1 else None
        gbase = self.gslab[0] if self.splits > 1 else self.fp.grad
        self.dW = [self.fp.view(gbase, 2 * k) for k in range(4)]
        self.db = [self.fp.view(gbase, 2 * k + 1) for k in range(4)]

        f32 = dict(dtype=torch.float32, device=self.dev)
        nl, dp = self.nl, self.dp
        self.z_full = torch.zeros(n, self.L, **f32)
        widths = [self.L] + [m.out_features for m in lin]
        self.acts = [self.z_full[self.lo:self.lo + nl]] + [torch.zeros(nl, w, **f32) for w in widths[1:]]
        self.dacts = [None] + [torch.zeros(nl, w, **f32) for w in widths[1:4]]
        self.S = torch.zeros(nl, d, **f32)
        self.Z = torch.zeros(2 * n, dp, **f32)
        self.sqn = torch.zeros(2 * n, **f32)
        self.Wg = torch.zeros(nl, 2 * n, **f32)
        # the backward GEMM contracts over the 2n rows of Z: sliced so that every SIMD holds several waves; the
        # partial slabs are summed by the mask-backward kernel
        import os
        self.bsplits = max(1, int(os.environ.get("VGAN_BWD_SPLITS", "2")))
        self.gU_slabs = torch.zeros(self.bsplits, nl, dp, **f32)
        self.gU = self.gU_slabs[0]
        self.dlogits = torch.zeros(nl, d, **f32)
        self.perm = torch.zeros(self.nb, n, dtype=torch.int32, device=self.dev)
        self.tiles = ops.build_tiles(n, 1, rank, world, device=self.dev)
        self.partial = torch.zeros(self.tiles.shape[0], 4, **f32)
        # The XX block only feeds the loss value, never the gradient: on one GPU it is launched from its own tile
        # table on a side stream so that it fills the CUs the gradient path (XY/YY tiles -> backward GEMM) leaves idle.
        # Measured on MI355X / ROCm 7.2 (c3): the fork/join graph is SLOWER than the plain chain (357 vs 313 us/step) --
        # every cross-stream edge of a replayed HIP graph costs more than the overlap returns -- so it is opt-in.
        import os
        self.concurrent = bool(data.is_cuda and world == 1 and os.environ.get("VGAN_CONCURRENT", "0") not in ("0", ""))
        self.concurrent_dw = os.environ.get("VGAN_CONCURRENT", "0") in ("1", "dw")
        self.concurrent_xx = os.environ.get("VGAN_CONCURRENT", "0") in ("1", "xx")
        if self.concurrent:
            slot = (self.tiles[:, 4] & 3)
            self.tiles_xx = self.tiles[slot == 0].contiguous()
            self.tiles_g = self.tiles[slot != 0].contiguous()
            self.partial_xx = torch.zeros(self.tiles_xx.shape[0], 4, **f32)
            self.partial_g = torch.zeros(self.tiles_g.shape[0], 4, **f32)
            self.side = [torch.cuda.Stream(device=self.dev) for _ in range(2)]
        self.stats = torch.zeros(4, dtype=torch.float64, device=self.dev)
        self.bw = torch.zeros(1, **f32)
        self.has_bw = False
        self.loss = torch.zeros(1, **f32)
        self.loss_accum = torch.zeros(1, **f32)
        self.accum_scale = (1.0 / self.nb) if loss_accum_scale is None else float(loss_accum_scale)
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.colpart = torch.zeros(ops.colmax_chunks(nl) * d, dtype=torch.int64, device=self.dev)
        self.colkey = torch.zeros(d, dtype=torch.int64, device=self.dev)

    # ---- host-side controls ---------------------------------------------------------------------
    def set_epoch_batches(self, idx):
        """idx: [batches_per_epoch, n] integer tensor of shuffled row indices (DataLoader order)."""
        self.perm.copy_(idx.to(dtype=torch.int32), non_blocking=True)

    def set_bandwidth(self, value):
        self.bw.fill_(float(value))
        self.has_bw = True

    def set_noise(self, z):
        """Host-provided noise for the next step (parity runs: the reference draws it on the CPU)."""
        self.z_full.copy_(z.to(dtype=torch.float32), non_blocking=True)

    def epoch_loss(self):
        """Mean loss of the steps since the last call (one host sync), as the reference's
        ``generator_loss += loss / batch_number`` (src/vgan.py:620-621)."""
        v = float(self.loss_accum.item())
        self.loss_accum.zero_()
        return v

    # ---- the step -----------------------------------------------------------------------------------
    def _collect(self):
        import torch.distributed as dist
        return dist

    def _forward(self):
        ops, n, nl, lo = self.ops, self.n, self.nl, self.lo
        if self.noise_mode == "device":
            ops.noise_normal(self.z_full, self.seed, self.step_counter, 0)
        for k in range(4):
            ops.linear_forward(self.acts[k], self.W[k], self.b[k], self.acts[k + 1])
        rowsel = dict(row_cursor=self.step_counter, row_batches=self.nb, row_stride=n)
        if self.world == 1:
            ops.mask_project_forward(self.acts[4], self.data, self.perm, self.S, None, self.Z[:n], self.Z[n:], self.sqn[:n],
                                     self.sqn[n:], row_offset=0, **rowsel)
        else:
            dist = self._collect()
            ops.gather_rows(self.data, self.perm, self.Z[:n], self.sqn[:n], row_offset=0, **rowsel)
            ops.mask_project_forward(self.acts[4], self.data, self.perm, self.S, None, None, self.Z[n + lo:n + lo + nl], None,
                                     self.sqn[n + lo:n + lo + nl], row_offset=lo, **rowsel)
            dist.all_gather_into_tensor(self.Z[n:], self.Z[n + lo:n + lo + nl], group=self.group)
            dist.all_gather_into_tensor(self.sqn[n:], self.sqn[n + lo:n + lo + nl], group=self.group)

    def _calibrate(self):
        """First-call bandwidth (src/models/Mmd_loss_constrained.py:16-20): sum(L) / (N^2 - N)."""
        ops = self.ops
        ops.mmd_gram(self.Z, self.sqn, self.n, self.dp, None, self.tiles, True, None, 0, self.partial)
        ops.mmd_reduce(self.partial, self.tiles, self.stats, True)
        if self.world > 1:
            dist = self._collect()
            dist.all_reduce(self.stats, group=self.group)
        ops.mmd_set_bandwidth(self.stats, self.n, self.bw)
        self.has_bw = True

    def _loss_backward_update(self):
        ops, n, nl, lo, d = self.ops, self.n, self.nl, self.lo, self.d
        dist = self._collect() if self.world > 1 else None
        gstride = nl * self.dp
        if dist is None:
            ops.colmax_partial(self.S, lo, self.colpart, True)
            ops.mmd_gram(self.Z, self.sqn, n, self.dp, self.bw, self.tiles, False, self.Wg, n + lo, self.partial)
            ops.mmd_finalize(self.partial, self.tiles, self.colpart, ops.colmax_chunks(nl), self.colkey, n, d, self.pen, self.stats,
                             self.loss, self.loss_accum, self.accum_scale, self.step_counter)
        else:
            ops.colmax(self.S, lo, self.colpart, self.colkey, True)
            dist.all_reduce(self.colkey, op=dist.ReduceOp.MAX, group=self.group)
            ops.mmd_gram(self.Z, self.sqn, n, self.dp, self.bw, self.tiles, False, self.Wg, n + lo, self.partial)
            ops.mmd_reduce(self.partial, self.tiles, self.stats, True)
            dist.all_reduce(self.stats, group=self.group)
            ops.mmd_loss(self.stats, self.colkey, n, d, self.pen, self.loss, self.loss_accum, self.accum_scale, self.step_counter)
        ops.mmd_backward(self.Wg, self.Z, n + lo, nl, 2 * n, self.dp, self.Z[lo:lo + nl], self.gU, self.bsplits, gstride)
        ops.mask_backward(self.gU, self.S, self.colkey, self.pen, lo, self.dlogits, self.bsplits, gstride)
        g = self.dlogits
        for k in (3, 2, 1, 0):
            ops.linear_backward_params(g, self.acts[k], self.dW[k], self.db[k], self.splits, self.fp.total)
            if k:
                ops.linear_backward_input(g, self.W[k], self.dacts[k])
                g = self.dacts[k]
        if dist:
            if self.splits > 1:
                ops.reduce_slabs(self.gslab, self.fp.total, self.splits, self.fp.grad)
            dist.all_reduce(self.fp.grad, group=self.group)
            ops.adadelta_step(self.fp.flat, self.fp.grad, self.fp.sq, self.fp.acc, self.lr, ADADELTA_RHO, ADADELTA_EPS, self.wd, 1.0)
        elif self.splits > 1:  # single rank: Adadelta sums the slabs itself
            ops.adadelta_step(self.fp.flat, self.gslab[0], self.fp.sq, self.fp.acc, self.lr, ADADELTA_RHO, ADADELTA_EPS, self.wd, 1.0,
                              self.splits, self.fp.total)
        else:
            ops.adadelta_step(self.fp.flat, self.fp.grad, self.fp.sq, self.fp.acc, self.lr, ADADELTA_RHO, ADADELTA_EPS, self.wd, 1.0)

    def grad_view(self, k):
        """Gradient of parameter tensor k as of the last step (sums the split-K slabs if they were not reduced)."""
        if self.world == 1 and self.splits > 1:
            return sum(self.fp.view(self.gslab[sl], k) for sl in range(self.splits))
        return self.fp.view(self.fp.grad, k)

    def _loss_backward_update_concurrent(self):
        """Single-GPU step tail as a fork/join graph over three streams (captured into the same HIP graph):
             main : gram(XY,YY) -> backward GEMM -> mask backward -> dh chain -> slab reduce -> Adadelta
             side0: column arg-max, gram(XX) -> block sums -> loss           (joins at the end of the step)
             side1: the four weight-gradient GEMMs, each as soon as its dy exists
        Independent kernels overlap, and the short or under-filled ones (XX tiles, dW GEMMs) run in the holes of
        the long ones (528 Gram tiles / 208 backward tiles do not divide 256 CUs)."""
        ops, n, d = self.ops, self.n, self.d
        main = torch.cuda.current_stream()
        s0, s1 = self.side
        if self.concurrent_xx:
            s0.wait_stream(main)
            with torch.cuda.stream(s0):
                ops.colmax(self.S, 0, self.colpart, self.colkey, True)
                ev_colmax = torch.cuda.Event()
                ev_colmax.record(s0)
                ops.mmd_gram(self.Z, self.sqn, n, self.dp, self.bw, self.tiles_xx, False, None, 0, self.partial_xx)
            ops.mmd_gram(self.Z, self.sqn, n, self.dp, self.bw, self.tiles_g, False, self.Wg, n, self.partial_g)
            ev_gram = torch.cuda.Event()
            ev_gram.record(main)
            with torch.cuda.stream(s0):
                s0.wait_event(ev_gram)
                ops.mmd_reduce(self.partial_xx, self.tiles_xx, self.stats, True)
                ops.mmd_reduce(self.partial_g, self.tiles_g, self.stats, False)
                ops.mmd_loss(self.stats, self.colkey, n, d, self.pen, self.loss, self.loss_accum, self.accum_scale, self.step_counter)
            ops.mmd_backward(self.Wg, self.Z, n, n, 2 * n, self.dp, self.Z[:n], self.gU, self.bsplits, n * self.dp)
            main.wait_event(ev_colmax)
        else:
            ops.colmax(self.S, 0, self.colpart, self.colkey, True)
            ops.mmd_gram(self.Z, self.sqn, n, self.dp, self.bw, self.tiles, False, self.Wg, n, self.partial)
            ops.mmd_reduce(self.partial, self.tiles, self.stats, True)
            ops.mmd_loss(self.stats, self.colkey, n, d, self.pen, self.loss, self.loss_accum, self.accum_scale, self.step_counter)
            ops.mmd_backward(self.Wg, self.Z, n, n, 2 * n, self.dp, self.Z[:n], self.gU, self.bsplits, n * self.dp)
        ops.mask_backward(self.gU, self.S, self.colkey, self.pen, 0, self.dlogits, self.bsplits, n * self.dp)
        g = self.dlogits
        for k in (3, 2, 1, 0):
            if self.concurrent_dw:
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(s1):
                    s1.wait_event(ev)
                    ops.linear_backward_params(g, self.acts[k], self.dW[k], self.db[k], self.splits, self.fp.total)
            else:
                ops.linear_backward_params(g, self.acts[k], self.dW[k], self.db[k], self.splits, self.fp.total)
            if k:
                ops.linear_backward_input(g, self.W[k], self.dacts[k])
                g = self.dacts[k]
        if self.concurrent_dw:
            main.wait_stream(s1)
        if self.splits > 1:
            ops.reduce_slabs(self.gslab, self.fp.total, self.splits, self.fp.grad)
        ops.adadelta_step(self.fp.flat, self.fp.grad, self.fp.sq, self.fp.acc, self.lr, ADADELTA_RHO, ADADELTA_EPS, self.wd, 1.0)
        if self.concurrent_xx:
            main.wait_stream(s0)

    def _step_body(self):
        self._forward()
        if self.concurrent:
            self._loss_backward_update_concurrent()
        else:
            self._loss_backward_update()

    def step(self):
        """Runs one training step asynchronously.  The first step also calibrates the bandwidth."""
        if not self.has_bw:
            self._forward()
            self._calibrate()
            if self.concurrent:
                self._loss_backward_update_concurrent()
            else:
                self._loss_backward_update()
        elif self.use_graph:
            if self.graph is None:
                self._capture()
            self.graph.replay()
        else:
            self._step_body()
        self.steps_done += 1

    def _capture(self):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_body()
        # capture does not execute: the step that triggered it still has to run once
        self.graph = g

    # ---- sampling (generate_subspaces) ---------------------------------------------------------
    def generator_logits(self, z):
        """Generator forward on caller noise [m, L] -> logits [m, d] (fresh buffers, eager)."""
        ops = self.ops
        h = z.to(device=self.dev, dtype=torch.float32).contiguous()
        for k in range(4):
            y = torch.empty(h.shape[0], self.W[k].shape[0], dtype=torch.float32, device=self.dev)
            ops.linear_forward(h, self.W[k], self.b[k], y)
            h = y
        return h
