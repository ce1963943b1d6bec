"""User surface: ``VGAN_no_kl`` and ``VGAN`` with the reference's constructor arguments, attributes
and methods (reference: src/vgan.py:20-431 and :434-708), training on the MI355X kernels.

``from src.vgan import VGAN, VGAN_no_kl`` (the notebook's import, test.ipynb:16) resolves to these
classes through the ``src`` package at the repository root.
"""
import os
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader

from .modules import Decoder, Detector, Encoder, Generator_big, MMDLossConstrained
from .ops import default_ops
from .kl_trainer import KLStepEngine
from .trainer import NoKLStepEngine


def _device():
    # src/vgan.py:46-47 picks cuda:0 -> mps:0 -> cpu; this build only runs on a HIP device
    return torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


def _epoch_batches_dataloader(train_size, batch_size):
    loader = DataLoader(torch.arange(train_size), batch_size=batch_size, drop_last=True, shuffle=True)
    return torch.stack(list(loader))


def _epoch_batches_direct(train_size, batch_size):
    # what iterating that DataLoader draws from the default generator (torch/utils/data/dataloader.py
    # `_BaseDataLoaderIter.__init__` -> base seed; sampler.py `RandomSampler.__iter__` -> seed of a private
    # generator for randperm), without the per-batch collate overhead
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    nb = train_size // batch_size
    return torch.randperm(train_size, generator=g)[:nb * batch_size].view(nb, batch_size)


_direct_ok = None


def epoch_batches(train_size, batch_size):
    """Shuffled ``drop_last`` index batches of one epoch [batches, batch_size], drawn exactly as the
    reference's ``DataLoader(X, batch_size, shuffle=True, drop_last=True)`` draws them
    (src/vgan.py:578-584), i.e. consuming torch's default CPU generator in the same order.  The
    direct path is verified once per process against a real DataLoader from a forked RNG state; if
    this torch version draws differently, the DataLoader itself is used."""
    global _direct_ok
    if _direct_ok is None:
        with torch.random.fork_rng(devices=[]):
            state = torch.random.get_rng_state()
            a = _epoch_batches_dataloader(64, 8)
            after_a = torch.random.get_rng_state()
            torch.random.set_rng_state(state)
            b = _epoch_batches_direct(64, 8)
            _direct_ok = bool(torch.equal(a, b) and torch.equal(after_a, torch.random.get_rng_state()))
    fn = _epoch_batches_direct if _direct_ok else _epoch_batches_dataloader
    return fn(train_size, batch_size)


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class _RunFolder:
    """Side outputs and sampling helpers shared by both model classes: the run folder a fit leaves behind
    (<dir>/models/generator_<k>.pt, <dir>/train_history/generator_loss_<k>.csv, <dir>/params.csv, <dir>/train_history.pdf --
    the layout of src/vgan.py:111-140, 339-350, 626-635, pinned by fixture f6), load_models, generate_subspaces and
    approx_subspace_dist.  Host code except for the generator forward and the unique-count, which run on the GPU."""

    def _write_loss_plot(self, folder):
        """train_history.pdf: one curve per recorded loss history (generator; detector too for VGAN)."""
        try:
            import matplotlib
            matplotlib.use("Agg")
            from matplotlib import pyplot
        except Exception:  # a missing plotting stack must not fail a training run
            return
        curves = [("generator_loss", "Generator loss", "cornflowerblue")]
        if hasattr(self, "lr_D"):
            curves.append(("detector_loss", "Detector loss", "black"))
        figure, axes = pyplot.subplots()
        for key, label, colour in curves:
            values = list(self.train_history[key])
            axes.plot(range(1, len(values) + 1), values, label=label, color=colour, linewidth=2)
        axes.set_xlabel("Epoch")
        axes.set_ylabel("Loss")
        axes.legend(loc="upper right")
        figure.savefig(Path(folder) / "train_history.pdf", format="pdf", dpi=1200)
        pyplot.close(figure)

    def model_snapshot(self, path_to_directory=None, run_number=0, show=False):
        """Writes this run's loss history and hyper-parameters into the run folder (``show`` is accepted and ignored, as in
        the reference, which deprecated it)."""
        import pandas as pd
        folder = Path(self.path_to_directory if path_to_directory is None else path_to_directory)
        (folder / "train_history").mkdir(parents=True, exist_ok=True)
        history = pd.Series(list(self.train_history["generator_loss"]))
        history.to_csv(folder / "train_history" / f"generator_loss_{run_number}.csv", header=False, index=False)
        # params.csv: one row per run number; an existing file keeps its other rows, this run's row is added or replaced
        table_path = folder / "params.csv"
        # (reference quirk kept, src/vgan.py:130-132: a params.csv that does not exist yet gets its first row under index 0
        #  whatever run_number is)
        fresh = not table_path.is_file()
        table = pd.DataFrame() if fresh else pd.read_csv(table_path, index_col=0)
        row = pd.DataFrame.from_dict({0 if fresh else run_number: self.get_params()}, orient="index")
        table = row.combine_first(table)[list(row.columns) + [c for c in table.columns if c not in row.columns]]
        table.sort_index().to_csv(table_path)
        self._write_loss_plot(folder)

    def load_models(self, path_to_generator, ndims, device=None):
        """src/vgan.py:142-158 / 511-527: restore a generator for sampling.  The file is a plain state_dict (keys main.N.weight /
        main.N.bias, as the reference writes it) and is read with ``weights_only=True``."""
        device = self.device if device is None else device
        latent = max(int(ndims / 16), 1)
        generator = Generator_big(img_size=ndims, latent_size=latent).to(device)
        generator.load_state_dict(torch.load(path_to_generator, map_location=device, weights_only=True))
        generator.eval()
        self.generator = generator
        self._latent_size = latent
        self.generator_optimizer = f"Loaded Model from {path_to_generator} with {ndims} dimensions in the latent space"

    def _ops(self):
        return getattr(self, "_ops_override", None) or default_ops()

    def _generator_forward(self, z):
        """Generator_big forward (4 Linear + upper_softmax) straight on the kernel provider, no autograd."""
        ops = self._ops()
        h = z.to(device=self.device, dtype=torch.float32).contiguous()
        for m in self.generator.main:
            if isinstance(m, torch.nn.Linear):
                y = torch.empty(h.shape[0], m.out_features, dtype=torch.float32, device=h.device)
                ops.linear_forward(h, m.weight.detach(), m.bias.detach(), y)
                h = y
        S, U = torch.empty_like(h), torch.empty_like(h)
        ops.upper_softmax_forward(h, S, U)
        return U

    def generate_subspaces(self, nsubs):
        """src/vgan.py:355-370 / 639-647: seeded CPU noise -> generator -> bool mask ``u >= 1/d``."""
        noise_tensor = torch.Tensor(nsubs, self._latent_size).to("cpu")
        if self.seed is not None:
            torch.manual_seed(self.seed)
        noise_tensor.normal_()
        u = self._generator_forward(noise_tensor)
        return torch.greater_equal(u, 1 / u.shape[1])

    def sample(self, nsubs):
        """Alias of generate_subspaces (BASELINE.json's north_star calls it sample())."""
        return self.generate_subspaces(nsubs)

    def approx_subspace_dist(self, subspace_count=500, add_leftover_features=False):
        """src/vgan.py:372-382 / 649-659: the distinct sampled subspaces (rows in np.unique's order) and their empirical
        probabilities.  The sort-free unique-count runs on the GPU (vgan_mask_unique); only the distinct rows come back."""
        masks = self.generate_subspaces(subspace_count)
        distinct, counts = self._ops().mask_unique(masks)
        distinct, weights = distinct.to("cpu").numpy(), counts.to("cpu").numpy().astype(np.float64)
        never_selected = ~distinct.any(axis=0)
        if add_leftover_features and never_selected.any():  # one more "subspace" holding every feature no sample selected
            distinct = np.vstack([distinct, never_selected[None, :]])
            weights = np.append(weights / weights.sum(), 1.0)
        self.subspaces = distinct
        self.proba = weights / weights.sum()

    def check_if_myopic(self, x_data, bandwidth=0.01, count=500, n_permutations=1000):
        """src/vgan.py:384-431: two-sample (MMD, permutation) test of P(x) against P(u * x + mean(x) * ~u), once per given
        bandwidth and once with the recommended one; returns a DataFrame of p-values.

        The reference delegates the statistic and the permutation p-value to torch-two-sample, which is neither vendored
        nor pinned; the algorithm used here is that package's published one as restated in
        oracle/vgan_oracle.py -- parity unpinned.  The kernel matrix, the P x m by m x m product and the per-permutation
        dot products run on the GPU; random choices (row sample, permutations) come from numpy's Generator seeded with
        ``self.seed`` instead of numpy's global state."""
        import pandas as pd
        from .modules import MMDLossConstrained
        x_data = np.asarray(x_data, dtype=np.float64)
        assert count <= x_data.shape[0], "Selected 'count' is greater than the number of samples in the dataset"
        ops, dev = self._ops(), self.device
        rng = np.random.default_rng(self.seed)
        norms = np.sqrt((x_data * x_data).sum(axis=0))          # sklearn.preprocessing.normalize(x, axis=0), L2
        x_data = x_data / np.where(norms == 0.0, 1.0, norms)
        rows = rng.choice(x_data.shape[0], size=count, replace=False)
        x_sample = torch.as_tensor(x_data[rows], dtype=torch.float32).to(dev)
        u = self.generate_subspaces(count)
        ux_sample = torch.where(u, x_sample, x_sample.mean(dim=0, keepdim=True).expand_as(x_sample)).contiguous()
        if isinstance(bandwidth, float):
            bandwidth = [bandwidth]
        bandwidth = sorted(bandwidth)
        if not hasattr(self, "bandwidth") or self.bandwidth is None:
            mmd_loss = MMDLossConstrained(0)
            mmd_loss.forward(x_sample, ux_sample, u * 1.0)
            self.bandwidth = mmd_loss.bandwidth
        rec = float(self.bandwidth.item()) if torch.is_tensor(self.bandwidth) else float(self.bandwidth)

        m, P = 2 * count, int(n_permutations)
        dp = (x_sample.shape[1] + 3) // 4 * 4
        Z = torch.zeros(m, dp, dtype=torch.float32, device=dev)
        Z[:count, :x_sample.shape[1]] = x_sample
        Z[count:, :x_sample.shape[1]] = ux_sample
        sq = torch.empty(m, dtype=torch.float32, device=dev)
        ops.row_sqnorm(Z, sq, dp)
        mp = (m + 3) // 4 * 4
        # assignment rows: 0 = observed (first `count` in group 1), 1..P = shuffles, P+1 = all ones (gives r = 1'K)
        Ut = np.zeros((P + 2, mp), dtype=np.float32)
        Ut[0, :count] = 1.0
        for q in range(1, P + 1):
            Ut[q, rng.permutation(m)[:count]] = 1.0
        Ut[P + 1, :m] = 1.0
        Ut = torch.as_tensor(Ut).to(dev)
        K = torch.zeros(mp, mp, dtype=torch.float32, device=dev)
        T = torch.empty(P + 2, mp, dtype=torch.float32, device=dev)
        a = torch.empty(P + 2, dtype=torch.float64, device=dev)
        b = torch.empty(P + 2, dtype=torch.float64, device=dev)
        n1 = n2 = float(count)
        a00, a11, a01 = 1.0 / (n1 * (n1 - 1)), 1.0 / (n2 * (n2 - 1)), -1.0 / (n1 * n2)
        results = []
        for alpha in list(bandwidth) + [rec]:
            ops.rbf_kernel_matrix(Z, sq, alpha, K[:m])          # rows/cols >= m stay zero
            ops.gemm_grouped([("NN", Ut, K, T)])                # T = Ut . K
            ops.rows_dot(Ut, T, a)                              # u'Ku
            ops.rows_dot(Ut, T[P + 1:P + 2], b, broadcast_b=True)  # u'K1
            ah, bh = a.cpu().numpy(), b.cpu().numpy()
            total = ah[P + 1]
            uKv = bh - ah
            vKv = total - 2.0 * bh + ah
            stat = a00 * (ah + n1) + a11 * (vKv + n2) + 2.0 * a01 * uKv   # K_ii = 1
            results.append(float((stat[0] <= stat[1:P + 1]).sum()) / P)
        return pd.DataFrame([results], columns=list(bandwidth) + ["recommended bandwidth"], index=["p-val"])

    def _save_run(self, generator, detector_too):
        path_to_directory = Path(self.path_to_directory)
        os.makedirs(path_to_directory / "models", exist_ok=True)
        files = len(os.listdir(path_to_directory / "models"))
        run_number = int(files / 2) if detector_too else int(files)
        torch.save(generator.state_dict(), path_to_directory / "models" / f"generator_{run_number}.pt")
        if detector_too:  # the reference writes the GENERATOR's state under the detector's name (src/vgan.py:348-349)
            torch.save(generator.state_dict(), path_to_directory / "models" / f"detector_{run_number}.pt")
        self.model_snapshot(path_to_directory, run_number, show=True)


class VGAN_no_kl(_RunFolder):
    """V-GAN without kernel learning (reference: src/vgan.py:434-708).  Same constructor defaults."""

    def __init__(self, batch_size=500, epochs=2000, lr=0.007, momentum=0.99, seed=777, weight_decay=0.04, path_to_directory=None):
        self.storage = locals()
        self.train_history = defaultdict(list)
        self.batch_size = batch_size
        self.epochs = epochs
        self.lr = lr
        self.momentum = momentum  # stored, never used -- as in the reference (src/vgan.py:448)
        self.seed = seed
        self.weight_decay = weight_decay
        self.path_to_directory = path_to_directory
        self.generator_optimizer = None
        self.device = _device()
        # build-specific knobs (not constructor arguments, so the reference signature is unchanged)
        self.noise_source = "device"   # "device": Philox on the GPU; "host": torch CPU generator (reference CPU-path RNG order)
        self.shuffle_source = "host"   # "host": the DataLoader's own draws (reference RNG order); "device": vgan_shuffle_epoch
        self.use_graph = True
        self.mmd_precision = None      # None: VGAN_MMD_PRECISION or "auto" (see NoKLStepEngine); "fp32" | "bf16x3"
        self.verbose = True

    def get_params(self):
        return {"batch size": self.batch_size, "epochs": self.epochs, "lr_g": self.lr, "momentum": self.momentum,
                "weight decay": self.weight_decay, "batch_size": self.batch_size, "seed": self.seed,
                "generator optimizer": self.generator_optimizer}

    def get_the_networks(self, ndims, latent_size, device=None):
        if device is None:
            device = self.device
        return Generator_big(img_size=ndims, latent_size=latent_size).to(device)

    def _make_engine(self, generator, data, batches_per_epoch, loss_function):
        rank, world = _dist_info()
        eng = NoKLStepEngine(self._ops(), generator, data, self.batch_size, batches_per_epoch, lr=self.lr,
                             weight_decay=self.weight_decay, penalty_weight=loss_function.weight, seed=self.seed or 0,
                             noise=self.noise_source, rank=rank, world=world, use_graph=self.use_graph,
                             mmd_precision=self.mmd_precision)
        shared = loss_function.kernel.bandwidth  # the process-wide RBF may already be calibrated (reference quirk)
        if shared is not None:
            eng.set_bandwidth(float(shared))
        return eng

    def fit(self, X):
        """src/vgan.py:546-637.  X: [Ntrain, d] array-like.  Returns None; sets generator, bandwidth,
        train_history['generator_loss'] (epoch means), batch_size = min(batch_size, Ntrain)."""
        torch.manual_seed(self.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(self.seed)

        epochs = self.epochs
        X = np.asarray(X) if not torch.is_tensor(X) else X
        self._latent_size = latent_size = max(int(X.shape[1] / 16), 1)
        ndims = X.shape[1]
        train_size = X.shape[0]
        self.batch_size = min(self.batch_size, train_size)
        batches_per_epoch = train_size // self.batch_size

        device = self.device
        # the generator is initialised on the host with PyTorch's default Linear init (src/vgan.py:565-566),
        # consuming the seeded CPU RNG exactly like the reference, then moved to the GPU
        generator = Generator_big(img_size=ndims, latent_size=latent_size)
        generator = generator.to(device)
        self.generator_optimizer = "Adadelta"
        loss_function = MMDLossConstrained(weight=10)  # src/vgan.py:571

        data = torch.as_tensor(X).to(device=device, dtype=torch.float32).contiguous()  # resident in HBM for the whole fit
        engine = self._make_engine(generator, data, batches_per_epoch, loss_function)
        self._engine = engine

        # An epoch's loss is reported AFTER the next epoch's steps have been launched (the read-out waits for its own epoch only:
        # NoKLStepEngine.close_epoch / read_epoch_loss), so the GPU never idles behind the host's print, history append and next
        # launches; the printed lines keep the reference's order (src/vgan.py:589-625).
        pending = None

        def report(slot):
            generator_loss = engine.read_epoch_loss(slot)
            if self.verbose:
                print(f"Average loss in the epoch: {generator_loss}")
            self.train_history["generator_loss"].append(generator_loss)

        for epoch in range(epochs):
            if self.verbose and epoch == 0:
                print(f"\rEpoch {epoch} of {epochs}")
            if self.shuffle_source == "device":   # counter-based permutation evaluated on the GPU: no host draw, no copy
                engine.shuffle_epoch(epoch)
            else:
                if not engine.epoch_staged:
                    engine.stage_epoch_batches(epoch_batches(train_size, self.batch_size))
                engine.begin_epoch()
            if self.noise_source == "host":
                noise_tensor = torch.Tensor(self.batch_size, latent_size)  # src/vgan.py:594
            if self.noise_source == "host":
                for _ in range(batches_per_epoch):
                    engine.set_noise(noise_tensor.normal_())  # src/vgan.py:610
                    engine.step()
            else:
                engine.run_steps(batches_per_epoch)  # blocks of steps per graph launch (NoKLStepEngine.run_steps)
                # the next epoch's table is drawn and uploaded while this epoch's steps run (only with device noise: host noise
                # interleaves its draws with the DataLoader's on the same generator; and only if an epoch follows, so that the
                # fit consumes exactly the reference's draws)
                if self.shuffle_source != "device" and epoch + 1 < epochs:
                    engine.stage_epoch_batches(epoch_batches(train_size, self.batch_size))
            slot = engine.close_epoch()
            if loss_function.kernel.bandwidth is None:
                loss_function.kernel.bandwidth = engine.bw.view(())
            self.bandwidth = loss_function.kernel.bandwidth
            if os.environ.get("VGAN_FIT_SYNC_EACH_EPOCH") == "1":  # (A/B knob: report at once, as rounds 1-2 did)
                report(slot)
                if self.verbose and epoch + 1 < epochs:
                    print(f"\rEpoch {epoch + 1} of {epochs}")
                continue
            if pending is not None:
                report(pending)
                if self.verbose:
                    print(f"\rEpoch {epoch} of {epochs}")
            pending = slot
        if pending is not None:
            report(pending)

        self.generator = generator
        if self.path_to_directory is not None:
            self._save_run(generator, detector_too=False)


class VGAN(_RunFolder):
    """V-GAN with kernel learning (reference: src/vgan.py:20-431): an auto-encoder "detector" in front of
    the MMD, trained in alternation with the generator.  Runs on its own step engine (kl_trainer.KLStepEngine: explicit
    forward/backward on the HIP kernels, no autograd, each step kind replayed from a captured HIP graph); DESIGN.md section 7."""

    def __init__(self, batch_size=500, temperature=0, epochs=2000, lr_G=0.007, lr_D=0.007, iternum_d=1, iternum_g=5,
                 momentum=0.99, seed=777, weight_decay=0.04, path_to_directory=None):
        self.storage = locals()
        self.train_history = defaultdict(list)
        self.batch_size = batch_size
        self.temperature = temperature
        self.epochs = epochs
        self.lr_G = lr_G
        self.lr_D = lr_D
        self.iternum_d = iternum_d
        self.iternum_g = iternum_g
        self.momentum = momentum
        self.weight_decay = weight_decay
        self.path_to_directory = path_to_directory
        self.generator_optimizer = None
        self.device = _device()
        self.seed = 777  # the reference overrides the constructor's seed (src/vgan.py:48)
        # build-specific knobs (not constructor arguments, so the reference signature is unchanged): as in VGAN_no_kl
        self.noise_source = "device"   # "device": Philox on the GPU; "host": torch CPU generator (reference CPU-path RNG order)
        self.shuffle_source = "host"   # "host": the DataLoader's own draws (reference RNG order); "device": vgan_shuffle_epoch
        self.use_graph = True
        self.verbose = True

    def get_params(self):
        return {"batch size": self.batch_size, "epochs": self.epochs, "lr_g": self.lr_G, "momentum": self.momentum,
                "weight decay": self.weight_decay, "batch_size": self.batch_size, "seed": self.seed,
                "generator optimizer": self.generator_optimizer}

    @staticmethod
    def _weights_init(m):
        # src/vgan.py:69-78: Linear weights ~ N(0, 0.1), bias 0
        if m.__class__.__name__.find("Linear") != -1:
            m.weight.data.normal_(0.0, 0.1)
            m.bias.data.fill_(0)

    def get_the_networks(self, ndims, latent_size, device=None):
        if device is None:
            device = self.device
        generator = Generator_big(img_size=ndims, latent_size=latent_size).to(device)
        detector = Detector(latent_size, ndims, Encoder, Decoder).to(device)
        return generator, detector

    def fit(self, X):
        """src/vgan.py:178-353: 1 detector epoch, then 5 generator epochs, repeating."""
        torch.manual_seed(self.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(self.seed)
        X = np.asarray(X) if not torch.is_tensor(X) else X
        self._latent_size = latent_size = max(int(X.shape[1] / 16), 1)
        ndims = X.shape[1]
        train_size = X.shape[0]
        self.batch_size = min(self.batch_size, train_size)
        device = self.device

        generator = Generator_big(img_size=ndims, latent_size=latent_size)
        detector = Detector(latent_size, ndims, Encoder, Decoder)
        generator.apply(self._weights_init)
        detector.apply(self._weights_init)
        generator, detector = generator.to(device), detector.to(device)

        # the reference builds torch.optim.Adadelta for both networks (src/vgan.py:206-209); the generator's never sees a
        # gradient (see below) and the detector's update runs in the step engine's Adadelta kernel
        self.generator_optimizer = "Adadelta"
        self.detector_optimizer = "Adadelta"
        loss_function = MMDLossConstrained(weight=self.temperature)

        rank, world = _dist_info()
        if world > 1:
            # KLStepEngine has no row sharding: under torchrun every rank would train the whole problem and report it as if it
            # were a share.  The data-parallel path is VGAN_no_kl's (trainer.py); say so instead of silently replicating.
            import warnings
            warnings.warn(f"vgan_amd: VGAN.fit is not data-parallel -- torch.distributed is initialised with {world} ranks and "
                          f"rank {rank} will train the WHOLE problem on its own GPU (identical replicas, no speed-up); "
                          "use VGAN_no_kl for the row-sharded step", RuntimeWarning, stacklevel=2)
        data = torch.as_tensor(X).to(device=device, dtype=torch.float32).contiguous()
        batch_number = train_size // self.batch_size
        engine = KLStepEngine(self._ops(), generator, detector, data, self.batch_size, self.lr_D, self.weight_decay,
                              loss_function.weight, use_graph=self.use_graph, batches_per_epoch=batch_number,
                              noise=self.noise_source, seed=self.seed)
        self._engine = engine
        if loss_function.kernel.bandwidth is not None:  # process-wide RBF already calibrated (reference quirk)
            engine.set_bandwidth(float(loss_function.kernel.bandwidth))
        iternum_d = iternum_g = 1
        detector_loss = generator_loss = np.nan
        encoder_trainable = True

        def sync_bandwidth():
            if loss_function.kernel.bandwidth is None:
                loss_function.kernel.bandwidth = engine.bw.view(())
            loss_function.bandwidth = loss_function.kernel.bandwidth
            self.bandwidth = loss_function.bandwidth

        def new_epoch_table(epoch):
            # the epoch's shuffled drop_last batches become a device-resident table; the engine's step counter walks it
            if self.shuffle_source == "device":
                engine.shuffle_epoch(epoch)
            else:
                engine.set_epoch_batches(epoch_batches(train_size, self.batch_size))

        for epoch in range(self.epochs):
            if self.verbose:
                print(f"\rEpoch {epoch} of {self.epochs}")
            host_noise = self.noise_source == "host"
            if host_noise:
                noise_tensor = torch.Tensor(self.batch_size, latent_size)  # src/vgan.py:236
            if iternum_d <= self.iternum_d:
                new_epoch_table(epoch)
                for p in detector.decoder.parameters():  # src/vgan.py:257-258 (every step there; idempotent)
                    p.requires_grad = True
                if host_noise:
                    for _ in range(batch_number):
                        engine.detector_step(noise=noise_tensor.normal_(), train_encoder=encoder_trainable)
                else:
                    engine.detector_step(train_encoder=encoder_trainable, count=batch_number)
                mmd_sum, mse_sum = engine.epoch_sums()
                # batch_loss_D = -(MMD - .1 mse(batch, batch_dec) - .1 mse(projected, projected_dec)), src/vgan.py:275-277
                detector_loss = -(mmd_sum - 0.1 * mse_sum) / batch_number
                sync_bandwidth()
                iternum_d += 1
                iternum_g = 1
            elif iternum_g <= self.iternum_g:
                # Reference quirk, kept because it decides every number this phase produces: src/vgan.py:308-310 wraps the
                # generator output in the legacy ``Variable(...)`` constructor, which DETACHES it, and then sets requires_grad
                # on the detached leaf.  ``batch_loss_G.backward()`` (:326) therefore never reaches the generator's parameters,
                # ``gen_optimizer.step()`` (:327) sees no gradients and changes nothing: in VGAN.fit the generator keeps its
                # N(0, 0.1) initialisation (fixture f4: genT == gen0 bit for bit) and this phase only evaluates the loss.
                new_epoch_table(epoch)
                if host_noise:
                    for _ in range(batch_number):
                        engine.generator_phase_step(noise=noise_tensor.normal_())
                else:
                    engine.generator_phase_step(count=batch_number)
                for p in detector.parameters():  # src/vgan.py:319-320 (every step there): freezes the detector for good
                    p.requires_grad = False
                encoder_trainable = False
                generator_loss = engine.epoch_sums()[0] / batch_number
                sync_bandwidth()
                iternum_g += 1
                if iternum_g > self.iternum_g:
                    iternum_d = 1
            if self.verbose:
                print(f"Average loss in the epoch Generator: {generator_loss}")
                print(f"Average loss in the epoch Detector: {detector_loss}")
            self.train_history["generator_loss"].append(generator_loss)
            self.train_history["detector_loss"].append(detector_loss)

        self.generator = generator
        self.detector = detector
        if self.path_to_directory is not None:
            self._save_run(generator, detector_too=True)
