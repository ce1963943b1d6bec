#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Run only in the build container (the reference is mounted read-only at /root/reference and
never travels to the GPU box; only the .npz files written here do):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from the reference: ``src.models.{Generator,Detector,Mmd_loss_constrained}``
and ``src.vgan`` (``VGAN_no_kl``, ``VGAN``).  ``src/vgan.py:5`` imports ``torch_two_sample``,
which is not installed; it is only touched inside ``check_if_myopic``, so an
EMPTY module object is registered under that name to let the import statement succeed --
nothing of torch-two-sample is emulated and ``check_if_myopic`` is never called.

Fixtures store ARRAYS (inputs, recorded noise, recorded batch indices, outputs), never seeds
alone, so they stay valid across torch versions.

  f1_ops_*.npz     op level: (logits, X) -> U, Y ; (X, Y, U, weight) -> loss, bw, dY, dU
  f2_step_*.npz    one full VGAN_no_kl step: params0, noise, batch -> loss, grads, params1, state
  f3_traj_c1.npz   VGAN_no_kl.fit at c1 (d=20, batch=128, 200 steps): recorded batches + noise,
                   per-step loss, final params, generate_subspaces(500) masks
  f4_kl_c1.npz     VGAN.fit (kernel learning) 12 epochs at c1: both loss histories + masks
  f5_c3_scalars.npz  c3 (d=784, n=1024) single step, fp64 + fp32 scalars on documented inputs
  f7_mmd_unequal.npz  MMDLossConstrained with DIFFERENT row counts of X and Y (Mmd_loss_constrained.py:46-49 takes block means
                   over whatever shapes it is given; no call site of the reference does this): loss, bandwidth, dX, dY, dU, and
                   a second call on the frozen bandwidth
  f6_ref_generator_c1.pt, f6_ref_run.npz   a run folder written by the reference's fit(path_to_directory=...): its saved
                   generator state_dict, the masks the reference samples from it after load_models, its CSV texts
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
# the reference's `src` is a namespace package (no __init__.py) while this repository's `src` is a regular one, which would
# win wherever it sits on sys.path: the repository root is only added AFTER the reference's modules have been imported
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("torch_two_sample", types.ModuleType("torch_two_sample"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from src.models.Generator import Generator_big, upper_softmax  # noqa: E402  (reference)
from src.models.Mmd_loss_constrained import MMDLossConstrained, RBF  # noqa: E402  (reference)
import src.vgan as ref_vgan  # noqa: E402  (reference)

assert ref_vgan.__file__.startswith("/root/reference"), ref_vgan.__file__

sys.path.insert(1, REPO)
from oracle import vgan_oracle as orc  # noqa: E402  (only for the documented synthetic inputs)

torch.set_num_threads(8)


def reset_shared_rbf():
    """The default ``kernel=RBF()`` is one process-wide object (Mmd_loss_constrained.py:35)."""
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None


def t2n(t):
    return t.detach().cpu().numpy().copy()


# ---------------------------------------------------------------------------- F1
def make_f1():
    for (n, d) in [(8, 4), (64, 12), (128, 20), (512, 166)]:
        rng = np.random.default_rng(100 + n)
        logits32 = rng.normal(size=(n, d)).astype(np.float32) * 1.5
        X32 = (rng.normal(size=(n, d)) + rng.choice([-2.0, 2.0], size=(n, 1))).astype(np.float32)
        gU32 = rng.normal(size=(n, d)).astype(np.float32)
        out = dict(logits=logits32, X=X32, gU=gU32, weight=np.float64(10.0))
        for dt, tag in [(torch.float32, "f32"), (torch.float64, "f64")]:
            logits = torch.tensor(logits32, dtype=dt, requires_grad=True)
            X = torch.tensor(X32, dtype=dt)
            U = upper_softmax()(logits)
            Ud = U.detach().clone().requires_grad_(True)
            Y = (Ud * X).detach().clone().requires_grad_(True)
            loss_fn = MMDLossConstrained(weight=10, kernel=RBF())
            loss = loss_fn(X, Y, Ud)  # first call: calibrates the bandwidth
            dY, dU = torch.autograd.grad(loss, [Y, Ud])
            # second call with the frozen bandwidth on perturbed Y (exercises the non-calibrating path)
            Y2 = (Y.detach() * 0.9).requires_grad_(True)
            loss2 = loss_fn(X, Y2, Ud)
            (dY2,) = torch.autograd.grad(loss2, [Y2])
            # upper_softmax backward with an arbitrary upstream gradient
            gU = torch.tensor(gU32, dtype=dt)
            (dlogits,) = torch.autograd.grad(U, [logits], grad_outputs=gU)
            big = (n * d > 20000 and tag == "f64")
            out[f"U_{tag}"] = t2n(U)
            out[f"loss_{tag}"] = t2n(loss)
            out[f"bw_{tag}"] = t2n(loss_fn.bandwidth)
            out[f"loss2_{tag}"] = t2n(loss2)
            if not big:
                out[f"dY_{tag}"] = t2n(dY)
                out[f"dU_{tag}"] = t2n(dU)
                out[f"dY2_{tag}"] = t2n(dY2)
                out[f"dlogits_{tag}"] = t2n(dlogits)
        np.savez_compressed(os.path.join(HERE, f"f1_ops_n{n}_d{d}.npz"), **out)
        print("f1", n, d, out["loss_f32"], out["loss_f64"], out["bw_f32"])


# ---------------------------------------------------------------------------- F2
def make_f2():
    for tag, n, d in [("c1", 128, 20), ("c2", 512, 166)]:
        data = orc.synthetic_dataset(tag)
        rng = np.random.default_rng(7)
        idx = rng.permutation(data.shape[0])[:n]
        batch = torch.tensor(data[idx])
        L = max(int(d / 16), 1)
        torch.manual_seed(123)
        gen = Generator_big(latent_size=L, img_size=d)
        params0 = [t2n(p) for p in gen.parameters()]
        noise = torch.randn(n, L)
        opt = torch.optim.Adadelta(gen.parameters(), lr=0.007, weight_decay=0.04)
        loss_fn = MMDLossConstrained(weight=10, kernel=RBF())
        out = dict(batch=t2n(batch), noise=t2n(noise))
        for step in range(2):  # two steps: the second one has non-zero Adadelta state + frozen bw
            opt.zero_grad()
            U = gen(noise)
            loss = loss_fn(batch, U * batch, U)
            loss.backward()
            grads = [t2n(p.grad) for p in gen.parameters()]
            opt.step()
            out[f"loss{step}"] = t2n(loss)
            for i, g in enumerate(grads):
                out[f"grad{step}_{i}"] = g
            for i, p in enumerate(gen.parameters()):
                out[f"param{step + 1}_{i}"] = t2n(p)
                st = opt.state[p]
                out[f"sq{step + 1}_{i}"] = t2n(st["square_avg"])
                out[f"acc{step + 1}_{i}"] = t2n(st["acc_delta"])
            if step == 0:
                out["U0"] = t2n(U)
        for i, p in enumerate(params0):
            out[f"param0_{i}"] = p
        out["bw"] = t2n(loss_fn.bandwidth)
        np.savez_compressed(os.path.join(HERE, f"f2_step_{tag}.npz"), **out)
        print("f2", tag, out["loss0"], out["loss1"], out["bw"])


# ---------------------------------------------------------------------------- recording hooks
class Recorder:
    """Wraps the reference's Generator_big.forward / MMDLossConstrained.forward at class level
    to record what the reference's own fit() feeds them.  Nothing is altered."""

    def __init__(self):
        self.noise, self.batches, self.losses, self.bws, self.us = [], [], [], [], []

    def __enter__(self):
        self._g, self._m = Generator_big.forward, MMDLossConstrained.forward
        rec = self

        def g_fwd(mod, z):
            rec.noise.append(t2n(z))
            return rec._g(mod, z)

        def m_fwd(mod, X, Y, U):
            out = rec._m(mod, X, Y, U)
            rec.batches.append(t2n(X))
            rec.losses.append(float(out.detach()))
            rec.bws.append(float(mod.bandwidth))
            return out

        Generator_big.forward, MMDLossConstrained.forward = g_fwd, m_fwd
        return self

    def __exit__(self, *a):
        Generator_big.forward, MMDLossConstrained.forward = self._g, self._m


def rows_to_indices(data, batches):
    """Recover DataLoader's shuffled indices from the recorded batches (rows are unique)."""
    key = {row.tobytes(): i for i, row in enumerate(data)}
    assert len(key) == data.shape[0]
    return np.array([[key[r.tobytes()] for r in b] for b in batches], dtype=np.int32)


# ---------------------------------------------------------------------------- F3
def make_f3():
    data = orc.synthetic_dataset("c1", rows=1280)  # 10 batches of 128 per epoch, 20 epochs = 200 steps
    reset_shared_rbf()
    model = ref_vgan.VGAN_no_kl(batch_size=128, epochs=20, seed=777)
    model.device = torch.device("cpu")
    captured = {}
    orig = ref_vgan.VGAN_no_kl.get_the_networks

    def grab(self, *a, **k):
        g = orig(self, *a, **k)
        captured["params0"] = [t2n(p) for p in g.parameters()]
        return g

    ref_vgan.VGAN_no_kl.get_the_networks = grab
    try:
        with Recorder() as rec:
            model.fit(data)
            nfit = len(rec.noise)
            masks = model.generate_subspaces(500)
            mask_noise = rec.noise[-1]
    finally:
        ref_vgan.VGAN_no_kl.get_the_networks = orig
    assert nfit == 200 and len(rec.batches) == 200
    out = dict(data=data, idx=rows_to_indices(data, rec.batches), noise=np.stack(rec.noise[:nfit]),
               losses=np.array(rec.losses), bw=np.float64(rec.bws[0]),
               epoch_losses=np.array(model.train_history["generator_loss"]),
               masks=t2n(masks), mask_noise=mask_noise,
               lr=np.float64(model.lr), weight_decay=np.float64(model.weight_decay))
    assert all(b == rec.bws[0] for b in rec.bws)
    for i, p in enumerate(captured["params0"]):
        out[f"param0_{i}"] = p
    for i, p in enumerate(model.generator.parameters()):
        out[f"paramT_{i}"] = t2n(p)
    np.savez_compressed(os.path.join(HERE, "f3_traj_c1.npz"), **out)
    print("f3", out["epoch_losses"][:3], out["epoch_losses"][-1], out["bw"], out["masks"].sum(0))


# ---------------------------------------------------------------------------- F4
def make_f4():
    data = orc.synthetic_dataset("c1", rows=1280)
    reset_shared_rbf()
    model = ref_vgan.VGAN(batch_size=128, epochs=12)
    model.device = torch.device("cpu")
    first, det_inputs = {}, []
    orig_det, orig_gen = ref_vgan.Detector.forward, Generator_big.forward

    def det_fwd(mod, x):  # called twice per step: raw batch first, then U * batch (src/vgan.py:268,274)
        if "d" not in first:  # parameters right after the reference's weights_init (src/vgan.py:202-205)
            first["d"] = [t2n(p) for p in mod.parameters()]
        det_inputs.append(t2n(x))
        return orig_det(mod, x)

    def gen_fwd(mod, z):
        if "g" not in first:
            first["g"] = [t2n(p) for p in mod.parameters()]
        return orig_gen(mod, z)

    ref_vgan.Detector.forward, Generator_big.forward = det_fwd, gen_fwd
    try:
        with Recorder() as rec:
            model.fit(data)
            nfit = len(rec.noise)
            masks = model.generate_subspaces(500)
            mask_noise = rec.noise[-1]
    finally:
        ref_vgan.Detector.forward, Generator_big.forward = orig_det, orig_gen
    raw_batches = det_inputs[0::2][:nfit]
    out = dict(data=data, idx=rows_to_indices(data, raw_batches), noise=np.stack(rec.noise[:nfit]),
               losses=np.array(rec.losses), bw=np.float64(rec.bws[0]),
               generator_loss=np.array(model.train_history["generator_loss"], dtype=np.float64),
               detector_loss=np.array(model.train_history["detector_loss"], dtype=np.float64),
               masks=t2n(masks), mask_noise=mask_noise)
    for i, p in enumerate(first["g"]):
        out[f"gen0_{i}"] = p
    for i, p in enumerate(first["d"]):
        out[f"det0_{i}"] = p
    for i, p in enumerate(model.generator.parameters()):
        out[f"genT_{i}"] = t2n(p)
    for i, p in enumerate(model.detector.parameters()):
        out[f"detT_{i}"] = t2n(p)
        out[f"detT_rg_{i}"] = np.bool_(p.requires_grad)
    np.savez_compressed(os.path.join(HERE, "f4_kl_c1.npz"), **out)
    print("f4", out["generator_loss"], out["detector_loss"], out["bw"], nfit)


# ---------------------------------------------------------------------------- F5
def make_f5():
    n, d = 1024, 784
    data = orc.synthetic_dataset("c3", rows=2048)
    batch = data[:n]
    L = orc.latent_size(d)
    z = np.random.default_rng(5).normal(size=(n, L)).astype(np.float32)
    params = orc.synthetic_generator_params(d)
    out = {}
    for dt, tag in [(torch.float64, "f64"), (torch.float32, "f32")]:
        gen = Generator_big(latent_size=L, img_size=d).to(dt)
        with torch.no_grad():
            for p, v in zip(gen.parameters(), params):
                p.copy_(torch.tensor(v, dtype=dt))
        X = torch.tensor(batch, dtype=dt)
        loss_fn = MMDLossConstrained(weight=10, kernel=RBF())
        U = gen(torch.tensor(z, dtype=dt))
        loss = loss_fn(X, U * X, U)
        loss.backward()
        K = loss_fn.kernel(torch.vstack([X, (U * X).detach()]))
        out[f"loss_{tag}"] = t2n(loss)
        out[f"bw_{tag}"] = t2n(loss_fn.bandwidth)
        out[f"xx_{tag}"] = t2n(K[:n, :n].mean())
        out[f"xy_{tag}"] = t2n(K[:n, n:].mean())
        out[f"yy_{tag}"] = t2n(K[n:, n:].mean())
        out[f"nsel_{tag}"] = np.int64((U.detach() >= 1 / d).sum())
        for i, p in enumerate(gen.parameters()):
            g = t2n(p.grad)
            out[f"gnorm_{tag}_{i}"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            if i == 6:  # last weight: keep a thin slice of the gradient as a direct check
                out[f"g6slice_{tag}"] = g[:8, :16].copy()
        # second step's loss with the frozen bandwidth, after one Adadelta update
    np.savez_compressed(os.path.join(HERE, "f5_c3_scalars.npz"), **out)
    print("f5", out["loss_f64"], out["loss_f32"], out["bw_f64"], out["nsel_f64"])


# ---------------------------------------------------------------------------- F6
def make_f6():
    """A run folder written by the REFERENCE (fit with path_to_directory, src/vgan.py:626-635): its generator_0.pt (a plain
    state_dict, loadable with weights_only=True) is kept as a fixture together with what the reference's own load_models +
    generate_subspaces return for it, and the text of params.csv / generator_loss_0.csv (the file layout to reproduce)."""
    import shutil
    import tempfile
    data = orc.synthetic_dataset("c1", rows=640)
    reset_shared_rbf()
    with tempfile.TemporaryDirectory() as tmp:
        run = os.path.join(tmp, "run")
        model = ref_vgan.VGAN_no_kl(batch_size=128, epochs=3, seed=5, path_to_directory=run)
        model.device = torch.device("cpu")
        model.fit(data)
        shutil.copyfile(os.path.join(run, "models", "generator_0.pt"), os.path.join(HERE, "f6_ref_generator_c1.pt"))
        fresh = ref_vgan.VGAN_no_kl(seed=5)
        fresh.device = torch.device("cpu")
        fresh.load_models(os.path.join(run, "models", "generator_0.pt"), ndims=20, device="cpu")
        fresh._latent_size = 1  # set by fit() in the reference; load_models alone leaves it unset (src/vgan.py:511-527)
        masks = t2n(fresh.generate_subspaces(64))
        torch.manual_seed(5)  # what generate_subspaces drew (src/vgan.py:641-644), recorded as an array
        mask_noise = torch.Tensor(64, 1).normal_()
        assert np.array_equal(t2n(fresh.generator(mask_noise) >= 1 / 20), masks)
        out = dict(masks=masks, mask_noise=t2n(mask_noise), files=np.array(sorted(os.listdir(run))), model_files=np.array(sorted(os.listdir(os.path.join(run, "models")))),
                   params_csv=np.array(open(os.path.join(run, "params.csv")).read()),
                   loss_csv=np.array(open(os.path.join(run, "train_history", "generator_loss_0.csv")).read()),
                   epoch_losses=np.array(model.train_history["generator_loss"]))
        for i, q in enumerate(model.generator.parameters()):
            out[f"param_{i}"] = t2n(q)
    np.savez_compressed(os.path.join(HERE, "f6_ref_run.npz"), **out)
    print("f6", out["files"], out["model_files"], str(out["params_csv"])[:200], masks.sum(0))


# ---------------------------------------------------------------------------- F7
def make_f7():
    out = {}
    for k, (nx, ny, nu, d) in enumerate([(48, 80, 80, 12), (200, 72, 50, 30)]):
        rng = np.random.default_rng(700 + k)
        X32 = (rng.normal(size=(nx, d)) + rng.choice([-1.5, 1.5], size=(nx, 1))).astype(np.float32)
        Y32 = (rng.normal(size=(ny, d)) * 0.8).astype(np.float32)
        U32 = rng.uniform(0.0, 1.0, size=(nu, d)).astype(np.float32)
        out.update({f"X{k}": X32, f"Y{k}": Y32, f"U{k}": U32})
        for dt, tag in [(torch.float32, "f32"), (torch.float64, "f64")]:
            X = torch.tensor(X32, dtype=dt, requires_grad=True)
            Y = torch.tensor(Y32, dtype=dt, requires_grad=True)
            U = torch.tensor(U32, dtype=dt, requires_grad=True)
            loss_fn = MMDLossConstrained(weight=3.0, kernel=RBF())
            loss = loss_fn(X, Y, U)  # first call: calibrates the bandwidth on the stacked [X; Y]
            dX, dY, dU = torch.autograd.grad(loss, [X, Y, U])
            Y2 = (Y.detach() * 1.1 + 0.05).requires_grad_(True)
            loss2 = loss_fn(X, Y2, U)  # frozen bandwidth
            (dY2,) = torch.autograd.grad(loss2, [Y2])
            for name, v in (("loss", loss), ("bw", loss_fn.bandwidth), ("dX", dX), ("dY", dY), ("dU", dU), ("loss2", loss2), ("dY2", dY2)):
                out[f"{name}{k}_{tag}"] = t2n(v)
        print("f7", k, out[f"loss{k}_f32"], out[f"loss{k}_f64"], out[f"bw{k}_f32"])
    out["weight"] = np.float64(3.0)
    np.savez_compressed(os.path.join(HERE, "f7_mmd_unequal.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5", "f6", "f7"]
    for w in which:
        globals()[f"make_{w}"]()
