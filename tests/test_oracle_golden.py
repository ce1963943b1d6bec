"""Pins the CPU oracle (oracle/vgan_oracle.py, oracle/torch_port.py) against fixtures produced by
running the reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import vgan_oracle as orc
from oracle import torch_port as port
from conftest import load_golden

F1 = ["f1_ops_n8_d4.npz", "f1_ops_n64_d12.npz", "f1_ops_n128_d20.npz", "f1_ops_n512_d166.npz"]


def check_penalty_grad(dU, ref, U, rtol):
    """-weight/d lands on ONE maximal row per column (src/models/Mmd_loss_constrained.py:50,
    topk(U,1,0)).  Which one torch.topk picks among exact ties (several rows snapped to 1.0) is
    unspecified, so tied columns are checked structurally; unique maxima must match exactly."""
    colmax = U.max(axis=0)
    ties = (U == colmax).sum(axis=0) > 1
    np.testing.assert_allclose(dU[:, ~ties], ref[:, ~ties], rtol=rtol, atol=1e-12)
    np.testing.assert_allclose(dU.sum(0), ref.sum(0), rtol=rtol)
    for a in (dU, ref):
        assert ((a != 0).sum(0) == 1).all() and (U[a != 0] == np.broadcast_to(colmax, U.shape)[a != 0]).all()


@pytest.mark.parametrize("name", F1)
@pytest.mark.parametrize("tag,dt,rtol", [("f32", np.float32, 2e-5), ("f64", np.float64, 1e-10)])
def test_f1_ops(name, tag, dt, rtol):
    g = load_golden(name)
    logits, X = g["logits"].astype(dt), g["X"].astype(dt)
    U, s = orc.upper_softmax_forward(logits)
    # mask decisions must agree exactly with the reference, values to rounding
    assert np.array_equal(U == 1, g[f"U_{tag}"] == 1)
    np.testing.assert_allclose(U, g[f"U_{tag}"], rtol=rtol, atol=0)
    Y = U * X
    f = orc.mmd_forward(X, Y, U, 10.0)
    np.testing.assert_allclose(f["loss"], g[f"loss_{tag}"], rtol=rtol)
    np.testing.assert_allclose(f["bw"], g[f"bw_{tag}"], rtol=rtol)
    f2 = orc.mmd_forward(X, (Y * dt(0.9)), U, 10.0, bw=f["bw"])
    np.testing.assert_allclose(f2["loss"], g[f"loss2_{tag}"], rtol=rtol)
    if f"dY_{tag}" in g:
        dY, dU = orc.mmd_backward(X, Y, U, 10.0, f["bw"])
        scale = np.abs(g[f"dY_{tag}"]).max()
        np.testing.assert_allclose(dY, g[f"dY_{tag}"], rtol=0, atol=scale * (50 * rtol))
        check_penalty_grad(dU, g[f"dU_{tag}"], U, rtol)
        dY2, _ = orc.mmd_backward(X, Y * dt(0.9), U, 10.0, f["bw"])
        np.testing.assert_allclose(dY2, g[f"dY2_{tag}"], rtol=0, atol=np.abs(g[f"dY2_{tag}"]).max() * 50 * rtol)
        dl = orc.upper_softmax_backward(g["gU"].astype(dt), s)
        np.testing.assert_allclose(dl, g[f"dlogits_{tag}"], rtol=0, atol=np.abs(g[f"dlogits_{tag}"]).max() * 50 * rtol)


@pytest.mark.parametrize("cfg", ["c1", "c2"])
def test_f2_full_step(cfg):
    g = load_golden(f"f2_step_{cfg}.npz")
    for dt, tol in [(np.float32, 3e-4), (np.float64, 2e-5)]:
        params = [g[f"param0_{i}"].astype(dt) for i in range(8)]
        tr = orc.NoKLTrainer(params)
        X, z = g["batch"].astype(dt), g["noise"].astype(dt)
        for step in range(2):
            out = tr.step(X, z)
            assert abs(out["loss"] - g[f"loss{step}"]) < 1e-5
            for i in range(8):
                ref = g[f"grad{step}_{i}"]
                np.testing.assert_allclose(out["grads"][i], ref, rtol=0, atol=tol * max(np.abs(ref).max(), 1e-8))
                np.testing.assert_allclose(tr.params[i], g[f"param{step + 1}_{i}"], rtol=0, atol=2e-6)
                np.testing.assert_allclose(tr.sq[i], g[f"sq{step + 1}_{i}"], rtol=2e-3, atol=1e-14)
                np.testing.assert_allclose(tr.acc[i], g[f"acc{step + 1}_{i}"], rtol=2e-3, atol=1e-14)
            if step == 0:
                assert np.array_equal(out["U"] == 1, g["U0"] == 1)
        np.testing.assert_allclose(tr.bw, g["bw"], rtol=1e-5)


def test_f3_trajectory_c1():
    """200 steps of VGAN_no_kl.fit at c1 (BASELINE.json configs[0]) on the recorded batches/noise."""
    g = load_golden("f3_traj_c1.npz")
    data = g["data"]
    assert np.array_equal(data, orc.synthetic_dataset("c1", rows=1280))  # documented generator
    for dt in (np.float32, np.float64):
        tr = orc.NoKLTrainer([g[f"param0_{i}"].astype(dt) for i in range(8)],
                             lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
        losses = []
        for t in range(200):
            out = tr.step(data[g["idx"][t]].astype(dt), g["noise"][t].astype(dt))
            losses.append(out["loss"])
        np.testing.assert_allclose(losses, g["losses"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(tr.bw, g["bw"], rtol=1e-5)
        ep = np.array(losses).reshape(20, 10).mean(1)
        np.testing.assert_allclose(ep, g["epoch_losses"], rtol=0, atol=1e-5)
        masks = tr.masks(g["mask_noise"].astype(dt))
        assert np.array_equal(masks, g["masks"])  # Hamming distance 0
        for i in range(8):
            np.testing.assert_allclose(tr.params[i], g[f"paramT_{i}"], rtol=0, atol=5e-5)


def test_f5_c3_scalars():
    g = load_golden("f5_c3_scalars.npz")
    n, d = 1024, 784
    X = orc.synthetic_dataset("c3", rows=2048)[:n]
    z = np.random.default_rng(5).normal(size=(n, orc.latent_size(d))).astype(np.float32)
    params = orc.synthetic_generator_params(d)
    for dt, tag, tol in [(np.float64, "f64", 1e-9), (np.float32, "f32", 2e-5)]:
        out = orc.step_forward_backward([p.astype(dt) for p in params], X.astype(dt), z.astype(dt), 10.0)
        assert abs(out["loss"] - g[f"loss_{tag}"]) < max(tol, 1e-9) * 5
        np.testing.assert_allclose(out["bw"], g[f"bw_{tag}"], rtol=max(tol, 1e-9))
        np.testing.assert_allclose(out["xx"], g[f"xx_{tag}"], rtol=max(tol, 1e-9))
        np.testing.assert_allclose(out["xy"], g[f"xy_{tag}"], rtol=max(tol, 1e-9))
        np.testing.assert_allclose(out["yy"], g[f"yy_{tag}"], rtol=max(tol, 1e-9))
        assert int(orc.subspace_mask(out["U"]).sum()) == int(g[f"nsel_{tag}"])
        for i in range(8):
            gn = np.sqrt((out["grads"][i].astype(np.float64) ** 2).sum())
            np.testing.assert_allclose(gn, g[f"gnorm_{tag}_{i}"], rtol=2e-3 if dt == np.float32 else 1e-7)


def test_torch_port_matches_reference_step():
    """The op-for-op PyTorch-CPU port (timed as cpu_baseline) reproduces the reference's two steps."""
    g = load_golden("f2_step_c1.npz")
    tr = port.PortNoKL([g[f"param0_{i}"] for i in range(8)])
    X, z = torch.tensor(g["batch"]), torch.tensor(g["noise"])
    for step in range(2):
        loss = tr.step(X, z)
        assert abs(loss - float(g[f"loss{step}"])) < 1e-6
        for i in range(8):
            np.testing.assert_allclose(tr.params[i].detach().numpy(), g[f"param{step + 1}_{i}"], rtol=0, atol=1e-7)


def test_f6_reference_saved_checkpoint_loads_weights_only():
    """The generator_0.pt a reference fit wrote (src/vgan.py:626-635) is a plain state_dict: keys main.{0..3}.{weight,bias},
    loadable with weights_only=True; the oracle's generator on the recorded noise reproduces the masks the reference sampled
    from it after its own load_models (src/vgan.py:511-527, :639-647)."""
    import os
    from conftest import REPO
    g = load_golden("f6_ref_run.npz")
    sd = torch.load(os.path.join(REPO, "tests", "golden", "f6_ref_generator_c1.pt"), map_location="cpu", weights_only=True)
    assert list(sd) == [f"main.{k}.{w}" for k in range(4) for w in ("weight", "bias")]
    params = [v.numpy() for v in sd.values()]
    for i, q in enumerate(params):
        assert np.array_equal(q, g[f"param_{i}"])
    masks = orc.NoKLTrainer(params).masks(g["mask_noise"])
    assert np.array_equal(masks, g["masks"])
    assert list(g["files"]) == ["models", "params.csv", "train_history", "train_history.pdf"]


def test_f4_port_kl_reproduces_reference_vgan_fit():
    """oracle/torch_port.PortKL (the op-for-op port of VGAN.fit's two step bodies) replayed on the batches and noise the
    reference's own 12-epoch run recorded (fixture f4; epoch 0 and 6 detector, the others generator phases): every step's MMD
    term, both epoch histories, the bandwidth, the final detector parameters and their requires_grad flags."""
    g = load_golden("f4_kl_c1.npz")
    data = g["data"]
    tr = port.PortKL([g[f"gen0_{i}"] for i in range(8)], [g[f"det0_{i}"] for i in range(16)], weight=0.0)
    nb = 10
    det_hist, gen_hist, mmds = [], [], []
    det_loss = gen_loss = np.nan
    for epoch in range(12):
        steps = range(epoch * nb, (epoch + 1) * nb)
        if epoch % 6 == 0:
            out = [tr.detector_step(torch.as_tensor(data[g["idx"][t]]), torch.as_tensor(g["noise"][t])) for t in steps]
            det_loss = sum(o[0] for o in out) / nb
            mmds += [o[1] for o in out]
        else:
            out = [tr.generator_phase_step(torch.as_tensor(data[g["idx"][t]]), torch.as_tensor(g["noise"][t])) for t in steps]
            gen_loss = sum(out) / nb
            mmds += out
        det_hist.append(det_loss)
        gen_hist.append(gen_loss)
    np.testing.assert_allclose(mmds, g["losses"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(det_hist, g["detector_loss"], rtol=0, atol=2e-6)
    assert np.isnan(gen_hist[0]) and np.isnan(g["generator_loss"][0])
    np.testing.assert_allclose(gen_hist[1:], g["generator_loss"][1:], rtol=0, atol=2e-6)
    np.testing.assert_allclose(float(tr.kernel.bandwidth), float(g["bw"]), rtol=1e-6)
    for i, q in enumerate(tr.det):
        np.testing.assert_allclose(q.detach().numpy(), g[f"detT_{i}"], rtol=0, atol=1e-6)
        assert bool(q.requires_grad) == bool(g[f"detT_rg_{i}"])


def test_oracle_mmd_with_unequal_row_counts_vs_reference():
    """Fixture F7: the reference's MMDLossConstrained on X and Y of DIFFERENT row counts (Mmd_loss_constrained.py:46-49: block
    means over n_x^2, n_x n_y, n_y^2 entries; U with its own row count) -- loss, calibrated bandwidth, dX, dY, dU and a second
    call on the frozen bandwidth, float64 and float32."""
    g = load_golden("f7_mmd_unequal.npz")
    w = float(g["weight"])
    for k in (0, 1):
        for tag, dt, tol in (("f64", np.float64, 1e-10), ("f32", np.float32, 2e-5)):
            X, Y, U = g[f"X{k}"].astype(dt), g[f"Y{k}"].astype(dt), g[f"U{k}"].astype(dt)
            f = orc.mmd_forward(X, Y, U, w)
            assert abs(f["loss"] - float(g[f"loss{k}_{tag}"])) < tol * max(1.0, abs(float(g[f"loss{k}_{tag}"])))
            np.testing.assert_allclose(float(f["bw"]), float(g[f"bw{k}_{tag}"]), rtol=10 * tol)
            dX, dY, dU = orc.mmd_backward(X, Y, U, w, f["bw"], with_dx=True)
            for got, name in ((dX, "dX"), (dY, "dY"), (dU, "dU")):
                ref = g[f"{name}{k}_{tag}"]
                np.testing.assert_allclose(got, ref, rtol=0, atol=(1e-9 if dt is np.float64 else 2e-5) * max(np.abs(ref).max(), 1e-12) + 1e-12)
            Y2 = (Y * dt(1.1) + dt(0.05)).astype(dt)
            f2 = orc.mmd_forward(X, Y2, U, w, bw=f["bw"])
            assert abs(f2["loss"] - float(g[f"loss2{k}_{tag}"])) < tol * max(1.0, abs(float(g[f"loss2{k}_{tag}"])))
