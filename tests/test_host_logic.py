"""Host logic of the product on the CPU tier: RNG/shuffle order, flat parameter layout, the step engine and
the fit loop, run over the test-only CpuOps provider (tests/cpu_ops.py) and checked against the oracle and
the reference-generated fixtures."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from cpu_ops import CpuOps
from kl_cases import kl_resident_feed_equals_per_step_feed
from oracle import vgan_oracle as orc


def test_epoch_batches_match_dataloader_rng_order():
    from vgan_amd import vgan as V
    for N, n in [(1280, 128), (2000, 500), (77, 10)]:
        torch.manual_seed(5)
        a = V._epoch_batches_dataloader(N, n)
        sa = torch.random.get_rng_state()
        torch.manual_seed(5)
        b = V.epoch_batches(N, n)
        sb = torch.random.get_rng_state()
        assert torch.equal(a, b) and torch.equal(sa, sb)
        assert b.shape == (N // n, n)


def test_synthetic_inputs_identical_to_oracle_copy():
    from vgan_amd import synth
    for cfg, rows in [("c1", 300), ("c2", 200), ("c3", 64), ("c4", 8), ("c5", 4)]:
        assert np.array_equal(synth.synthetic_dataset(cfg, rows=rows), orc.synthetic_dataset(cfg, rows=rows))
    for d in (20, 166, 784, 2048, 4096):  # c1 .. c5
        assert synth.latent_size(d) == orc.latent_size(d)
        for a, b in zip(synth.synthetic_generator_params(d), orc.synthetic_generator_params(d)):
            assert np.array_equal(a, b)


def make_engine(params, data, n, nb, **kw):
    from vgan_amd.modules import Generator_big
    from vgan_amd.trainer import NoKLStepEngine
    d = data.shape[1]
    gen = Generator_big(orc.latent_size(d), d)
    with torch.no_grad():
        for q, v in zip(gen.parameters(), params):
            q.copy_(torch.as_tensor(v))
    return NoKLStepEngine(CpuOps(), gen, torch.as_tensor(data), n, nb, noise="host", loss_accum_scale=1.0, **kw), gen


def test_flat_params_layout_and_views():
    g = load_golden("f2_step_c2.npz")
    eng, gen = make_engine([g[f"param0_{i}"] for i in range(8)], g["batch"], 512, 1)
    fp = eng.fp
    assert all(o % 4 == 0 for o in fp.offsets) and fp.total % 4 == 0
    for i, q in enumerate(gen.parameters()):
        assert q.data.data_ptr() == fp.view(fp.flat, i).data_ptr()  # module parameters ARE the flat buffer
        assert np.array_equal(q.detach().numpy(), g[f"param0_{i}"])
    assert list(gen.state_dict()) == [f"main.{k}.{w}" for k in range(4) for w in ("weight", "bias")]


@pytest.mark.parametrize("mode", ["collapsed", "layered"])
@pytest.mark.parametrize("cfg", ["c1", "c2"])
def test_engine_two_steps_vs_reference_fixture(cfg, mode):
    g = load_golden(f"f2_step_{cfg}.npz")
    n = g["batch"].shape[0]
    eng, _ = make_engine([g[f"param0_{i}"] for i in range(8)], g["batch"], n, 1, generator_mode=mode)
    eng.set_epoch_batches(torch.arange(n).view(1, n))
    for step in range(2):
        eng.set_noise(torch.as_tensor(g["noise"]))
        eng.step()
        assert abs(float(eng.loss) - float(g[f"loss{step}"])) < 2e-5
        for i in range(8):
            ref = g[f"grad{step}_{i}"]
            np.testing.assert_allclose(eng.grad_view(i).numpy(), ref, rtol=0, atol=1e-3 * max(np.abs(ref).max(), 1e-8))
            np.testing.assert_allclose(eng.fp.view(eng.fp.flat, i).numpy(), g[f"param{step + 1}_{i}"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(float(eng.bw), float(g["bw"]), rtol=1e-5)


def test_engine_flop_minimal_chain_association_vs_reference_fixture():
    """chain_assoc='flops' (At_k = Wt_k At_{k-1}, M_{k-1} = Wt_k^T M_k: what the engine picks once the suffix product B_3 passes
    1 GFLOP, i.e. from c4 up) against the reference's two steps of fixture f2 (c2): same losses, gradients and parameters."""
    g = load_golden("f2_step_c2.npz")
    n = g["batch"].shape[0]
    eng, _ = make_engine([g[f"param0_{i}"] for i in range(8)], g["batch"], n, 1, chain_assoc="flops")
    assert eng.chain_flops
    eng.set_epoch_batches(torch.arange(n).view(1, n))
    for step in range(2):
        eng.set_noise(torch.as_tensor(g["noise"]))
        eng.step()
        assert abs(float(eng.loss) - float(g[f"loss{step}"])) < 2e-5
        for i in range(8):
            ref = g[f"grad{step}_{i}"]
            np.testing.assert_allclose(eng.grad_view(i).numpy(), ref, rtol=0, atol=1e-3 * max(np.abs(ref).max(), 1e-8))
            np.testing.assert_allclose(eng.fp.view(eng.fp.flat, i).numpy(), g[f"param{step + 1}_{i}"], rtol=0, atol=5e-6)


def test_engine_fused_update_path_equals_separate_optimiser_launch():
    """fuse_update=True (Adadelta in the epilogue of the last chain launch, vgan_gemm_grouped_ex) against the default separate
    optimiser launch, on the CPU provider: same losses and parameters over three steps of fixture f2 (c2)."""
    g = load_golden("f2_step_c2.npz")
    res = {}
    for fused in (False, True):
        eng, _ = make_engine([g[f"param0_{i}"] for i in range(8)], g["batch"], 512, 1, fuse_update=fused)
        assert eng.fuse_update == fused
        eng.set_epoch_batches(torch.arange(512).view(1, 512))
        losses = []
        for _ in range(3):
            eng.set_noise(torch.as_tensor(g["noise"]))
            eng.step()
            losses.append(float(eng.loss))
        res[fused] = (losses, eng.fp.flat.clone(), eng.Wt_all.clone())
    assert res[True][0] == res[False][0]
    for a, b in zip(res[True][1:], res[False][1:]):
        assert torch.equal(a, b)


def test_xx_tiles_riding_in_the_forward_launch_equal_the_gram_launch(monkeypatch):
    """bf16x3 mode, fused forward: the X-X tiles computed as a job of the mask / projection launch -- from the data set's split
    images through the batch index table (csrc/mmd_xx.hpp) -- give the same losses and parameters as the same tiles inside the
    Gram launch, across an epoch boundary and a dropped remainder (CPU provider)."""
    g = load_golden("f3_traj_c1.npz")
    res = {}
    for ride in ("1", "0"):
        monkeypatch.setenv("VGAN_XX_RIDE", ride)
        eng, _ = make_engine([g[f"param0_{i}"] for i in range(8)], g["data"], 128, 10, mmd_precision="bf16x3")
        assert eng.xx_ride == (ride == "1") and eng.fused_prepare
        losses = []
        for t in range(14):
            if t % 10 == 0:
                eng.set_epoch_batches(torch.as_tensor(g["idx"][t:t + 10].astype(np.int64)))
            eng.set_noise(torch.as_tensor(g["noise"][t]))
            eng.step()
            losses.append(float(eng.loss))
        res[ride] = (np.array(losses), eng.fp.flat.clone())
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(res["1"][0], g["losses"][:14], rtol=0, atol=1e-4)
    assert torch.equal(res["1"][1], res["0"][1])  # the X-X sums feed the reported loss only, never a gradient


@pytest.mark.parametrize("carrier", ["backward", "m4"])
def test_xx_tiles_in_the_m4_launch_with_split_step_tail(monkeypatch, carrier):
    """bf16x3 mode: the X-X tiles computed inside the M_4 launch (two launches after the MMD backward that carries the step
    tail) with the tail split in two -- everything but the X-X block sum early, the X-X sum and the loss in the first chain
    launch of the backward -- give the same per-step losses, statistics and parameters as the unsplit schedule (CPU provider,
    across an epoch boundary)."""
    g = load_golden("f3_traj_c1.npz")
    rng = np.random.default_rng(6)
    n, nb = 256, 5   # the M_4 launch that carries the tiles is the 16-wave tall-skinny kernel: batch >= 256
    idx = np.stack([rng.permutation(1280)[:n] for _ in range(2 * nb)])
    noise = rng.normal(size=(2 * nb, n, 1)).astype(np.float32)
    ref = orc.NoKLTrainer([g[f"param0_{i}"].astype(np.float64) for i in range(8)])
    want = [ref.step(g["data"][idx[t]].astype(np.float64), noise[t].astype(np.float64))["loss"] for t in range(7)]
    res = {}
    # 16 XY + 10 YY + 10 X-X tiles: the Gram launch is given 30 slots -- 4 X-X tiles early, 6 behind M_4 (on the GPU it has
    # 512, and at this size everything would fit the one launch)
    monkeypatch.setenv("VGAN_GRAM_SLOTS", "30")
    monkeypatch.setenv("VGAN_XX_LATE", carrier)   # the late tiles ride in the MMD backward launch (default) or behind M_4
    for late in ("1", "0"):
        monkeypatch.setenv("VGAN_XX_IN_M4", late)
        eng, _ = make_engine([g[f"param0_{i}"] for i in range(8)], g["data"], n, nb, mmd_precision="bf16x3")
        assert eng.xx_in_m4 == (late == "1") and (eng.n_main, eng.tiles.shape[0]) == ((30, 36) if late == "1" else (36, 36))
        assert eng.xx_late_in_backward == (late == "1" and carrier == "backward")
        losses, sxx = [], []
        for t in range(7):
            if t % nb == 0:
                eng.set_epoch_batches(torch.as_tensor(idx[t:t + nb].astype(np.int64)))
            eng.set_noise(torch.as_tensor(noise[t]))
            eng.step()
            losses.append(float(eng.loss))
            sxx.append(float(eng.stats[0]))
        res[late] = (np.array(losses), np.array(sxx), eng.fp.flat.clone(), float(eng.loss_accum))
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(res["1"][1], res["0"][1], rtol=1e-6)
    np.testing.assert_allclose(res["1"][0], want, rtol=0, atol=1e-4)
    assert torch.equal(res["1"][2], res["0"][2]) and abs(res["1"][3] - res["0"][3]) < 1e-5


def test_engine_bf16x3_precision_mode_vs_reference_fixture():
    """Split-bf16 MMD mode (emulated on the CPU stand-in with torch.bfloat16 roundings): the step still meets the loss bar."""
    g = load_golden("f2_step_c2.npz")
    n = g["batch"].shape[0]
    eng, _ = make_engine([g[f"param0_{i}"] for i in range(8)], g["batch"], n, 1, mmd_precision="bf16x3")
    eng.set_epoch_batches(torch.arange(n).view(1, n))
    for step in range(2):
        eng.set_noise(torch.as_tensor(g["noise"]))
        eng.step()
        assert abs(float(eng.loss) - float(g[f"loss{step}"])) < 1e-4
        for i in range(8):
            ref = g[f"grad{step}_{i}"]
            np.testing.assert_allclose(eng.grad_view(i).numpy(), ref, rtol=0, atol=2e-3 * max(np.abs(ref).max(), 1e-8))


def test_fit_loop_reproduces_reference_run_on_cpu_provider():
    """The whole VGAN_no_kl.fit host loop (seeding, init order, shuffles, noise draws, epoch means, shared-RBF
    bandwidth hand-over, sampling) against the reference's own run (fixture f3), kernels emulated by CpuOps."""
    from src.vgan import VGAN_no_kl
    from src.models.Mmd_loss_constrained import MMDLossConstrained
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None
    g = load_golden("f3_traj_c1.npz")
    model = VGAN_no_kl(batch_size=128, epochs=20, seed=777)
    model._ops_override = CpuOps()
    model.device = torch.device("cpu")
    model.noise_source = "host"
    model.verbose = False
    model.fit(g["data"])
    np.testing.assert_allclose(model.train_history["generator_loss"], g["epoch_losses"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(float(model.bandwidth), float(g["bw"]), rtol=1e-5)
    assert np.array_equal(model.generate_subspaces(500).numpy(), g["masks"])
    for i, q in enumerate(model.generator.parameters()):
        np.testing.assert_allclose(q.detach().numpy(), g[f"paramT_{i}"], rtol=0, atol=5e-5)
    # reference quirk kept: the process-wide default RBF now carries this run's bandwidth (Mmd_loss_constrained.py:35)
    assert float(MMDLossConstrained(weight=1).kernel.bandwidth) == pytest.approx(float(g["bw"]), rel=1e-5)
    model.approx_subspace_dist()
    assert model.subspaces.shape[1] == 20 and abs(model.proba.sum() - 1) < 1e-12
    with pytest.raises(AssertionError):  # the reference's only explicit check (src/vgan.py:398)
        model.check_if_myopic(g["data"][:100], count=500)


def test_batch_size_clamp_and_history_shape():
    from src.vgan import VGAN_no_kl
    from src.models.Mmd_loss_constrained import MMDLossConstrained
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None
    X = orc.synthetic_dataset("c1", rows=96)
    model = VGAN_no_kl(batch_size=500, epochs=3, seed=1)
    model._ops_override, model.device, model.verbose, model.noise_source = CpuOps(), torch.device("cpu"), False, "host"
    assert model.fit(X) is None
    assert model.batch_size == 96 and len(model.train_history["generator_loss"]) == 3
    assert model.get_params()["generator optimizer"] == "Adadelta"
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None


def test_device_shuffle_is_a_permutation_and_drives_fit():
    """vgan_shuffle_epoch's counter-based permutation (host evaluation, vgan_shuffle_index): a bijection of [0, N) for sizes
    that are and are not powers of four, different per epoch and per seed; a fit with shuffle_source='device' consumes it
    (every epoch's batches are disjoint rows of the data set) and draws nothing from torch's generator for the shuffle."""
    ops = CpuOps()
    for N in (1, 2, 7, 64, 1000, 4096, 5001):
        p = [ops.shuffle_index(i, N, 777, 3) for i in range(N)]
        assert sorted(p) == list(range(N)), N
    a = [ops.shuffle_index(i, 5001, 777, 3) for i in range(5001)]
    b = [ops.shuffle_index(i, 5001, 777, 4) for i in range(5001)]
    c = [ops.shuffle_index(i, 5001, 778, 3) for i in range(5001)]
    assert sum(x == y for x, y in zip(a, b)) < 20 and sum(x == y for x, y in zip(a, c)) < 20
    assert abs(np.corrcoef(a, np.arange(5001))[0, 1]) < 0.05
    assert ops.shuffle_index(5, 5, 1, 1) == -1 and ops.shuffle_index(-1, 5, 1, 1) == -1

    from src.vgan import VGAN_no_kl
    from src.models.Mmd_loss_constrained import MMDLossConstrained
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None
    X = orc.synthetic_dataset("c1", rows=300)
    model = VGAN_no_kl(batch_size=128, epochs=2, seed=9)
    model._ops_override, model.device, model.verbose, model.noise_source = ops, torch.device("cpu"), False, "host"
    model.shuffle_source = "device"
    model.fit(X)
    perm = model._engine.perm.numpy()                     # the last epoch's table: 2 batches of 128 distinct rows of 300
    assert perm.shape == (2, 128) and len(set(perm.ravel().tolist())) == 256 and perm.min() >= 0 and perm.max() < 300
    assert perm.ravel().tolist() == [ops.shuffle_index(i, 300, 9, 1) for i in range(256)]
    assert np.isfinite(model.train_history["generator_loss"]).all()
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None


def _myopic_reference(model, data, bandwidths, count, n_perm):
    """p-values of oracle.two_sample_pvalue on exactly the samples / permutations check_if_myopic draws."""
    x = np.asarray(data, dtype=np.float64)
    norms = np.sqrt((x * x).sum(axis=0))
    x = x / np.where(norms == 0.0, 1.0, norms)
    rng = np.random.default_rng(model.seed)
    rows = rng.choice(x.shape[0], size=count, replace=False)
    xs = x[rows].astype(np.float32)
    u = model.generate_subspaces(count).cpu().numpy()
    ux = np.where(u, xs, xs.mean(axis=0, keepdims=True, dtype=np.float32)).astype(np.float32)
    m = 2 * count
    assign = np.zeros((n_perm, m), dtype=bool)
    for q in range(n_perm):
        assign[q, rng.permutation(m)[:count]] = True
    out = []
    for alpha in sorted(bandwidths) + [float(model.bandwidth)]:
        K = orc.two_sample_kernel_matrix(xs, ux, alpha)
        out.append(orc.two_sample_pvalue(K, count, assign)[0])
    return out


def test_check_if_myopic_matches_stated_definition_on_cpu_provider():
    """check_if_myopic (src/vgan.py:384-431): host logic + provider calls against the oracle's restatement of
    torch-two-sample's statistic and permutation p-value (parity unpinned against that absent dependency)."""
    from src.vgan import VGAN_no_kl
    from src.models.Mmd_loss_constrained import MMDLossConstrained
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None
    g = load_golden("f3_traj_c1.npz")
    model = VGAN_no_kl(batch_size=128, epochs=2, seed=777)
    model._ops_override = CpuOps()
    model.device = torch.device("cpu")
    model.noise_source = "host"
    model.verbose = False
    model.fit(g["data"])
    df = model.check_if_myopic(g["data"], bandwidth=[1.0, 0.01], count=90, n_permutations=120)
    assert list(df.columns) == [0.01, 1.0, "recommended bandwidth"] and list(df.index) == ["p-val"]
    want = _myopic_reference(model, g["data"], [1.0, 0.01], 90, 120)
    np.testing.assert_allclose(df.to_numpy()[0].astype(float), want, rtol=0, atol=1.0 / 120 + 1e-12)
    # a sample against itself shifted far away is rejected; against an identical copy it is not
    K = orc.two_sample_kernel_matrix(g["data"][:60], g["data"][:60] + 3.0, 0.05)
    rng = np.random.default_rng(0)
    assign = np.zeros((200, 120), dtype=bool)
    for q in range(200):
        assign[q, rng.permutation(120)[:60]] = True
    assert orc.two_sample_pvalue(K, 60, assign)[0] == 0.0
    K = orc.two_sample_kernel_matrix(g["data"][:60], g["data"][60:120], 0.05)
    assert orc.two_sample_pvalue(K, 60, assign)[0] > 0.05


def test_vgan_kl_fit_reproduces_reference_run_on_cpu_provider():
    """VGAN.fit (kernel learning; step engine of v-gan_amd/kl_trainer.py, explicit forward/backward, no autograd) against
    the reference's own 12-epoch run (fixture f4): both loss histories, the bandwidth, the never-trained generator, the
    detector parameters and their requires_grad flags (encoder-freeze quirk)."""
    from src.vgan import VGAN
    from src.models.Mmd_loss_constrained import MMDLossConstrained
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None
    g = load_golden("f4_kl_c1.npz")
    model = VGAN(batch_size=128, epochs=12)
    model.noise_source = "host"
    model._ops_override = CpuOps()
    model.device = torch.device("cpu")
    model.verbose = False
    model.fit(g["data"])
    gl, dl = np.array(model.train_history["generator_loss"]), np.array(model.train_history["detector_loss"])
    assert np.isnan(gl[0]) and np.isnan(g["generator_loss"][0])
    np.testing.assert_allclose(gl[1:], g["generator_loss"][1:], rtol=0, atol=5e-3)
    np.testing.assert_allclose(dl, g["detector_loss"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(float(model.bandwidth), float(g["bw"]), rtol=1e-3)
    for i, q in enumerate(model.generator.parameters()):
        assert np.array_equal(q.detach().numpy(), g[f"genT_{i}"])
    for i, q in enumerate(model.detector.parameters()):
        np.testing.assert_allclose(q.detach().numpy(), g[f"detT_{i}"], rtol=0, atol=1e-3)
        assert bool(q.requires_grad) == bool(g[f"detT_rg_{i}"])
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None


def test_kl_resident_feed_equals_per_step_feed_cpu_provider():
    kl_resident_feed_equals_per_step_feed(CpuOps(), torch.device("cpu"))


def test_vgan_kl_fit_device_feed_runs_on_cpu_provider():
    """VGAN.fit with the noise and the shuffle both drawn on the device (the default noise source; shuffle_source="device"):
    finite histories of the right length, the generator untouched, and the same run twice gives the same numbers."""
    from src.vgan import VGAN
    from src.models.Mmd_loss_constrained import MMDLossConstrained
    hist = []
    for _ in range(2):
        MMDLossConstrained.__init__.__defaults__[0].bandwidth = None
        model = VGAN(batch_size=64, epochs=7)
        model._ops_override, model.device, model.verbose = CpuOps(), torch.device("cpu"), False
        model.shuffle_source = "device"
        model.fit(orc.synthetic_dataset("c1", rows=64 * 4 + 9))
        hist.append((list(model.train_history["detector_loss"]), list(model.train_history["generator_loss"])))
        assert len(hist[-1][0]) == 7 and np.isfinite(hist[-1][0]).all() and np.isfinite(hist[-1][1][1:]).all()
    assert hist[0][0] == hist[1][0] and hist[0][1][1:] == hist[1][1][1:]
    MMDLossConstrained.__init__.__defaults__[0].bandwidth = None


def test_run_steps_needs_the_device_noise_stream():
    """Several steps per call are only defined when the engine draws its own noise; with host-provided noise the per-step
    feed is the contract (set_noise + step), and run_steps(1) is that single step."""
    g = load_golden("f2_step_c1.npz")
    eng, _ = make_engine([g[f"param0_{i}"] for i in range(8)], g["batch"], 128, 1)
    eng.set_epoch_batches(torch.arange(128).view(1, 128))
    eng.set_noise(torch.as_tensor(g["noise"]))
    eng.run_steps(1)
    assert eng.steps_done == 1
    with pytest.raises(ValueError):
        eng.run_steps(2)


def test_epoch_table_staged_behind_the_running_epoch():
    """stage_epoch_batches + begin_epoch = set_epoch_batches; a table staged while an epoch runs does not disturb that epoch's
    steps, and begin_epoch without a staged table is an error (the fit and bench.py stage the next epoch's table right after
    launching the current epoch's steps)."""
    g = load_golden("f2_step_c1.npz")
    n = g["batch"].shape[0]
    data = np.concatenate([g["batch"], g["batch"][::-1]])  # two batches per epoch
    perm_a = torch.arange(2 * n).view(2, n)
    perm_b = torch.arange(2 * n).flip(0).view(2, n)
    losses = []
    for staged in (False, True):
        eng, _ = make_engine([g[f"param0_{i}"] for i in range(8)], data, n, 2)
        out = []
        if staged:
            eng.stage_epoch_batches(perm_a)
            eng.begin_epoch()
            assert not eng.epoch_staged
        else:
            eng.set_epoch_batches(perm_a)
        for step in range(4):
            if step == 1 and staged:
                eng.stage_epoch_batches(perm_b)  # mid-epoch: the running epoch keeps reading its own table
                assert eng.epoch_staged and torch.equal(eng.perm, perm_a.to(torch.int32))
            if step == 2:
                if staged:
                    eng.begin_epoch()
                else:
                    eng.set_epoch_batches(perm_b)
                assert torch.equal(eng.perm, perm_b.to(torch.int32))
            eng.set_noise(torch.as_tensor(g["noise"]))
            eng.step()
            out.append(float(eng.loss))
        losses.append(out)
        if staged:
            with pytest.raises(RuntimeError):
                eng.begin_epoch()
    assert losses[0] == losses[1]


def test_gram_launch_round_model_and_boundary():
    """_launch_rounds / _best_boundary (the boundary between the two Gram launches of the sharded front): the modelled time grows
    with the tile count inside a round, a K-split remainder is never dearer than a whole one, the boundary only moves down, never
    to zero, never to a dearer place, and stays put unless the model gains at least 0.15 round."""
    from vgan_amd.trainer import _best_boundary, _launch_rounds
    for slots in (256, 512, 1024):
        for can_split in (False, True):
            prev = 0.0
            for t in range(1, 3 * slots + 1):
                c = _launch_rounds(t, slots, can_split)
                if t % slots != 1 and not can_split:
                    assert c >= prev - 1e-12
                assert c <= _launch_rounds(t, slots, False) + 1e-12
                assert (t + slots - 1) // slots - 0.75 - 1e-9 <= c <= (t + slots - 1) // slots
                prev = c
    rng = np.random.default_rng(3)
    for _ in range(300):
        slots = int(rng.choice([256, 512, 1024]))
        total = int(rng.integers(2, 6 * slots))
        n_main = int(rng.integers(1, total))
        can_split = bool(rng.integers(0, 2))
        cost = lambda k: _launch_rounds(k, slots, can_split) + _launch_rounds(total - k, slots, can_split)
        k = _best_boundary(n_main, total, slots, can_split)
        assert 1 <= k <= n_main
        assert k == n_main or cost(k) <= cost(n_main) - 0.15 + 1e-9
    assert _best_boundary(772, 1256, 256, False) == 768   # c5, 8 ranks, 128-wide tiles: the four stragglers go over
    assert _best_boundary(388, 632, 256, True) == 388     # c5, 8 ranks, wide tiles: nothing to gain
