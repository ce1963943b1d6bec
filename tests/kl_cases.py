"""Cases of VGAN.fit's step engine (v-gan_amd/kl_trainer.py) shared by the CPU-provider tier (tests/test_host_logic.py) and the
GPU tier (tests/test_hip_parity.py): the same assertions over either kernel provider."""
import numpy as np
import pytest
import torch

from oracle import vgan_oracle as orc


def _kl_engine(ops, device, nb, noise, rows, n=64, d=36, use_graph=True):
    from vgan_amd.kl_trainer import KLStepEngine
    from vgan_amd.modules import Decoder, Detector, Encoder, Generator_big
    L = orc.latent_size(d)
    torch.manual_seed(11)
    gen, det = Generator_big(L, d), Detector(L, d, Encoder, Decoder)
    for mod in (gen, det):
        for q in mod.parameters():
            q.data.normal_(0.0, 0.1) if q.dim() == 2 else q.data.zero_()
    data = torch.as_tensor(orc.synthetic_dataset("c1", rows=rows)[:, :d] if d <= 20 else
                           np.random.default_rng(3).normal(size=(rows, d)).astype(np.float32))
    eng = KLStepEngine(ops, gen.to(device), det.to(device), data.to(device), n, 0.007, 0.04, 10.0, use_graph=use_graph,
                       batches_per_epoch=nb, noise=noise, seed=5)
    return eng, det


def kl_resident_feed_equals_per_step_feed(ops, device):
    """The device-resident epoch table walked by the step counter (VGAN.fit's feed) gives the steps the per-step feed gives:
    two epochs of three batches, one step kind after the other, detector parameters bit for bit."""
    n, nb, rows = 64, 3, 64 * 5
    a, det_a = _kl_engine(ops, device, nb, "host", rows)
    b, det_b = _kl_engine(ops, device, 1, "host", rows)
    rng = np.random.default_rng(21)
    kinds = [("d", True), ("d", True), ("d", True), ("g", False), ("g", False), ("g", False), ("d", False), ("d", False), ("d", False)]
    k = 0
    for epoch in range(3):
        table = torch.as_tensor(rng.permutation(rows)[:nb * n].reshape(nb, n))
        a.set_epoch_batches(table)
        for t in range(nb):
            z = torch.as_tensor(rng.normal(size=(n, a.L)).astype(np.float32))
            kind, enc = kinds[k]
            k += 1
            if kind == "d":
                a.detector_step(noise=z, train_encoder=enc)
                b.detector_step(table[t], z, train_encoder=enc)
            else:
                a.generator_phase_step(noise=z)
                b.generator_phase_step(table[t], z)
            assert a.epoch_sums() == b.epoch_sums()
    assert int(a.step_counter.item()) == 9
    for qa, qb in zip(det_a.parameters(), det_b.parameters()):
        assert torch.equal(qa.detach().cpu(), qb.detach().cpu())
    with pytest.raises(ValueError):
        a.detector_step(table[0], z)               # per-step indices need a one-row table
    c, _ = _kl_engine(ops, device, nb, "device", rows)
    with pytest.raises(ValueError):
        c.generator_phase_step(noise=z)            # the engine draws its own noise
    # device feed: shuffle + noise on the device; the same (seed, epoch, step) gives the same steps again
    out = []
    for _ in range(2):
        c, _ = _kl_engine(ops, device, nb, "device", rows)
        sums = []
        for epoch in range(2):
            c.shuffle_epoch(epoch)
            for t in range(nb):
                c.detector_step(train_encoder=True) if epoch == 0 else c.generator_phase_step()
            sums.append(c.epoch_sums())
        out.append(sums)
    assert out[0] == out[1] and all(np.isfinite(v) for s in out[0] for v in s)
    # ... and an epoch issued as ONE call (blocks of steps per graph launch on the GPU) equals the same epoch step by step
    c, det_c = _kl_engine(ops, device, nb, "device", rows)
    sums = []
    for epoch in range(2):
        c.shuffle_epoch(epoch)
        c.detector_step(train_encoder=True, count=nb) if epoch == 0 else c.generator_phase_step(count=nb)
        sums.append(c.epoch_sums())
    assert sums == out[0] and int(c.step_counter.item()) == 2 * nb
    with pytest.raises(ValueError):
        c.generator_phase_step(noise=z, count=2)
