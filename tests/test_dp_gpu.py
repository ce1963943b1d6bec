"""Row-sharded data parallel on the REAL kernels: two processes share the one GPU of the test box (gloo moves the
all-reduce through the host, so no graph capture here) and must reproduce the single-process losses of the same
engine on the same inputs, stay bit-identical to each other, and match the reference trajectory fixture."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, steps, out_dir, precision, n, d_case, front=None):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import vgan_oracle as orc
    from vgan_amd.modules import Generator_big
    from vgan_amd.ops import HipOps
    from vgan_amd.trainer import NoKLStepEngine
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    if d_case == "c1":
        g = load_golden("f3_traj_c1.npz")
        data, params = g["data"], [g[f"param0_{i}"] for i in range(8)]
        idx = g["idx"][:steps].astype(np.int64)
        noise = g["noise"][:steps]
        kw = dict(lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    else:
        d = 784 if d_case == "c3" else 2048
        rng = np.random.default_rng(3)
        data = orc.synthetic_dataset(d_case, rows=4 * n)
        params = orc.synthetic_generator_params(d)
        idx = np.stack([rng.permutation(4 * n)[:n] for _ in range(steps)])
        noise = rng.normal(size=(steps, n, orc.latent_size(d))).astype(np.float32)
        kw = {}
    d = data.shape[1]
    gen = Generator_big(orc.latent_size(d), d)
    with torch.no_grad():
        for q, v in zip(gen.parameters(), params):
            q.copy_(torch.as_tensor(v))
    eng = NoKLStepEngine(HipOps(), gen.cuda(), torch.as_tensor(data).cuda(), n, 1, noise="host", use_graph=False, loss_accum_scale=1.0,
                         rank=rank, world=world, mmd_precision=precision, front=front if world > 1 else None, **kw)
    assert eng.front_sharded == (world > 1 and (front == "sharded" or (front is None and n * d >= (1 << 22))))
    losses = []
    for t in range(steps):
        eng.set_epoch_batches(torch.as_tensor(idx[t:t + 1]))
        eng.set_noise(torch.as_tensor(noise[t]))
        eng.step()
        losses.append(eng.step_loss())
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"), losses=np.array(losses), flat=eng.fp.flat.cpu().numpy(), bw=float(eng.bw))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("case,n,precision,steps,front", [
    ("c1", 128, "fp32", 12, "replicated"), ("c3", 1024, "bf16x3", 4, "replicated"), ("c3", 512, "fp32", 3, "replicated"),
    # the sharded front (all-gather of the Y rows; trainer.py): small and metric-sized shapes forced, a c4-like shape
    # (d = 2048 > 1024: unfused operand split, 128-wide tiles, flop-minimal chain) by the engine's own size rule
    ("c1", 128, "fp32", 12, "sharded"), ("c3", 1024, "bf16x3", 4, "sharded"), ("c3", 512, "fp32", 3, "sharded"),
    ("c4", 2048, "bf16x3", 3, None), ("c4", 2048, "fp32", 2, None)])
def test_two_ranks_on_the_real_kernels(case, n, precision, steps, front, tmp_path):
    mp.spawn(_run, args=(1, 0, steps, str(tmp_path), precision, n, case, front), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), steps, str(tmp_path), precision, n, case, front), nprocs=2, join=True)
    one = np.load(tmp_path / "w1_r0.npz")
    r0, r1 = np.load(tmp_path / "w2_r0.npz"), np.load(tmp_path / "w2_r1.npz")
    assert np.array_equal(r0["flat"], r1["flat"])                        # replicas stay bit-identical
    assert np.array_equal(r0["losses"], r1["losses"])
    np.testing.assert_allclose(r0["losses"], one["losses"], rtol=0, atol=2e-6)   # same statistic as the unsharded engine
    np.testing.assert_allclose(r0["bw"], one["bw"], rtol=1e-6)
    np.testing.assert_allclose(r0["flat"], one["flat"], rtol=0, atol=2e-6)
    if case == "c1":
        g = load_golden("f3_traj_c1.npz")
        assert np.abs(r0["losses"] - g["losses"][:steps]).max() < 2e-5   # and of the reference's own run


def test_rccl_wrappers_of_the_c_abi_single_rank():
    """vgan_dp_*: the RCCL wrappers a non-torch caller of the C ABI uses for the step's one collective.  A one-GPU box can
    only form a single-member communicator: id -> communicator -> in-place all-reduce(SUM) on the current stream (also
    inside a side stream, as the overlapped schedule issues it) -> destroy."""
    from vgan_amd.ops import HipOps
    ops = HipOps()
    uid = ops.dp_unique_id()
    assert len(uid) == 128
    comm = ops.dp_comm_create(1, uid, 0)
    t = torch.arange(40000, dtype=torch.float32, device="cuda")
    want = t.clone()
    ops.dp_allreduce_sum(comm, t)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.dp_allreduce_sum(comm, t)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(t, want)
    # the sharded front's exchange: in-place all-gather of raw bytes (int16 split images, float norms, int64 keys)
    for tt in (torch.arange(4096, dtype=torch.int16, device="cuda").view(64, 64), torch.rand(1000, device="cuda"),
               torch.arange(77, dtype=torch.int64, device="cuda")):
        keep = tt.clone()
        ops.dp_allgather(comm, tt, 1)
        torch.cuda.synchronize()
        assert torch.equal(tt, keep)
    ops.dp_comm_destroy(comm)
