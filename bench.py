#!/usr/bin/env python3
"""V-GAN training-step benchmark (BASELINE.json metric: train steps/sec, batch=1024, d=784).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one VGAN_no_kl.fit step body (reference src/vgan.py:597-621) at the GLOBAL batch of
1024 rows, d=784, L=49, on synthetic MNIST-like data resident in HBM: Philox noise -> Generator_big
-> upper_softmax -> U*X -> 5-bandwidth RBF MMD^2 (+penalty) -> backward -> Adadelta.  N > 1 shards
the batch rows across ranks (exact data parallel; strong scaling: total work fixed).

Rank 0 prints ONE JSON line.  Besides the driver's contract fields it carries
  roofline      the Gram/MMD kernel against the fp32-MFMA peak (durations measured here with HIP events)
  cpu_baseline  the op-for-op PyTorch-CPU port of the reference step timed on this host's cores
  parity        |loss_gpu - loss_cpu| on identical inputs (bar 1e-4)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_BATCH, D_FEAT, EPOCH_BATCHES = 1024, 784, 16
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
WORKLOAD = "configs[2]: MNIST-pixels-as-features stand-in, d=784, batch=1024 (VGAN_no_kl step, fp32)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed steps run right after graph capture so that the GPU clock has ramped (DVFS) before the "
                         "W warm-up steps; a fit runs for minutes, so the ramped state is the representative one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def build_engine(rank, world, use_graph, **engine_kw):
    import vgan_amd
    from vgan_amd import synth
    from vgan_amd.ops import HipOps
    from vgan_amd.trainer import NoKLStepEngine
    data = synth.synthetic_dataset("c3")  # [16*1024, 784] float32
    params = synth.synthetic_generator_params(D_FEAT)
    gen = vgan_amd.Generator_big(synth.latent_size(D_FEAT), D_FEAT)
    with torch.no_grad():
        for q, v in zip(gen.parameters(), params):
            q.copy_(torch.as_tensor(v))
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = NoKLStepEngine(HipOps(), gen.to(dev), torch.as_tensor(data).to(dev), N_BATCH, EPOCH_BATCHES, lr=0.007,
                         weight_decay=0.04, penalty_weight=10.0, seed=777, noise="device", rank=rank, world=world,
                         use_graph=use_graph, **engine_kw)
    return eng, data, params


def run_steps(eng, count, start_step):
    from vgan_amd.vgan import epoch_batches
    for t in range(start_step, start_step + count):
        if t % EPOCH_BATCHES == 0:  # new shuffled epoch, as fit() does
            eng.set_epoch_batches(epoch_batches(eng.data.shape[0], N_BATCH))
        eng.step()


def time_kernel(fn, iters=30):
    """Average duration (ms) of one launch, HIP events on the stream the kernel is launched on."""
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_rooflines(eng):
    """Per-launch durations of the two MFMA kernels of the MMD, as launched inside the step."""
    ops, n, p = eng.ops, eng.n, eng.dp
    nl = eng.nl
    t_gram = time_kernel(lambda: ops.mmd_gram(eng.Z, eng.sqn, n, p, eng.bw, eng.tiles, False, eng.Wg, n + eng.lo, eng.partial))
    t_bwd = time_kernel(lambda: ops.mmd_backward(eng.Wg, eng.Z, n + eng.lo, nl, 2 * n, p, eng.Z[eng.lo:eng.lo + nl], eng.gU))
    # algorithmic FLOPs: forward 2n^2 unique pairs x 2p = 4 n^2 p ; backward Gs[n x 2n] . Z[2n x p] = 4 n^2 p
    f_gram = 4.0 * n * n * D_FEAT / eng.world
    f_bwd = 4.0 * n * n * D_FEAT / eng.world
    return {
        "mmd_gram": {"ms": t_gram, "tflops": f_gram / (t_gram * 1e-3) / 1e12, "flop": f_gram},
        "mmd_backward": {"ms": t_bwd, "tflops": f_bwd / (t_bwd * 1e-3) / 1e12, "flop": f_bwd},
    }


def usable_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU boxes expose
    256 logical CPUs but grant a 16-CPU quota; oversubscribing them makes the baseline ~10x slower than it is)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(data, params, seconds):
    """The reference's CPU path cannot travel to this box; its op-for-op PyTorch port (oracle/torch_port.py,
    pinned to the reference by tests/test_oracle_golden.py) is timed instead on a bounded sample."""
    from oracle import torch_port as port
    cores = usable_cores()
    torch.set_num_threads(cores)
    tr = port.PortNoKL(params)
    rng = np.random.default_rng(0)
    L = params[0].shape[1]
    X = torch.as_tensor(data[:N_BATCH])
    z = torch.as_tensor(rng.normal(size=(N_BATCH, L)).astype(np.float32))
    first_loss = tr.step(X, z)  # warm-up + bandwidth calibration (also the parity probe)
    tr.step(X, z)
    t0 = time.perf_counter()
    k = 0
    while k < 3 or (time.perf_counter() - t0 < seconds and k < 200):
        idx = rng.permutation(data.shape[0])[:N_BATCH]
        tr.step(torch.as_tensor(data[idx]), torch.as_tensor(rng.normal(size=(N_BATCH, L)).astype(np.float32)))
        k += 1
    dt = time.perf_counter() - t0
    return {"value": k / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{k} VGAN_no_kl steps (batch=1024, d=784, fp32) of the PyTorch-CPU port in {dt:.1f} s"}, first_loss, (X, z)


def gpu_first_loss(params, X, z):
    """Loss of the first step on the same params / batch / noise as the CPU probe."""
    import vgan_amd
    from vgan_amd.ops import HipOps
    from vgan_amd.trainer import NoKLStepEngine
    gen = vgan_amd.Generator_big(params[0].shape[1], D_FEAT)
    with torch.no_grad():
        for q, v in zip(gen.parameters(), params):
            q.copy_(torch.as_tensor(v))
    eng = NoKLStepEngine(HipOps(), gen.cuda(), X.cuda(), N_BATCH, 1, noise="host", use_graph=False, loss_accum_scale=1.0)
    eng.set_epoch_batches(torch.arange(N_BATCH).view(1, -1))
    eng.set_noise(z)
    eng.step()
    return float(eng.loss)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    torch.manual_seed(1234)
    use_graph = not args.no_graph
    eng, data, params = build_engine(rank, world, use_graph)
    def prewarm(e):
        t_end, k = time.perf_counter() + args.prewarm_seconds, 0
        while time.perf_counter() < t_end:
            run_steps(e, EPOCH_BATCHES, k)
            k += EPOCH_BATCHES
            torch.cuda.synchronize()

    try:
        prewarm(eng)
        run_steps(eng, args.warmup, 0)
        torch.cuda.synchronize()
    except Exception as e:  # a collective that cannot be captured: fall back to eager launches
        if not use_graph:
            raise
        print(f"[bench] HIP-graph path failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
        use_graph = False
        eng, data, params = build_engine(rank, world, False)
        prewarm(eng)
        run_steps(eng, args.warmup, 0)
        torch.cuda.synchronize()
    eng.epoch_loss()

    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(eng, args.steps, args.warmup)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mean_loss = eng.epoch_loss() * EPOCH_BATCHES / max(args.steps, 1)

    kern = kernel_rooflines(eng)
    if rank == 0:
        steps_per_s = args.steps / elapsed
        out = {
            "metric": "V-GAN train steps/sec (batch=1024, d=784)", "value": steps_per_s, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD, "global_batch": N_BATCH, "features": D_FEAT, "latent": eng.L,
                       "rows_per_gpu": eng.nl, "parallelism": f"dp{world} (row-sharded Gram, replicated generator)",
                       "hip_graph": bool(use_graph), "mean_loss": mean_loss, "prewarm_s": args.prewarm_seconds,
                       "generator": eng.mode},
        }
        g = kern["mmd_gram"]
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and world == 1:
            try:
                traffic = json.load(open(tp)).get("mmd_gram_kernel", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": "mmd_gram_kernel<4,false>", "achieved": g["tflops"],
                           "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": g["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                           "traffic": traffic, "avg_launch_ms": g["ms"], "algorithmic_flop_per_launch": g["flop"],
                           "also": {"mmd_backward_kernel<4>": {"achieved": kern["mmd_backward"]["tflops"],
                                                              "frac": kern["mmd_backward"]["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                                                              "avg_launch_ms": kern["mmd_backward"]["ms"]}},
                           "step_frac": (8.0 * N_BATCH * N_BATCH * D_FEAT + 6.0 * N_BATCH * (eng.fp.total)) * steps_per_s
                           / world / (FP32_MFMA_PEAK_TFLOPS * 1e12)}
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_loss, (X, z) = cpu_baseline(data, params, args.cpu_seconds)
            out["cpu_baseline"] = cb
            gl = gpu_first_loss(params, X, z)
            out["parity"] = {"loss_gpu": gl, "loss_cpu_port": cpu_loss, "abs_diff": abs(gl - cpu_loss), "bar": 1e-4}
            out["speedup_vs_cpu"] = steps_per_s / cb["value"]
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
