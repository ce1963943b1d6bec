#!/usr/bin/env python3
"""V-GAN training-step benchmark (BASELINE.json metric: train steps/sec, batch=1024, d=784).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one VGAN_no_kl.fit step body (reference src/vgan.py:597-621) at the GLOBAL batch of
1024 rows, d=784, L=49, on synthetic MNIST-like data resident in HBM: Philox noise -> Generator_big
-> upper_softmax -> U*X -> 5-bandwidth RBF MMD^2 (+penalty) -> backward -> Adadelta.  N > 1 shards
the batch rows across ranks (exact data parallel; strong scaling: total work fixed).

Timing: W untimed warm-up steps, then `--repeats` blocks of EXACTLY K steps, each bracketed by barrier +
torch.cuda.synchronize() on both sides and reduced by MAX over ranks; `value` is the MEDIAN block (every block's rate is kept
in `repeat_steps_per_s`).  Every timed block starts at an epoch boundary and one untimed block of the same K steps runs
first, so the blocks replay exactly the HIP graphs a fit replays (16 steps per graph launch + one graph for the remainder).

Rank 0 prints ONE JSON line.  Besides the driver's contract fields it carries
  roofline      the LONGEST MMD kernel of the active precision mode against the MFMA peak of the instruction it issues
                (durations measured here with HIP events); every other MMD kernel, both modes, under `also`
  fp32_mode     the same workload timed again with the fp32-MFMA kernels (the reference's own arithmetic is fp32)
  cpu_baseline  the op-for-op PyTorch-CPU port of the reference step timed on this host's cores (rank 0, any N)
  parity        |loss_gpu - loss_cpu| on identical inputs (bar 1e-4)
  extra         (default workload only) the larger BASELINE.json configurations in the same run: configs[3] (c4) and
                configs[4] (c5: fp32 as BASELINE specifies, and the engine's bf16x3 choice), and with N > 1 the alternative
                exchange schedules (c3: all-reduce overlapped with the next batch's X-X tiles; c4 / c5: replicated front
                instead of the sharded one) -- so that ONE invocation per N yields every number of the scaling study
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

N_BATCH, D_FEAT, EPOCH_BATCHES, CONFIG = 1024, 784, 16, "c3"
WORKLOADS = {  # --workload: the metric is quoted on c3; c4 / c5 are the larger BASELINE.json configurations
    "c1": (128, 20, 16, "configs[0]: 2-Gaussian mixture, d=20, batch=128 (VGAN_no_kl step; launch-latency bound)"),
    "c2": (512, 166, 5, "configs[1]: ADBench 'musk' stand-in, d=166, batch=512 (VGAN_no_kl step)"),
    "c3": (1024, 784, 16, "configs[2]: MNIST-pixels-as-features stand-in, d=784, batch=1024 (VGAN_no_kl step)"),
    "c4": (4096, 2048, 4, "configs[3]: synthetic tabular, d=2048, batch=4096 (VGAN_no_kl step)"),
    "c5": (8192, 4096, 4, "configs[4]: synthetic tabular, d=4096, batch=8192, 5-bandwidth RBF (VGAN_no_kl step)"),
}
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
WORKLOAD = WORKLOADS["c3"][3]
_DATA_CACHE = {}


def select_workload(cfg):
    global N_BATCH, D_FEAT, EPOCH_BATCHES, CONFIG, WORKLOAD
    CONFIG = cfg
    N_BATCH, D_FEAT, EPOCH_BATCHES, WORKLOAD = WORKLOADS[cfg]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed steps run right after graph capture so that the GPU clock has ramped (DVFS) before the "
                         "W warm-up steps; a fit runs for minutes, so the ramped state is the representative one")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--path", choices=["nokl", "kl"], default="nokl",
                    help="nokl: the metric's VGAN_no_kl step (default); kl: VGAN.fit's step mix (SURVEY 8a10), an extra "
                         "informational line, one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c4 / c5 / alternative-schedule legs of the default run")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--precision", choices=["auto", "fp32", "bf16x3"], default=None,
                    help="MMD contraction mode (default: VGAN_MMD_PRECISION or the engine's 'auto' rule, bf16x3 at this size)")
    ap.add_argument("--front", choices=["auto", "replicated", "sharded"], default=None,
                    help="data-parallel front of the step (trainer.py; default: the engine's size rule)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--optional-seconds", type=float, default=600.0,
                    help="budget of everything behind the metric's own leg (fp32 mode, c4 / c5 legs, CPU baseline): past it the line "
                         "is printed as it stands and the process ends")
    return ap.parse_args()


def build_engine(rank, world, use_graph, **engine_kw):
    import vgan_amd
    from vgan_amd import synth
    from vgan_amd.ops import HipOps
    from vgan_amd.trainer import NoKLStepEngine
    if CONFIG not in _DATA_CACHE:  # (c5: 537 MB of numpy draws -- once per process, not once per leg)
        _DATA_CACHE.clear()
        _DATA_CACHE[CONFIG] = synth.synthetic_dataset(CONFIG)  # [EPOCH_BATCHES * batch, d] float32
    data = _DATA_CACHE[CONFIG]
    params = synth.synthetic_generator_params(D_FEAT)
    gen = vgan_amd.Generator_big(synth.latent_size(D_FEAT), D_FEAT)
    with torch.no_grad():
        for q, v in zip(gen.parameters(), params):
            q.copy_(torch.as_tensor(v))
    dev = torch.device("cuda", torch.cuda.current_device())
    noise = engine_kw.pop("noise", "device")  # (tools/loss_curve.py feeds host-drawn noise to compare with the CPU port)
    eng = NoKLStepEngine(HipOps(), gen.to(dev), torch.as_tensor(data).to(dev), N_BATCH, EPOCH_BATCHES, lr=0.007,
                         weight_decay=0.04, penalty_weight=10.0, seed=777, noise=noise, rank=rank, world=world,
                         use_graph=use_graph, **engine_kw)
    return eng, data, params


def run_steps(eng, count, start_step):
    """`count` steps from global step index `start_step`, epoch by epoch as fit() runs them: a new shuffled table at every
    epoch start, then the epoch's steps through the engine's run_steps (blocks of steps per graph launch)."""
    from vgan_amd.vgan import epoch_batches
    t, end = start_step, start_step + count
    while t < end:
        if t % EPOCH_BATCHES == 0 and os.environ.get("VGAN_FEED_DIRECT") == "1":
            eng.set_epoch_batches(epoch_batches(eng.data.shape[0], N_BATCH))
        elif t % EPOCH_BATCHES == 0:  # new shuffled epoch, as fit() does: the table was drawn and uploaded behind the previous steps
            if not eng.epoch_staged:
                eng.stage_epoch_batches(epoch_batches(eng.data.shape[0], N_BATCH))
            eng.begin_epoch()
        k = min(end - t, EPOCH_BATCHES - t % EPOCH_BATCHES)
        eng.run_steps(k)
        t += k
        if not eng.epoch_staged and os.environ.get("VGAN_FEED_DIRECT") != "1":  # the next epoch's table, while the GPU works through these steps (NoKLStepEngine.stage_epoch_batches)
            eng.stage_epoch_batches(epoch_batches(eng.data.shape[0], N_BATCH))


def time_kernel(fn, iters=30, repeats=5):
    """Average duration (ms) of one launch, HIP events on the stream the kernel is launched on: `repeats` blocks of `iters`
    back-to-back launches, the median block (single blocks scatter by ~5 % with the clock state they start in)."""
    for _ in range(5):
        fn()
    blocks = []
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        blocks.append(e0.elapsed_time(e1) / iters)
    return sorted(blocks)[len(blocks) // 2]


def kernel_rooflines(eng):
    """Per-launch durations of the MFMA kernels of the MMD, as launched inside the step (both precision modes).
    ALGORITHMIC flop per launch: 2 d per pair the launch covers.  One GPU: the Gram covers the 2 n^2 unique pairs
    (XY block + upper triangles of XX and YY), the backward product W[n x 2n] . Z[2n x d] is 4 n^2 d.  A rank of a row-sharded
    run covers (own Y rows) x (all columns) -- the symmetric YY pairs of another rank's rows are ITS work too, counted where
    they are computed -- plus its share of the XX triangle."""
    ops, n, p, d = eng.ops, eng.n, eng.dp, eng.d
    nl, lo = eng.nl, eng.lo
    iters = 30 if n <= 2048 else 6
    tl = eng.tiles.cpu()
    er, ec = (256, 128) if eng.gram_tile == 256 else (eng.gram_tile, eng.gram_tile)  # tile 256 = 256 rows x 128 columns

    def pairs_of(t):  # pairs a tile table covers, mirrored halves of a symmetric tile counted once
        r = (torch.minimum(t[:, 0] + er, t[:, 2]) - t[:, 0]).double()
        c = (torch.minimum(t[:, 1] + ec, t[:, 3]) - t[:, 1]).double()
        # the tiles of a symmetric block's diagonal squares (one 64 / 128 tile, two 256 x 128 tiles side by side) hold each
        # unordered pair twice: r (r + 1) / 2 unique pairs per square of edge r
        same = ((t[:, 0] >= n) == (t[:, 1] >= n)) & (t[:, 1] % n >= t[:, 0] % n) & (t[:, 1] % n < t[:, 0] % n + er) & ((t[:, 4] & 4) == 0) \
            & ((t[:, 4] & 3) != 1)
        return float((torch.where(same, r * (c + 1) / 2 if er == ec else r * c / 2 + c / 2, r * c)).sum())

    out = {}
    bwd_flop = 4.0 * nl * n * D_FEAT
    Wg = eng.Wg if not eng.bf3 else torch.zeros(nl, 2 * n, device=eng.Z.device)
    # the fp32 kernels work on 64-wide tiles: their own table when the engine runs the 128-wide bf16x3 Gram
    if eng.gram_tile == 64:
        t64 = eng.tiles[:eng.n_main] if not eng.front_sharded else eng.tiles
        flop64 = 2.0 * D_FEAT * pairs_of(tl[:t64.shape[0]])
    else:
        t64 = ops.build_tiles(n, 1, eng.rank, eng.world, device=eng.Z.device)
        flop64 = 2.0 * D_FEAT * pairs_of(tl)
    part64 = torch.zeros(t64.shape[0], 4, device=eng.Z.device)
    ms = time_kernel(lambda: ops.mmd_gram(eng.Z, eng.sqn, n, p, eng.bw, t64, False, Wg, n + lo, part64), iters)
    out["mmd_gram_kernel<4,false,1>"] = {"ms": ms, "tflops": flop64 / (ms * 1e-3) / 1e12, "flop": flop64, "tiles": int(t64.shape[0])}
    ms = time_kernel(lambda: ops.mmd_backward(Wg, eng.Z, n + lo, nl, 2 * n, p, eng.Z[lo:lo + nl], eng.gU, mul_shift=eng.center), iters)
    out["mmd_backward_kernel<4,2>"] = {"ms": ms, "tflops": bwd_flop / (ms * 1e-3) / 1e12, "flop": bwd_flop}
    if eng.bf3:
        gs = nl * eng.dp
        gname = {256: "mmd_gram_bf3_wide_kernel", 128: "mmd_gram_bf3_big_kernel"}.get(eng.gram_tile, "mmd_gram_bf3_kernel<64>")
        # as the step launches it: one launch of the first n_main tiles (every XY / YY tile and as many X-X tiles as the launch has
        # free slots for; the late X-X tiles ride in another launch) -- or, sharded front, the whole table in two launches
        nt = eng.tiles.shape[0] if eng.front_sharded else eng.n_main
        main_flop = 2.0 * D_FEAT * pairs_of(tl[:nt])
        ms = time_kernel(lambda: ops.mmd_gram_bf3(eng.Zh, eng.Zl, eng.sqn, n, eng.bw, eng.tiles[:nt], eng.Wh, eng.Wl, n + lo, eng.partial,
                                                  tile=eng.gram_tile, tail_ws=eng.gram_tail_ws, rs_part=eng.rs_part), iters)
        out[gname] = {"ms": ms, "tflops": main_flop / (ms * 1e-3) / 1e12, "flop": main_flop, "tiles": int(nt)}
        if nt < eng.tiles.shape[0]:
            xx_flop = 2.0 * D_FEAT * pairs_of(tl[nt:])
            ms = time_kernel(lambda: ops.mmd_gram_bf3(eng.Zh, eng.Zl, eng.sqn, n, eng.bw, eng.tiles[nt:], None, None, 0,
                                                      eng.partial[nt:], tile=eng.gram_tile), iters)
            out[gname + " [X-X tiles alone]"] = {"ms": ms, "tflops": xx_flop / (ms * 1e-3) / 1e12, "flop": xx_flop,
                                                        "tiles": int(eng.tiles.shape[0] - nt)}
        bwd_edge = ops.mmd_backward_bf3_tile(nl, d, eng.bsplits, eng.bwd_tile) if eng.rm_backward else min(128, ops.mmd_backward_bf3_tile(nl, d, eng.bsplits, eng.bwd_tile))
        bname = {256: "mmd_backward_bf3_wide_kernel", 128: "mmd_backward_bf3_big_kernel"}.get(bwd_edge, "mmd_backward_bf3_kernel<64>")  # the library's own choice
        if eng.rm_backward:  # B operand = the Gram's row-major images (transposed LDS reads)
            ms = time_kernel(lambda: ops.mmd_backward_bf3_rm(eng.Wh, eng.Wl, eng.Zh, eng.Zl, 2 * n, eng.Z, n + lo, nl, d, eng.Z[lo:lo + nl],
                                                             eng.gU, eng.bsplits, gs, mul_shift=eng.center, tile=eng.bwd_tile, rs_part=eng.rs_part), iters)
        else:
            ms = time_kernel(lambda: ops.mmd_backward_bf3(eng.Wh, eng.Wl, eng.ZTh, eng.ZTl, eng.Z, n + lo, nl, d, eng.Z[lo:lo + nl], eng.gU,
                                                          eng.bsplits, gs, mul_shift=eng.center, tile=eng.bwd_tile), iters)
        out[bname] = {"ms": ms, "tflops": bwd_flop / (ms * 1e-3) / 1e12, "flop": bwd_flop}
        out["bf3_prepare_kernel"] = {"ms": time_kernel(lambda: ops.mmd_bf3_prepare(eng.Z, 2 * n, d, eng.Zh, eng.Zl, eng.ZTh, eng.ZTl), iters)}
    return out


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
            "sample": f"{k} VGAN_no_kl steps (batch={N_BATCH}, d={D_FEAT}, fp32) of the PyTorch-CPU port in {dt:.1f} s"}, first_loss, (X, z)


def gpu_first_loss(params, X, z, **engine_kw):
    """Loss of the first step on the same params / batch / noise as the CPU probe (an unsharded engine on this rank's GPU)."""
    import vgan_amd
    from vgan_amd.ops import HipOps
    from vgan_amd.trainer import NoKLStepEngine
    gen = vgan_amd.Generator_big(params[0].shape[1], D_FEAT)
    with torch.no_grad():
        for q, v in zip(gen.parameters(), params):
            q.copy_(torch.as_tensor(v))
    eng = NoKLStepEngine(HipOps(), gen.cuda(), X.cuda(), N_BATCH, 1, noise="host", use_graph=False, loss_accum_scale=1.0,
                         **engine_kw)
    eng.set_epoch_batches(torch.arange(N_BATCH).view(1, -1))
    eng.set_noise(z)
    eng.step()
    return float(eng.loss)


def bench_kl(args):
    """VGAN.fit's step kinds (kl_trainer.KLStepEngine, reference src/vgan.py:253-329) at the selected workload, each replayed
    from its captured HIP graph: the detector step with a trainable encoder (first detector epoch only), the detector step
    with the encoder frozen (every later detector epoch) and the generator-phase step (loss evaluation; 5 of every 6 epochs
    with the reference's iternum_d=1, iternum_g=5).  `value` is the steady-state rate of a fit: 1 frozen-encoder detector epoch
    + 5 generator epochs.  An informational line next to the metric's VGAN_no_kl line."""
    import vgan_amd
    from vgan_amd import synth
    from vgan_amd.kl_trainer import KLStepEngine
    from vgan_amd.ops import HipOps
    torch.cuda.set_device(0)
    torch.manual_seed(777)
    data_np = synth.synthetic_dataset(CONFIG)
    data = torch.as_tensor(data_np).cuda()
    L = synth.latent_size(D_FEAT)
    gen = vgan_amd.Generator_big(L, D_FEAT)
    det = vgan_amd.Detector(L, D_FEAT, vgan_amd.Encoder, vgan_amd.Decoder)
    for mod in (gen, det):
        for q in mod.parameters():
            q.data.normal_(0.0, 0.1) if q.dim() == 2 else q.data.zero_()     # the reference's weights_init (src/vgan.py:69-78)
    gen_p = [q.detach().numpy().copy() for q in gen.parameters()]
    det_p = [q.detach().numpy().copy() for q in det.parameters()]
    ops = HipOps()
    # the feed VGAN.fit uses by default: device-resident epoch table walked by the step counter, Philox noise drawn in the step
    nb = data.shape[0] // N_BATCH
    eng = KLStepEngine(ops, gen.cuda(), det.cuda(), data, N_BATCH, 0.007, 0.04, 0.0, batches_per_epoch=nb, noise="device", seed=777)
    table = torch.randperm(data.shape[0])[:nb * N_BATCH].view(nb, N_BATCH)
    eng.set_epoch_batches(table)
    idx = table[0]
    z = torch.randn(N_BATCH, L)
    steps = min(args.steps, 1000)
    # an epoch of a step kind per call, as VGAN.fit issues them (blocks of steps per graph launch)
    kinds = {"detector_step_trainable_encoder": lambda: eng.detector_step(train_encoder=True, count=nb),
             "detector_step_frozen_encoder": lambda: eng.detector_step(train_encoder=False, count=nb),
             "generator_phase_step": lambda: eng.generator_phase_step(count=nb)}
    res = {}
    epochs = max(1, steps // nb)
    for name, fn in kinds.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(epochs):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / (epochs * nb)
    steps = epochs * nb
    mix = (res["detector_step_frozen_encoder"] + 5.0 * res["generator_phase_step"]) / 6.0
    # the MMD of this path runs at p = L (exp-bound regime): algorithmic 8 n^2 L flop and 2 n^2 exps per step
    n, p = N_BATCH, eng.pz
    gram_ms = time_kernel(lambda: ops.mmd_gram(eng.encZ, eng.sq, n, p, eng.bw, eng.tiles0, False, None, 0, eng.partial))
    flop = 4.0 * n * n * L
    out = {"metric": f"VGAN.fit steps/sec (batch={N_BATCH}, d={D_FEAT})", "value": 1.0 / mix, "unit": "steps/s", "n_gpus": 1,
           "steps": steps, "ms_per_step": 1e3 * mix, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": WORKLOAD + " [VGAN.fit: 1 detector epoch (encoder frozen) : 5 generator epochs, HIP-graph replay, device-resident feed]",
                      "step_kinds_ms": {k: 1e3 * v for k, v in res.items()}},
           "roofline": {"bound": "mfma", "kernel": "mmd_gram_kernel<4,false,1> at p = L", "achieved": flop / (gram_ms * 1e-3) / 1e12,
                        "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / (gram_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                        "traffic": None, "avg_launch_ms": gram_ms, "algorithmic_flop_per_launch": flop,
                        "note": "at p = L = d/16 the MMD is bound by its 2 n^2 exponentials and the epilogue, not by the Gram; the "
                                "step itself is launch-latency bound (see step_kinds_ms)"}}
    if not args.no_cpu_baseline:
        from oracle import torch_port as port
        cores = usable_cores()
        torch.set_num_threads(cores)
        tr = port.PortKL(gen_p, det_p, weight=0.0)
        rng = np.random.default_rng(0)
        X = torch.as_tensor(data_np[idx.numpy()])
        tr.detector_step(X, z)          # calibration + warm-up
        tr.generator_phase_step(X, z)   # freezes the encoder, as the first generator phase of a fit does
        t0 = time.perf_counter()
        k = 0
        while k < 6 or (time.perf_counter() - t0 < args.cpu_seconds and k < 600):
            rows = rng.permutation(data_np.shape[0])[:N_BATCH]
            Xb, zb = torch.as_tensor(data_np[rows]), torch.as_tensor(rng.normal(size=(N_BATCH, L)).astype(np.float32))
            if k % 6 == 0:
                tr.detector_step(Xb, zb)
            else:
                tr.generator_phase_step(Xb, zb)
            k += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / dt, "unit": "steps/s", "cores": cores, "kind": "port",
                               "sample": f"{k} VGAN.fit steps (1 detector : 5 generator-phase, batch={N_BATCH}, d={D_FEAT}, fp32) of the "
                                         f"PyTorch-CPU port (oracle/torch_port.PortKL) in {dt:.1f} s"}
        out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)


def traffic_of(kernel, world):
    """HBM-side bytes per launch of `kernel` from the COMMITTED PMC passes (profiles/traffic*.json, regenerated by
    tools/pmc_traffic.py under rocprofv3 --pmc: counters cannot be collected inside this timing run)."""
    name = {"c3": "traffic.json", "c4": "traffic_c4.json", "c5": "traffic_c5.json"}.get(CONFIG)
    if name is None or world != 1:
        return None, None
    tp = os.path.join(ROOT, "profiles", name)
    try:
        v = json.load(open(tp)).get(kernel.split("<")[0].split(" ")[0], {}).get("hbm_bytes_per_launch")
    except Exception:
        return None, None
    return v, (f"profiles/{name}: committed rocprofv3 --pmc pass (tools/pmc_traffic.py on tools/kbench.py), not measured in this run"
               if v is not None else None)


def roofline_of(e, kern, steps_per_s, world):
    """The roofline block: the LONGEST of the two MMD contractions of the active precision mode is the dominant kernel."""
    if e.bf3:
        # The kernels issue v_mfma_f32_32x32x16_bf16; `achieved` is ALGORITHMIC flops / launch time as the contract says,
        # `peak` the dense bf16 MFMA rate.  Each algorithmic product costs three bf16 products (hi.hi' + hi.lo' + lo.hi'),
        # so the executed MFMA rate is 3x `achieved`; both fractions are reported.
        peak = BF16_MFMA_PEAK_TFLOPS
        cands = [k for k in kern if "bf3" in k and "tflops" in kern[k] and "alone" not in k]
    else:
        peak = FP32_MFMA_PEAK_TFLOPS
        cands = [k for k in kern if "bf3" not in k and "tflops" in kern[k]]
    name = max(cands, key=lambda k: kern[k]["ms"])
    g = kern[name]
    extra = {}
    if e.bf3:
        extra = {"executed_mfma_tflops": 3.0 * g["tflops"], "executed_frac": 3.0 * g["tflops"] / peak,
                 "vs_fp32_mfma_peak": g["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                 "note": "fp32-accurate contraction on the bf16 MFMA via a 3-way operand split; the fp32-MFMA kernels of the same "
                         "contractions are timed as a whole step in 'fp32_mode'"}
    also = {}
    for k, v in kern.items():
        if k == name:
            continue
        if "tflops" not in v:
            also[k] = {"avg_launch_ms": v["ms"], "bound": "hbm"}
            continue
        pk = BF16_MFMA_PEAK_TFLOPS if "bf3" in k else FP32_MFMA_PEAK_TFLOPS
        tr, _ = traffic_of(k, world) if " [" not in k else (None, None)
        also[k] = {"achieved": v["tflops"], "peak": pk, "frac": v["tflops"] / pk, "avg_launch_ms": v["ms"],
                   "algorithmic_flop_per_launch": v["flop"], "traffic": tr}
        if "bf3" in k:
            also[k]["executed_frac"] = 3.0 * v["tflops"] / pk
    step_flop = 8.0 * N_BATCH * N_BATCH * D_FEAT + 6.0 * N_BATCH * e.fp.total
    tr, src = traffic_of(name, world)
    return {"bound": "mfma", "kernel": name, "achieved": g["tflops"], "peak": peak, "unit": "TFLOP/s",
            "frac": g["tflops"] / peak, "traffic": tr, "traffic_source": src,
            "avg_launch_ms": g["ms"], "algorithmic_flop_per_launch": g["flop"], **extra, "also": also,
            "step_tflops": step_flop * steps_per_s / world / 1e12,
            "step_frac_of_fp32_mfma_peak": step_flop * steps_per_s / world / (FP32_MFMA_PEAK_TFLOPS * 1e12)}


def main():
    args = parse()
    select_workload(args.workload)
    if args.path == "kl":
        return bench_kl(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    # rehearsal knobs (a one-GPU box cannot run RCCL between two ranks of the same device): VGAN_BENCH_BACKEND=gloo puts all
    # ranks on device VGAN_BENCH_DEVICE and carries the collectives through the host (no graph capture then)
    backend = os.environ.get("VGAN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = int(os.environ.get("VGAN_BENCH_DEVICE", "0"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    torch.manual_seed(1234)
    graph_ok = [not args.no_graph and backend == "nccl"]

    def prewarm(e):
        # every rank must run the SAME number of steps (each one holds a collective and draws the shared shuffles): rank 0
        # times one epoch and broadcasts how many more to run
        t0 = time.perf_counter()
        run_steps(e, EPOCH_BATCHES, 0)
        torch.cuda.synchronize()
        more = torch.tensor([max(0, min(2000, int(args.prewarm_seconds / max(time.perf_counter() - t0, 1e-6))))], device="cuda")
        if dist:
            dist.broadcast(more, src=0)
        for it in range(int(more.item())):
            run_steps(e, EPOCH_BATCHES, (it + 1) * EPOCH_BATCHES)
        torch.cuda.synchronize()

    def timed_leg(steps, warmup, **ekw):
        """Build an engine, pre-warm, W warm-up steps, one untimed block of K steps (captures every graph the block replays),
        then `repeats` blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides, MAX over ranks.
        Returns (engine, data, params, per-block seconds, mean loss of the timed steps)."""
        def warm():
            e, data, params = build_engine(rank, world, graph_ok[0], **ekw)
            prewarm(e)
            run_steps(e, warmup, 0)
            run_steps(e, steps, 0)
            torch.cuda.synchronize()
            return e, data, params
        try:
            eng, data, params = warm()
        except Exception as e:  # a collective that cannot be captured: fall back to eager launches (same process)
            if not graph_ok[0]:
                raise
            print(f"[bench] HIP-graph path failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
            graph_ok[0] = False
            eng, data, params = warm()
        eng.epoch_loss()
        eng.first_timed_step = eng.steps_done  # training steps already taken: `mean_loss` is the mean over the timed steps
        blocks = []
        for _ in range(max(1, args.repeats)):
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(eng, steps, 0)
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            if dist:
                t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            blocks.append(elapsed)
        mean_loss = eng.epoch_loss() * EPOCH_BATCHES / max(steps * len(blocks), 1)
        return eng, data, params, blocks, mean_loss

    def rates(blocks, steps):
        med = sorted(blocks)[len(blocks) // 2]
        return {"value": steps / med, "ms_per_step": 1e3 * med / steps, "steps": steps, "repeats": len(blocks),
                "repeat_steps_per_s": [steps / b for b in blocks], "min": steps / max(blocks), "max": steps / min(blocks)}

    steps, warmup = args.steps, args.warmup
    if CONFIG in ("c4", "c5"):
        steps, warmup = min(steps, 200), min(warmup, 8)
        args.no_cpu_baseline = args.no_cpu_baseline or CONFIG == "c5"  # N = 16384: ~17 GB and ~15 s per CPU step
    ekw = {}
    if args.precision:
        ekw["mmd_precision"] = args.precision
    if args.front:
        ekw["front"] = args.front
    eng, data, params, blocks, mean_loss = timed_leg(steps, warmup, **ekw)
    main_rate = rates(blocks, steps)
    kern = kernel_rooflines(eng)
    main_cfg = {"workload": WORKLOAD, "global_batch": N_BATCH, "features": D_FEAT, "latent": eng.L,
                "rows_per_gpu": eng.nl, "parallelism": f"dp{world} (row-sharded Gram, {'sharded' if eng.front_sharded else 'replicated'} front)",
                "hip_graph": bool(graph_ok[0] and eng.use_graph), "mean_loss": mean_loss, "first_timed_step": eng.first_timed_step,
                "prewarm_s": args.prewarm_seconds, "generator": eng.mode, "chain_association": "flops" if eng.chain_flops else "depth",
                "mmd_precision": eng.precision, "gram_tile": eng.gram_tile, "timed_blocks": len(blocks)}
    main_roof = roofline_of(eng, kern, main_rate["value"], world)
    main_bf3, main_precision = eng.bf3, eng.precision
    del eng
    torch.cuda.empty_cache()

    # The one line, as far as the metric's own leg goes.  Everything below is OPTIONAL evidence added to the same line: a leg that
    # raises is recorded as {"error": ...}, and if the optional legs together overrun --optional-seconds (a collective that never
    # returns on some rank, say) a watchdog prints the line as it stands and ends the process, so that the metric is never lost
    # to its side measurements.
    out = {
        "metric": f"V-GAN train steps/sec (batch={N_BATCH}, d={D_FEAT})", "value": main_rate["value"], "unit": "steps/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": main_rate["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("f32; MMD contractions of `value`: split-bf16 x3 on the bf16 MFMA with fp32 accumulation (centred operand); "
                  "the same step with the fp32 MFMA is timed in `fp32_mode`") if main_bf3 else "f32", "data": "synthetic",
        "config": main_cfg, "repeat_steps_per_s": main_rate["repeat_steps_per_s"],
        "timing": f"median of {main_rate['repeats']} blocks of {steps} steps, each bracketed by barrier + synchronize, max over ranks",
        "roofline": main_roof,
    }
    import threading
    lock, printed = threading.Lock(), [False]

    def emit(note=None):
        with lock:
            if printed[0]:
                return
            printed[0] = True
            if note:
                out["optional_legs"] = note
            if rank == 0:
                print(json.dumps(out), flush=True)

    def overrun():
        emit(f"stopped by the watchdog after {args.optional_seconds:.0f} s; legs finished until then are in the line")
        sys.stderr.write(f"[bench] rank {rank}: optional legs overran {args.optional_seconds:.0f} s, exiting\n")
        sys.stderr.flush()
        os._exit(0)

    watchdog = threading.Timer(args.optional_seconds + (0.0 if rank == 0 else 5.0), overrun)
    watchdog.daemon = True
    watchdog.start()

    def optional(name, fn):
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 -- an optional leg must not take the metric with it
            sys.stderr.write(f"[bench] optional leg {name} failed: {type(e).__name__}: {e}\n")
            return {"error": f"{type(e).__name__}: {e}"}

    # The reference's own arithmetic is fp32 end to end.  When the engine's choice is the split-bf16 mode, the SAME
    # workload is timed a second time with the fp32-MFMA kernels (same warm-up discipline, same step count), so that the
    # driver's line carries a step rate for both arithmetic modes.
    def fp32_leg():
        e32, _, _, b32, ml32 = timed_leg(steps, warmup, **{**ekw, "mmd_precision": "fp32"})
        r32 = rates(b32, steps)
        blk = {**r32, "unit": "steps/s", "warmup": warmup, "mmd_precision": "fp32", "mean_loss": ml32,
               "first_timed_step": e32.first_timed_step, "dtype": "f32 (fp32 MFMA, v_mfma_f32_32x32x2_f32)",
               "roofline": roofline_of(e32, kernel_rooflines(e32), r32["value"], world)}
        del e32
        torch.cuda.empty_cache()
        return blk

    if main_bf3:
        blk = optional("fp32_mode", fp32_leg)
        with lock:
            out["fp32_mode"] = blk

    # ---- the larger BASELINE.json configurations and the alternative exchange schedules, in the same invocation ----------
    if CONFIG == "c3" and not args.no_extra and not args.precision and not args.front:
        legs = []
        if world > 1:
            legs.append(("c3_overlapped_allreduce", "c3", dict(overlap_exchange=True)))
        legs.append(("c4", "c4", {}))
        if world > 1:
            legs.append(("c4_replicated_front", "c4", dict(front="replicated")))
        legs.append(("c5_fp32", "c5", dict(mmd_precision="fp32")))
        if world > 1:
            legs.append(("c5_fp32_replicated_front", "c5", dict(mmd_precision="fp32", front="replicated")))
        legs.append(("c5_bf16x3", "c5", {}))
        if world > 1:
            legs.append(("c5_bf16x3_replicated_front", "c5", dict(front="replicated")))

        def extra_leg(cfg, kw):
            select_workload(cfg)
            k, w = (steps, warmup) if cfg == "c3" else (min(steps, 40 if cfg == "c4" else 16), min(warmup, 8))
            e, _, _, b, ml = timed_leg(k, w, **kw)
            r = {"workload": WORKLOAD, **rates(b, k), "unit": "steps/s", "warmup": w, "n_gpus": world, "scaling": "strong",
                 "global_batch": N_BATCH, "features": D_FEAT, "rows_per_gpu": e.nl, "mmd_precision": e.precision,
                 "front": "sharded" if e.front_sharded else "replicated", "overlapped_allreduce": bool(e.overlap),
                 "chain_association": "flops" if e.chain_flops else "depth", "gram_tile": e.gram_tile,
                 "hip_graph": bool(graph_ok[0] and e.use_graph), "mean_loss": ml}
            del e
            torch.cuda.empty_cache()
            return r

        with lock:
            out["extra"] = {}
        for name, cfg, kw in legs:
            r = optional(name, lambda: extra_leg(cfg, kw))
            with lock:
                out["extra"][name] = r
        select_workload(args.workload)

    if rank == 0 and not args.no_cpu_baseline:  # rank 0, whatever N: the other ranks wait at the closing barrier

        def cpu_leg():
            cb, cpu_loss, (X, z) = cpu_baseline(data, params, args.cpu_seconds)
            gl = gpu_first_loss(params, X, z, mmd_precision=main_precision)
            par = {"mmd_precision": main_precision, "loss_gpu": gl, "loss_cpu_port": cpu_loss, "abs_diff": abs(gl - cpu_loss), "bar": 1e-4}
            if main_bf3:
                g32 = gpu_first_loss(params, X, z, mmd_precision="fp32")
                par["fp32_mode"] = {"loss_gpu": g32, "abs_diff": abs(g32 - cpu_loss)}
            return cb, par

        res = optional("cpu_baseline", cpu_leg)
        with lock:
            if isinstance(res, dict):
                out["cpu_baseline"] = res
            else:
                out["cpu_baseline"], out["parity"] = res
                out["speedup_vs_cpu"] = main_rate["value"] / res[0]["value"]
    watchdog.cancel()
    emit()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
