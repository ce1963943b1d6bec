"""Loss-vs-step curves of the bench workload (c3: d=784, batch=1024, device noise) in both MMD precision modes.

    python tools/loss_curve.py [--steps 6000] [--every 10] [--out profiles/r02_loss_curve_c3.csv]

Both runs see the same shuffles (torch seed 1234, as bench.py) and the same Philox noise stream, so the curves are
comparable step by step.  The CSV holds step, loss_fp32, loss_bf16x3; the summary printed at the end gives the window means
that bench.py's `mean_loss` would report for windows starting at different steps -- which is what explains the scatter of
that field between runs.  Measurement aid, not part of the product.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(precision, steps):
    import bench
    bench.N_BATCH, bench.D_FEAT, bench.EPOCH_BATCHES, bench.WORKLOAD = bench.WORKLOADS["c3"]
    bench.CONFIG = "c3"
    torch.manual_seed(1234)
    eng, _, _ = bench.build_engine(0, 1, True, mmd_precision=precision)
    hist = torch.zeros(steps, device="cuda")
    from vgan_amd.vgan import epoch_batches
    for t in range(steps):
        if t % bench.EPOCH_BATCHES == 0:
            eng.set_epoch_batches(epoch_batches(eng.data.shape[0], bench.N_BATCH))
        eng.step()
        hist[t:t + 1].copy_(eng.loss)
    torch.cuda.synchronize()
    return hist.cpu().numpy()


def run_against_cpu_port(precision, steps, out, every):
    """The same c3 training run on the GPU engine and on the op-for-op PyTorch-CPU port of the reference step
    (oracle/torch_port.PortNoKL: cdist / exp / mean / topk / autograd / torch.optim.Adadelta in fp32), step by step on
    IDENTICAL batches and host-drawn noise: the comparison with the reference's own arithmetic THROUGH the regime change of the
    loss (the penalty term of Mmd_loss_constrained.py:50 switching on near step 3 100), which the 200-step parity test stops
    short of.  CPU-bound (~9 port steps/s on a GPU box's 16 cores)."""
    import bench
    from oracle import torch_port as port
    from vgan_amd.vgan import epoch_batches
    bench.select_workload("c3")
    torch.manual_seed(1234)
    torch.set_num_threads(bench.usable_cores())
    eng, data, params = bench.build_engine(0, 1, True, mmd_precision=precision, noise="host")
    ref = port.PortNoKL(params)
    L = params[0].shape[1]
    data_t = torch.as_tensor(data)
    g, c = np.zeros(steps), np.zeros(steps)
    nb = bench.EPOCH_BATCHES
    for t in range(steps):
        if t % nb == 0:
            table = epoch_batches(data.shape[0], bench.N_BATCH)
            eng.set_epoch_batches(table)
        z = torch.randn(bench.N_BATCH, L)
        eng.set_noise(z)
        eng.step()
        c[t] = ref.step(data_t[table[t % nb]], z)
        g[t] = float(eng.loss)
        if t % 100 == 99 or t == steps - 1:
            d = np.abs(g[:t + 1] - c[:t + 1])
            print(f"step {t + 1}: loss gpu {g[t]:.6f} cpu-port {c[t]:.6f}; max |diff| so far {d.max():.3e} at step {int(d.argmax())}", flush=True)
            with open(out, "w") as f:
                f.write(f"step,loss_gpu_{precision},loss_cpu_port\n")
                for q in range(0, t + 1, every):
                    f.write(f"{q},{g[q]:.6f},{c[q]:.6f}\n")
    d = np.abs(g - c)
    jump = lambda v: int(np.argmax(v > 0.5 * (v.min() + v.max())))
    print(f"{precision} vs CPU port over {steps} steps: max |loss_gpu - loss_cpu_port| = {d.max():.3e} at step {int(d.argmax())}; "
          f"steps over the 1e-4 bar: {int((d > 1e-4).sum())}; first step above the loss midpoint: gpu {jump(g)}, cpu port {jump(c)}; "
          f"final losses {g[-1]:.5f} / {c[-1]:.5f}")
    for lo in range(0, steps, 500):
        hi = min(lo + 500, steps)
        print(f"   steps [{lo}, {hi}): max |diff| {d[lo:hi].max():.3e}, mean loss gpu {g[lo:hi].mean():.5f} cpu {c[lo:hi].mean():.5f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_loss_curve_c3.csv"))
    ap.add_argument("--with-cpu-port", choices=["fp32", "bf16x3"], default=None,
                    help="run the GPU engine (this precision) and the PyTorch-CPU port side by side on identical inputs instead")
    a = ap.parse_args()
    if a.with_cpu_port:
        return run_against_cpu_port(a.with_cpu_port, a.steps, a.out, a.every)
    curves = {p: run(p, a.steps) for p in ("fp32", "bf16x3")}
    with open(a.out, "w") as f:
        f.write("step,loss_fp32,loss_bf16x3\n")
        for t in range(0, a.steps, a.every):
            f.write(f"{t},{curves['fp32'][t]:.6f},{curves['bf16x3'][t]:.6f}\n")
    d = np.abs(curves["fp32"] - curves["bf16x3"])
    print(f"max |loss_fp32 - loss_bf16x3| over {a.steps} steps: {d.max():.3e} at step {int(d.argmax())}; "
          f"first step over 1e-4: {int(np.argmax(d > 1e-4)) if (d > 1e-4).any() else None}")
    for p, c in curves.items():
        jump = int(np.argmax(c > 0.5 * (c.min() + c.max())))
        print(f"{p}: loss[0] = {c[0]:.4f}, min = {c.min():.4f} at {int(c.argmin())}, max = {c.max():.4f} at {int(c.argmax())}, "
              f"first step above the midpoint: {jump}")
        for start in range(0, a.steps - 1999, 500):
            print(f"   mean over steps [{start}, {start + 2000}): {c[start:start + 2000].mean():.4f}")


if __name__ == "__main__":
    main()
