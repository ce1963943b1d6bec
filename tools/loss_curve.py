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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_loss_curve_c3.csv"))
    a = ap.parse_args()
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
