"""Error table of the two MMD precision modes on adversarial operands, with and without the centred operand.

    python tools/precision_probe.py [--cpu] [--n 1024] [--d 784]

For each case of tests/test_hip_parity.py::ADVERSARIAL it runs one whole training step on the engine (GPU kernels; --cpu
uses the numpy model of the kernels' arithmetic in tests/cpu_ops.py) and prints |loss - float64 oracle|, the relative
bandwidth error and the largest parameter-gradient error relative to that gradient's largest entry.  The oracle is test
infrastructure: this tool is a measurement aid, not part of the product.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from oracle import vgan_oracle as orc  # noqa: E402
from test_hip_parity import ADVERSARIAL, adversarial_case, oracle_step_with_decisions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--d", type=int, default=784)
    a = ap.parse_args()
    from vgan_amd.modules import Generator_big
    from vgan_amd.trainer import NoKLStepEngine
    if a.cpu:
        from cpu_ops import CpuOps
        ops, device = CpuOps(), "cpu"
    else:
        from vgan_amd.ops import HipOps
        ops, device = HipOps(), "cuda"
    n, d = a.n, a.d
    print(f"# n={n} d={d} provider={ops.name}")
    print(f"{'case':20s} {'mode':7s} {'centre':6s} {'|dloss|':>10s} {'bw rel':>10s} {'grad/max':>10s} {'loss':>10s}")
    for case in ADVERSARIAL:
        data, params = adversarial_case(case, n, d, rows=n)
        z = np.random.default_rng(5).normal(size=(n, orc.latent_size(d))).astype(np.float32)
        for precision in ("fp32", "bf16x3"):
            for centre in (True, False):
                gen = Generator_big(orc.latent_size(d), d)
                with torch.no_grad():
                    for q, v in zip(gen.parameters(), params):
                        q.copy_(torch.as_tensor(v))
                eng = NoKLStepEngine(ops, gen.to(device), torch.as_tensor(data).to(device), n, 1, noise="host", use_graph=False,
                                     loss_accum_scale=1.0, mmd_precision=precision, center_operand=centre)
                eng.set_epoch_batches(torch.arange(n).view(1, n))
                eng.set_noise(torch.as_tensor(z))
                eng.step()
                want = oracle_step_with_decisions(params, data, z, 10.0, eng.S.cpu().numpy() >= np.float32(1.0 / d))
                gerr = max(float(np.abs(eng.grad_view(i).cpu().numpy() - want["grads"][i]).max() / max(np.abs(want["grads"][i]).max(), 1e-30))
                           for i in range(8))
                print(f"{case:20s} {precision:7s} {str(centre):6s} {abs(float(eng.loss) - float(want['loss'])):10.2e} "
                      f"{abs(float(eng.bw) / float(want['bw']) - 1):10.2e} {gerr:10.2e} {float(want['loss']):10.5f} ties={want['ties']}", flush=True)


if __name__ == "__main__":
    main()
