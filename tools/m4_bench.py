import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from vgan_amd.ops import HipOps
ops = HipOps()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (n, d, e0) in [(8192, 4096, 260), (1024, 4096, 260), (4096, 2048, 132), (512, 2048, 132), (1024, 784, 52)]:
    dl = torch.randn(n, d, device="cuda"); z = torch.randn(n, e0, device="cuda"); M = torch.zeros(d, e0, device="cuda")
    base = t(lambda: ops.linear_backward_params(dl, z, M, None))
    ref = M.clone()
    res = [f"n={n} d={d} e0={e0}: linear_backward_params {base:.1f} us"]
    t64 = ((d + 63) // 64) * ((e0 + 63) // 64)
    for sp in (1, 2, 3, 4, 6, 8):
        if n // sp < 256: continue
        slabs = torch.zeros(sp, d, e0, device="cuda")
        def run():
            if sp == 1:
                ops.gemm_grouped([("TN", dl, z, M)])
            else:
                ops.gemm_grouped([("TN", dl, z, slabs, sp)])
                ops.reduce_slabs(slabs, d * e0, sp, M)
        us = t(run)
        err = float((M - ref).abs().max() / ref.abs().max())
        res.append(f"grouped TN x{sp} ({t64 * sp} wgs) {us:.1f} us (rel err {err:.1e})")
    print("; ".join(res), flush=True)
    # logits
    At4 = torch.randn(d, e0, device="cuda"); lg = torch.zeros(n, d, device="cuda")
    a = t(lambda: ops.linear_forward(z, At4, None, lg))
    b = t(lambda: ops.gemm_grouped([("NT", z, At4, lg)]))
    print(f"   logits: linear_forward {a:.1f} us, grouped NT {b:.1f} us", flush=True)
