#!/usr/bin/env python3
"""Micro-benchmark of the MMD kernels at a bench workload (default c3: n=1024, d=784; VGAN_KBENCH_WORKLOAD=c4|c5 for the
large configurations) -- the target of the separate rocprofv3 --pmc passes that tools/pmc_traffic.py folds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = os.environ.get("VGAN_KBENCH_WORKLOAD", "c3")
bench.CONFIG = cfg
bench.N_BATCH, bench.D_FEAT, bench.EPOCH_BATCHES, bench.WORKLOAD = bench.WORKLOADS[cfg]
eng, data, params = bench.build_engine(0, 1, False)
bench.run_steps(eng, 3, 0)
torch.cuda.synchronize()
k = bench.kernel_rooflines(eng)
print({n: (round(v["ms"] * 1e3, 1), round(v.get("tflops", 0.0), 1)) for n, v in k.items()})
