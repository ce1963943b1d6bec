#!/usr/bin/env python3
"""Micro-benchmark of the two MMD kernels at the bench configuration (n=1024, d=784) -- tuning aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

eng, data, params = bench.build_engine(0, 1, False)
bench.run_steps(eng, 3, 0)
torch.cuda.synchronize()
k = bench.kernel_rooflines(eng)
print({n: (round(v["ms"] * 1e3, 1), round(v.get("tflops", 0.0), 1)) for n, v in k.items()})
