#!/usr/bin/env python3
"""How long a wide (256 x 128 tile) Gram launch takes as a function of its tile count, with and without the K split of a short
last round (include/vgan_hip.h, tail_ws):   python3 tools/gram_rounds.py c4|c5
Launches the first nt tiles of the workload's table for nt around one round of 256 (one 768-thread workgroup holds a CU) and
prints the time of each, whole and split.  These numbers calibrate `_launch_rounds` in v-gan_amd/trainer.py."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from vgan_amd.ops import HipOps
w = sys.argv[1] if len(sys.argv) > 1 else "c4"
bench.select_workload(w)
eng, data, params = bench.build_engine(0, 1, False, mmd_precision="bf16x3")
bench.run_steps(eng, 3, 0)
torch.cuda.synchronize()
ops, n = eng.ops, eng.n
T = eng.tiles.shape[0]
print("tiles", T, "gram_tile", eng.gram_tile)
R = 256
for nt in [R, R + 16, R + 32, R + 64, R + 96, R + 128, R + 132, 2 * R]:
    if nt > T: continue
    for ws in (None, eng.gram_tail_ws):
        ms = bench.time_kernel(lambda: ops.mmd_gram_bf3(eng.Zh, eng.Zl, eng.sqn, n, eng.bw, eng.tiles[:nt], eng.Wh, eng.Wl, n, eng.partial, tile=eng.gram_tile, tail_ws=ws), 20)
        print(nt, "split" if ws is not None else "whole", round(ms * 1e3, 1), "us")
