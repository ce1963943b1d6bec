#!/usr/bin/env python3
"""Upper bound of what operand locality could buy the wide Gram: the launch on its real tile table, on a table in which every tile reads
the same two panels (every fill an L2 hit) and on one 4 x 8 block of tiles repeated; no W stores.   python3 tools/l2_bound.py c4|c5"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "c5"
bench.select_workload(w)
eng, data, params = bench.build_engine(0, 1, False, mmd_precision="bf16x3")
bench.run_steps(eng, 3, 0)
torch.cuda.synchronize()
ops, n = eng.ops, eng.n
T = eng.tiles.shape[0]
tl = eng.tiles.clone()
def t(tiles, label):
    ms = bench.time_kernel(lambda: ops.mmd_gram_bf3(eng.Zh, eng.Zl, eng.sqn, n, eng.bw, tiles, None, None, 0, eng.partial, tile=eng.gram_tile), 20)
    print(label, tiles.shape[0], "tiles", round(ms * 1e3, 1), "us")
nt = (T // 256) * 256
t(tl[:nt], "normal table, no W store")
same = tl[:nt].clone()
same[:, 0] = 0; same[:, 1] = n; same[:, 2] = 256; same[:, 3] = n + 128   # every tile: rows 0..255 x columns n..n+127
t(same, "every tile the SAME panels (all L2 hits)")
# eight distinct panels per XCD-round: tiles of one 4 x 8 block repeated
blk = tl[:nt].clone()
idx = torch.arange(nt, device=tl.device)
blk[:, 0] = ((idx // 8) % 4) * 256; blk[:, 2] = blk[:, 0] + 256
blk[:, 1] = n + ((idx // 32) % 8) * 128; blk[:, 3] = blk[:, 1] + 128
t(blk, "32 distinct tiles (4 x 8 block) repeated")
