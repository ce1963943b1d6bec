#!/usr/bin/env python3
"""Where did the Gram tiles run?  Histogram of workgroups per (XCD, SE, CU) from the diagnostic words the
kernel leaves in partial[tile].zw."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
eng, data, params = bench.build_engine(0, 1, False)
bench.run_steps(eng, 3, 0)
torch.cuda.synchronize()
p = eng.partial.view(torch.int32).cpu().numpy().reshape(-1, 4)
hw, xcc = p[:, 2].astype('uint32'), p[:, 3].astype('uint32') & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cnt = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print("tiles", len(p), "distinct CUs", len(cnt), "hist of WGs/CU", sorted(collections.Counter(cnt.values()).items()))
print("per XCD", sorted(collections.Counter(xcc.tolist()).items()))
