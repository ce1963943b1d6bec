#!/usr/bin/env python3
"""Exercises the data-parallel code path (RCCL collectives on the step's stream, with and without HIP-graph capture)
on however many GPUs the launcher provides -- including ONE (world_size 1 still issues every collective), which is
all a single-GPU box can check.  Launch:  python -m torch.distributed.run --standalone --nproc-per-node N tools/dp_selftest.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import bench

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))


only = os.environ.get("VGAN_SELFTEST_ONLY_G")  # profiling aid: skip everything but the 1/G shard emulation
for graph in (() if only else (False, True)):
    eng, data, params = bench.build_engine(rank, world, graph, force_exchange=True)
    losses = []
    for t in range(6):
        if t % bench.EPOCH_BATCHES == 0:
            eng.set_epoch_batches(torch.arange(bench.N_BATCH * bench.EPOCH_BATCHES).view(bench.EPOCH_BATCHES, -1))
        eng.step()
        losses.append(eng.step_loss())
    torch.cuda.synchronize()
    if rank == 0:
        print(f"graph={graph} world={world}: losses {np.round(losses, 6).tolist()} captured={eng.graph is not None}")

# Per-rank step time of a G-way row shard, measured on THIS GPU: the engine is built as rank 0 of G while the process
# group has `world` members, so the kernels do exactly one rank's share and the collective is issued but is cheap.
# A lower bound for the G-GPU step (the real all-reduce latency comes on top); numerics are meaningless here.
import time
if world == 1:
    for G in ((int(only),) if only else (1, 2, 4, 8)):
        plain_only = os.environ.get("VGAN_SELFTEST_PLAIN") == "1"  # profiling aid: the default schedule only
        for schedule in ((False,) if plain_only else (False, "serial", True)):  # plain | overlapped schedule on one stream | on a side stream
            eng, data, params = bench.build_engine(0, G, True, force_exchange=True, overlap_exchange=schedule)
            bench.run_steps(eng, 64, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bench.run_steps(eng, 800, 64)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 800
            print(f"emulated shard 1/{G}, exchange schedule {schedule!s:6s}: {dt * 1e6:.1f} us per step per rank "
                  f"({1.0 / dt:.0f} steps/s if collectives were free)", flush=True)
dist.barrier()
dist.destroy_process_group()
