#!/usr/bin/env python3
"""Exercises the data-parallel code path (RCCL collectives on the step's stream, with and without HIP-graph capture)
on however many GPUs the launcher provides -- including ONE (world_size 1 still issues every collective), which is
all a single-GPU box can check.

  python -m torch.distributed.run --standalone --nproc-per-node N tools/dp_selftest.py [--workload c3|c4|c5]
        [--shards 1,2,4,8] [--schedules plain,serial,overlap] [--fronts replicated,sharded] [--precision fp32|bf16x3]
        [--steps K] [--skip-check]

With one process it also EMULATES a G-way row shard on this GPU: the engine is built as rank 0 of G while the process group
has one member, so the kernels do exactly one rank's share of the step (its Gram rows, its share of the X-X triangle, its rows
of the backward product, of the mask backward and of M_4; with the sharded front also only its rows of the logits product,
mask / projection and operand split) and every collective is issued but moves nothing.  The per-rank step time is a LOWER
bound for the G-GPU step (the real all-reduce / all-gather time comes on top, where it is not hidden); numerics are
meaningless there (the other ranks' rows are never produced).  Run under `rocprofv3 --kernel-trace --stats` with one shard
count per invocation for the per-launch breakdown (tools/dp_shards.sh)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", choices=sorted(bench.WORKLOADS), default="c3")
ap.add_argument("--shards", default="1,2,4,8")
ap.add_argument("--schedules", default=None, help="c3 default: plain,serial,overlap (the all-reduce overlap schedules); others: plain")
ap.add_argument("--fronts", default=None, help="default: the engine's rule (auto); e.g. replicated,sharded to time both")
ap.add_argument("--precision", default=None)
ap.add_argument("--steps", type=int, default=None)
ap.add_argument("--skip-check", action="store_true", help="skip the collectives-in-graph check, emulate shards only")
args = ap.parse_args()
bench.select_workload(args.workload)

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
ekw = {"mmd_precision": args.precision} if args.precision else {}

if not args.skip_check:
    for graph in (False, True):
        for front in ("replicated", "sharded"):
            eng, data, params = bench.build_engine(rank, world, graph, force_exchange=True, front=front, **ekw)
            losses = []
            nb = bench.EPOCH_BATCHES
            for t in range(6):
                if t % nb == 0:
                    eng.set_epoch_batches(torch.arange(bench.N_BATCH * nb).view(nb, -1))
                eng.step()
                losses.append(eng.step_loss())
            torch.cuda.synchronize()
            if rank == 0:
                print(f"graph={graph} front={front} world={world}: losses {np.round(losses, 6).tolist()} captured={eng.graph is not None}",
                      flush=True)
            del eng
            torch.cuda.empty_cache()

if world == 1:
    scheds = {"plain": False, "serial": "serial", "overlap": True}
    names = (args.schedules or ("plain,serial,overlap" if args.workload == "c3" else "plain")).split(",")
    fronts = (args.fronts or "auto").split(",")
    steps = args.steps or {"c4": 120, "c5": 24}.get(args.workload, 800)
    warm = max(bench.EPOCH_BATCHES, steps // 8)
    for G in [int(g) for g in args.shards.split(",")]:
        for front in fronts:
            for name in names:
                if scheds[name] and front == "sharded":
                    continue  # (two schedules of the same exchange: the engine refuses the combination)
                eng, data, params = bench.build_engine(0, G, True, force_exchange=True, overlap_exchange=scheds[name], front=front, **ekw)
                # warm until the clock has ramped (a first configuration timed right after its captures reads up to 40 % slow:
                # c4 1/8 at 519-544 us instead of 373-381), then the median of three blocks
                tw = time.perf_counter()
                while True:
                    bench.run_steps(eng, warm, 0)
                    torch.cuda.synchronize()
                    if time.perf_counter() - tw > 0.4:
                        break
                dts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    bench.run_steps(eng, steps, 0)
                    torch.cuda.synchronize()
                    dts.append((time.perf_counter() - t0) / steps)
                dt = sorted(dts)[1]
                print(f"{args.workload} {eng.precision} emulated shard 1/{G}, front {'sharded' if eng.front_sharded else 'replicated':10s} "
                      f"exchange schedule {name:7s}: {dt * 1e6:.1f} us per step per rank ({1.0 / dt:.0f} steps/s if collectives were free)",
                      flush=True)
                del eng
                torch.cuda.empty_cache()
dist.barrier()
dist.destroy_process_group()
