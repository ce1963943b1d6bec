#!/usr/bin/env python3
"""Steady-state step rate of VGAN_no_kl.fit itself (not of the engine loop bench.py times): two fits of E1 and E2 epochs on the c3
synthetic data, rate = (E2 - E1) * batches / (t2 - t1).   python3 tools/fit_rate.py [E1 E2]
VGAN_FIT_SYNC_EACH_EPOCH=1 reads every epoch's loss before launching the next epoch (the behaviour of rounds 1-2)."""
import io, os, sys, time, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import vgan_amd
from vgan_amd import synth
from src.vgan import VGAN_no_kl
from src.models.Mmd_loss_constrained import MMDLossConstrained

E1, E2 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 500)
X = synth.synthetic_dataset("c3")
nb = X.shape[0] // 1024
for verbose in (False, True):
    ts = []
    for E in (E1, E2, E1, E2):
        MMDLossConstrained.__init__.__defaults__[0].bandwidth = None
        m = VGAN_no_kl(batch_size=1024, epochs=E, seed=777)
        m.verbose = verbose
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            m.fit(X)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    r = [(E2 - E1) * nb / (ts[i + 1] - ts[i]) for i in (0, 2)]
    print(f"verbose={verbose}: fit of {E1} / {E2} epochs x {nb} steps: {ts[0]:.3f} / {ts[1]:.3f} s -> {r[0]:.0f}, {r[1]:.0f} steps/s inside fit; "
          f"last losses {m.train_history['generator_loss'][-1]:.6f}")
