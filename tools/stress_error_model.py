#!/usr/bin/env python3
"""Randomised parity stress of the seqToIllumina path (rsq_sim_error_model against the oracle): 30 record sets of random size, template
length, seed and profile.  Run on a GPU box."""
import sys, pathlib, tempfile
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np
import parity_cases as P
from backends import GpuBackend
from reseq_amd import synth
wd=pathlib.Path(tempfile.mkdtemp())
rng=np.random.default_rng(77)
for t in range(30):
    cfg = synth.P0 if t % 5 == 4 else synth.TINY
    rl = int(rng.integers(5, 80)) if cfg is synth.TINY else int(rng.integers(50, 220))
    P._error_model(GpuBackend, wd, f"em{t}", cfg, int(rng.integers(1, 900)), rl, seed=int(rng.integers(1,1<<40)), prof_seed=int(rng.integers(1,500)), zero_frac=float(rng.random()))
print("error-model stress: 30 trials ok")
