#!/usr/bin/env python3
"""What a call of the records kernel costs by its size: rsq_sim_error_model_fasta on n records of FASTA text resident in HBM, n from one chunk of 64 to 1.7 M:
milliseconds of the read kernel ("fill_reads") and of the whole call, best of 5.  One JSON line.  Usage: python tools/time_small_calls.py [--option name=value ...]"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from reseq_amd import api, workloads  # noqa: E402

if "--lib" in sys.argv:
    api.use_library(sys.argv[sys.argv.index("--lib") + 1])
for k, a in enumerate(sys.argv):
    if a == "--option":
        name, value = sys.argv[k + 1].split("=")
        api.set_option(name, int(value))
tmp = tempfile.mkdtemp(prefix="rsq_small_")
ppath = os.path.join(tmp, "p0.rsqp")
arrays = workloads.p0_profile(ppath)
N = 1_709_824
rows, _ = workloads.seq_to_illumina_rows(N, arrays)
W = rows.shape[1]
d_text = api.DeviceArray.from_numpy(0, np.concatenate([rows.reshape(-1), np.zeros(8, np.uint8)]))
prof = api.Profile(ppath)
sim = api.Simulator(prof, None, 0)
sim.prepare(11)
d_out = api.DeviceArray(0, N * 400)
need, k, used = C.c_size_t(0), C.c_uint64(0), C.c_size_t(0)
out = {}
sizes = [int(x) for x in sys.argv[sys.argv.index("--sizes") + 1].split(",")] if "--sizes" in sys.argv else (64, 1024, 16384, 53440, 106880, 213760, 427520, 1_709_824)
for n in sizes:
    best = None
    for _ in range(6):
        t0 = time.perf_counter()
        api._check(api.lib().rsq_sim_error_model_fasta(sim.h, 0, d_text.ptr, n * W, 1, d_out.ptr, d_out.nbytes, C.byref(need), C.byref(k), C.byref(used), None))
        wall = (time.perf_counter() - t0) * 1e3
        ms = {name: sim.last_kernel_ms(name) for name in ("parse_records", "fill_reads", "format_write")}
        if best is None or wall < best[0]:
            best = (wall, ms)
    assert k.value == n
    out[str(n)] = {"call_ms": round(best[0], 3), **{a: round(b, 3) for a, b in best[1].items()}, "reads_per_s_call": round(n / best[0] * 1e3)}
print(json.dumps(out))
