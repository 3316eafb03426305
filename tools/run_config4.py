#!/usr/bin/env python3
"""BASELINE.json configs[3] restated on ONE GPU: a Drosophila-sized 143.7 Mb reference in 7
sequences plus 1000 short scaffolds (shorter than the longest insert, so without units), profile P0, coverage 30
(about 14.4 M pairs).  Prints one JSON line: sizes, pre-pass and generation times, checksums.  Not a bench line: a full-size
run of the path that checks that nothing overflows and that batching does not change the output."""
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from reseq_amd import api, synth  # noqa: E402

from reseq_amd import workloads  # noqa: E402

tmp = tempfile.mkdtemp(prefix="rsq_c4_")
ppath = os.path.join(tmp, "p0.rsqp")
workloads.p0_profile(ppath)
t0 = time.perf_counter()
fpath, lengths = workloads.drosophila_sized(tmp)                    # the inputs' one definition (bench.py's other_configs leg runs the same)
t_make = time.perf_counter() - t0
prof, ref = api.Profile(ppath), api.Reference(fpath, 7)
sim = api.Simulator(prof, ref, 0)
t0 = time.perf_counter()
info = sim.prepare(7, 0, 30.0)
t_prep = time.perf_counter() - t0
nb = info.total_blocks
r1 = r2 = None
out = {}
for name, batch in (("batch_25000", 25000), ("batch_7777", 7777)):
    h1, h2, n, nbytes = hashlib.sha256(), hashlib.sha256(), 0, 0
    t0 = time.perf_counter()
    t_gpu = 0.0
    kernel_ms = {}
    for lo in range(1, nb + 1, batch):
        hi = min(nb + 1, lo + batch)
        if r1 is None:
            _, l1, l2, _ = sim.pairs_device(lo, hi, None, None)
            r1, r2 = api.DeviceArray(0, int(l1 * 1.3) + 4096), api.DeviceArray(0, int(l2 * 1.3) + 4096)
        t1 = time.perf_counter()
        k, l1, l2, rc = sim.pairs_device(lo, hi, r1, r2)
        t_gpu += time.perf_counter() - t1
        if rc != api.RSQ_OK:
            raise SystemExit(f"rc {rc}: {api.lib().rsq_last_error().decode()}")
        for key in ("sieve", "sieve_screen", "sieve_emit", "fill_reads", "format_write", "scan"):
            kernel_ms[key] = kernel_ms.get(key, 0.0) + sim.last_kernel_ms(key)
        n += k
        nbytes += l1 + l2
        h1.update(r1.to_numpy(np.uint8, l1).tobytes())
        h2.update(r2.to_numpy(np.uint8, l2).tobytes())
    out[name] = {"pairs": n, "fastq_bytes": nbytes, "sha256_r1": h1.hexdigest(), "sha256_r2": h2.hexdigest(), "gpu_s": t_gpu, "kernel_ms": {k: round(v, 1) for k, v in kernel_ms.items()},
                 "wall_s_with_download_and_hash": time.perf_counter() - t0}
same = out["batch_25000"]["sha256_r1"] == out["batch_7777"]["sha256_r1"] and out["batch_25000"]["sha256_r2"] == out["batch_7777"]["sha256_r2"]
print(json.dumps({"config": "configs[3] Drosophila-sized, 1 GPU", "reference_bp": int(sum(lengths)), "sequences": len(lengths), "total_blocks": nb,
                  "pairs_requested_from_coverage_30": info.total_pairs, "adapter_only_pairs": info.adapter_only_pairs, "prepare_s": t_prep,
                  "sys_chain_passes": info.sys_chain_passes, "runs": out, "batching_invariant": same,
                  "pairs_per_s_gpu": out["batch_25000"]["pairs"] / out["batch_25000"]["gpu_s"]}))
sys.exit(0 if same else 1)
