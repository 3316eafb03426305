#!/usr/bin/env python3
"""gzip on the device (rsq_sim_gzip_device) on the simulator's own FASTQ text: python tools/bench_gzip.py [pairs] -- text bytes, compressed bytes, GB/s by the
kernels' time ("gzip") and by the call, zlib's levels 1 and 6 on a 16 MB sample for comparison.  One JSON line."""
import json
import os
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from reseq_amd import api, synth, workloads  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
tmp = tempfile.mkdtemp(prefix="rsq_gz_")
ppath, fpath = os.path.join(tmp, "p0.rsqp"), os.path.join(tmp, "ref.fa")
workloads.p0_profile(ppath)
genome = max(200_000, int(4_641_652 * pairs / 10_000_000))
synth.write_fasta(fpath, synth.make_reference(2, [genome], gc=0.508, names=[f"synthEcoli0 len={genome}"]))
prof, ref = api.Profile(ppath), api.Reference(fpath, 11)
sim = api.Simulator(prof, ref, 0)
info = sim.prepare(11, pairs)
n, l1, l2, _ = sim.pairs_device(1, info.total_blocks + 1, None, None)
r1, r2 = api.DeviceArray(0, l1 + 64), api.DeviceArray(0, l2 + 64)
n, l1, l2, rc = sim.pairs_device(1, info.total_blocks + 1, r1, r2)
assert rc == api.RSQ_OK
out = api.DeviceArray(0, l1 // 2)
sim.gzip_device(r1, l1, out, out.nbytes)
runs = []
for _ in range(3):
    t0 = time.perf_counter()
    size, rc = sim.gzip_device(r1, l1, out, out.nbytes)
    runs.append((time.perf_counter() - t0, sim.last_kernel_ms("gzip") / 1e3))
    assert rc == api.RSQ_OK
sample = r1.to_numpy(np.uint8, min(l1, 16 << 20)).tobytes()
t0 = time.perf_counter()
z1 = len(zlib.compress(sample, 1))
t_z1 = time.perf_counter() - t0
wall, kernel = min(runs)
print(json.dumps({"pairs": n, "text_bytes": l1, "members_bytes": size, "ratio": l1 / size, "zlib_1_ratio": len(sample) / z1, "zlib_6_ratio": len(sample) / len(zlib.compress(sample, 6)),
                  "size_over_zlib_1": (size / l1) / (z1 / len(sample)), "gbytes_per_s_kernels": l1 / kernel / 1e9, "gbytes_per_s_call": l1 / wall / 1e9,
                  "zlib_1_one_thread_mbytes_per_s": len(sample) / t_z1 / 1e6, "seconds": [round(w, 4) for w, _ in runs]}))
