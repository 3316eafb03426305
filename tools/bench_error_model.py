#!/usr/bin/env python3
"""BASELINE.json configs[2] restated, scaled to N records: seqToIllumina's hot path
(rsq_sim_error_model: ApplyErrorsAndQualityToFastaInput with the header fields already parsed) on templates resident in HBM.
Prints one JSON line with reads/s.  Not the bench line (bench.py measures the illuminaPE metric)."""
import json
import os
import sys
import tempfile
import time
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from reseq_amd import api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
tmp = tempfile.mkdtemp(prefix="rsq_em_")
ppath = os.path.join(tmp, "p0.rsqp")
arrays = synth.make_profile(synth.P0, seed=103741084)
synth.write_profile(ppath, arrays)
rec = synth.make_error_model_input(3, n, 150, arrays, zero_frac=0.97)
prof = api.Profile(ppath)
sim = api.Simulator(prof, None, 0)
sim.prepare(11)
dev = 0
ins = [api.DeviceArray.from_numpy(dev, np.ascontiguousarray(rec[k], dt)) for k, dt in
       (("seqs", np.uint8), ("seg", np.uint8), ("frag_len", np.uint32), ("dom", np.uint8), ("rate", np.uint8))]
out_stride, cigar_stride = 160, 64
outs = [api.DeviceArray(dev, n * out_stride), api.DeviceArray(dev, n * out_stride), api.DeviceArray(dev, n * 2), api.DeviceArray(dev, n * 2), api.DeviceArray(dev, n * 2),
        api.DeviceArray(dev, n * cigar_stride)]


def once():
    t0 = time.perf_counter()
    api._check(api.lib().rsq_sim_error_model(sim.h, 0, n, 150, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, ins[4].ptr, outs[0].ptr, outs[1].ptr, out_stride, outs[2].ptr,
                                             outs[3].ptr, outs[4].ptr, outs[5].ptr, cigar_stride, None))
    return time.perf_counter() - t0


once()
ts = [once() for _ in range(3)]
best = min(ts)
print(json.dumps({"config": "configs[2] seqToIllumina, templates and outputs resident in HBM", "records": n, "read_len": 150, "seconds": ts, "reads_per_s": n / best,
                  "fill_kernel_ms": sim.last_kernel_ms("fill_reads")}))
