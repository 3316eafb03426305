#!/usr/bin/env python3
"""BASELINE.json configs[2] restated, scaled to N records: seqToIllumina's hot path
(rsq_sim_error_model: ApplyErrorsAndQualityToFastaInput with the header fields already parsed) on templates resident in HBM, in calls
of at most 10 M records (the same templates with other record indices, i.e. other random streams, until N records are done); the
second figure is rsq_sim_error_model_fastq, which also formats the FASTQ text on the device.
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

total = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
n = min(total, 10_000_000)
calls = (total + n - 1) // n
tmp = tempfile.mkdtemp(prefix="rsq_em_")
ppath = os.path.join(tmp, "p0.rsqp")
# `python tools/bench_error_model.py N binned`: the same profile with 12 quality values (30 .. 41; K <= 12 takes the 3-quad instantiation of the
# read kernels; api.set_option("min_quality_quads", 10) forces the 10-quad one for comparison)
CFG = dict(synth.P0, name="P0b", qual_from=30, qual_to=42) if len(sys.argv) > 2 and sys.argv[2] == "binned" else synth.P0
arrays = synth.make_profile(CFG, seed=103741084)
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


ids = np.char.add("read", np.arange(n).astype(str)).astype("S")
id_len = np.char.str_len(ids).astype(np.uint64)
id_off = np.zeros(n + 1, np.uint64)
id_off[1:] = np.cumsum(id_len)
blob = b"".join(ids.tolist()) + b"\0"
d_ids, d_off = api.DeviceArray.from_numpy(dev, np.frombuffer(blob, np.uint8)), api.DeviceArray.from_numpy(dev, id_off)
text = api.DeviceArray(dev, n * (2 * 150 + 64) + int(id_off[-1]))


def once():
    t0 = time.perf_counter()
    for c in range(calls):
        api._check(api.lib().rsq_sim_error_model(sim.h, c * n, n, 150, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, ins[4].ptr, outs[0].ptr, outs[1].ptr, out_stride,
                                                 outs[2].ptr, outs[3].ptr, outs[4].ptr, outs[5].ptr, cigar_stride, None))
    return time.perf_counter() - t0


def once_text():
    need = C.c_size_t(0)
    t0 = time.perf_counter()
    for c in range(calls):
        api._check(api.lib().rsq_sim_error_model_fastq(sim.h, c * n, n, 150, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, ins[4].ptr, d_ids.ptr, d_off.ptr, text.ptr,
                                                       text.nbytes, C.byref(need), None))
    return time.perf_counter() - t0, need.value


once()
ts = [once() for _ in range(3)]
best = min(ts)
fill_ms = sim.last_kernel_ms("fill_reads")
once_text()
tt = [once_text() for _ in range(3)]
best_text = min(t for t, _ in tt)
# the same records as FASTA text resident in HBM, parsed on the device (rsq_sim_error_model_fasta): in one call (below 4 GB of text) and in blocks of 48 MB as the
# command line hands them over
rec["frag_len"] = np.clip(rec["frag_len"], 100, 999).astype(np.uint32)
n_text = min(n, 8_000_000)
rows = synth.fixed_width_fasta({k: v[:n_text] for k, v in rec.items()})
synth.number_rows(rows, 0)
W = rows.shape[1]
d_fasta = api.DeviceArray.from_numpy(dev, np.concatenate([rows.reshape(-1), np.zeros(8, np.uint8)]))
del rows


def once_fasta(block_records):
    need, k, used = C.c_size_t(0), C.c_uint64(0), C.c_size_t(0)
    ms = {"parse_records": 0.0, "fill_reads": 0.0, "format_write": 0.0}
    t0 = time.perf_counter()
    for first in range(0, n_text, block_records):
        m = min(block_records, n_text - first)
        api._check(api.lib().rsq_sim_error_model_fasta(sim.h, first, C.c_void_p(d_fasta.ptr.value + first * W), m * W, 1, text.ptr, text.nbytes, C.byref(need), C.byref(k), C.byref(used), None))
        assert k.value == m and used.value == m * W
        for name in ms:
            ms[name] += sim.last_kernel_ms(name)
    return time.perf_counter() - t0, ms


def callers_side_by_side(block_records, n_callers):
    """The same blocks handed to `n_callers` simulators of the profile, each on its own thread, stream and output buffer (a call is synchronous: a caller with
    small blocks keeps the device busy by having more than one in flight).  Caller c takes blocks c, c + n_callers, ..."""
    import threading
    sims, streams, outs = [], [], []
    for _ in range(n_callers):
        sm = api.Simulator(prof, None, dev)
        sm.prepare(11)
        st = C.c_void_p()
        api._check(api.lib().rsq_stream_create(dev, C.byref(st)))
        sims.append(sm)
        streams.append(st)
        outs.append(api.DeviceArray(dev, block_records * (W + 64)))
    starts = list(range(0, n_text, block_records))

    def work(c):
        need, k, used = C.c_size_t(0), C.c_uint64(0), C.c_size_t(0)
        for first in starts[c::n_callers]:
            m = min(block_records, n_text - first)
            api._check(api.lib().rsq_sim_error_model_fasta(sims[c].h, first, C.c_void_p(d_fasta.ptr.value + first * W), m * W, 1, outs[c].ptr, outs[c].nbytes, C.byref(need), C.byref(k),
                                                           C.byref(used), streams[c]))
            assert k.value == m and used.value == m * W

    def once():
        th = [threading.Thread(target=work, args=(c,)) for c in range(n_callers)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    once()
    secs = [once() for _ in range(3)]
    for sm, st in zip(sims, streams):
        sm.close()
        api.lib().rsq_stream_destroy(dev, st)
    return {"records_per_call": block_records, "callers": n_callers, "seconds": secs, "reads_per_s": n_text / min(secs)}


fasta = {}
for label, block_records in (("one_call", n_text), ("blocks_of_48_MB", (48 << 20) // W), ("blocks_of_192_MB", (192 << 20) // W)):
    once_fasta(block_records)
    runs = [once_fasta(block_records) for _ in range(3)]
    t, ms = min(runs, key=lambda r: r[0])
    fasta[label] = {"records_per_call": block_records, "calls": -(-n_text // block_records), "seconds": [r[0] for r in runs], "reads_per_s": n_text / t, "kernel_ms_summed": ms}
for callers in (2, 4):
    fasta[f"blocks_of_48_MB_{callers}_callers"] = callers_side_by_side((48 << 20) // W, callers)
print(json.dumps({"config": "configs[2] seqToIllumina, templates and outputs resident in HBM", "from_fasta_text_parsed_on_device": dict(fasta, records=n_text, text_bytes=n_text * W), "profile": CFG["name"], "quality_values": CFG["qual_to"] - CFG["qual_from"], "records": n * calls, "records_per_call": n, "read_len": 150, "seconds": ts,
                  "reads_per_s": n * calls / best, "fill_kernel_ms_last_call": fill_ms, "with_fastq_text_on_device": {"seconds": [t for t, _ in tt], "reads_per_s": n * calls / best_text,
                                                                                                                  "text_bytes_per_call": tt[0][1],
                                                                                                                  "format_ms_last_call": sim.last_kernel_ms("format_write")}}))

