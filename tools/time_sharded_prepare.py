#!/usr/bin/env python3
"""Per-rank cost of the sharded pre-pass (sharding.sharded_prepare) on the Drosophila-sized reference of tools/run_config4.py, measured on
ONE GPU: for world = 1, 2, 4, 8 the ranks' simulators live in this process and make the protocol's calls in lock step; the time every
rank spends in its own calls is what that rank would spend on its own GPU (the collectives carry a few MB at most and are not timed).
Also checks that a rank's block range simulates to the same FASTQ text as after the whole pre-pass.  Prints one JSON line."""
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from reseq_amd import api, sharding, synth  # noqa: E402

BIG = [32_079_331, 28_110_227, 25_286_936, 23_542_271, 23_513_712, 7_350_000 + 3_667_352 - 1_000 * 600, 1_348_131]
tmp = tempfile.mkdtemp(prefix="rsq_sp_")
ppath, fpath = os.path.join(tmp, "p0.rsqp"), os.path.join(tmp, "ref.fa")
synth.write_profile(ppath, synth.make_profile(synth.P0, seed=103741084))
lengths = BIG + [600] * 1000
synth.write_fasta(fpath, synth.make_reference(5, lengths, gc=0.42))
prof, ref = api.Profile(ppath), api.Reference(fpath, 7)


class Rank:
    """api.Simulator behind the interface sharding.sharded_prepare_in_process drives, with a stopwatch on every call"""

    def __init__(self):
        self.sim = api.Simulator(prof, ref, 0)
        self.seq_len = lengths
        self.seconds = {}

    def _timed(self, name, f, *a):
        t0 = time.perf_counter()
        r = f(*a)
        self.seconds[name] = self.seconds.get(name, 0.0) + time.perf_counter() - t0
        return r

    def ref_seq_bias(self):
        return self.sim.ref_seq_bias(len(lengths))

    def prepare_plan(self, *a):
        return self._timed("plan", self.sim.prepare_plan, *a)

    def bias_partials(self, lo, hi):
        return self._timed("bias_partials", self.sim.bias_partials, lo, hi)

    def prepare_normalization(self, s, m):
        return self._timed("normalization", self.sim.prepare_normalization, s, m)

    def prepare_sys_errors(self, lo, hi, st):
        return self._timed("sys_errors", self.sim.prepare_sys_errors, lo, hi, st)

    def prepare_finish(self):
        return self._timed("finish", self.sim.prepare_finish)


def text_hash(sim, lo, hi):
    n, l1, l2, _ = sim.pairs_device(lo, hi, None, None)
    r1, r2 = api.DeviceArray(0, l1 + 4096), api.DeviceArray(0, l2 + 4096)
    n, l1, l2, rc = sim.pairs_device(lo, hi, r1, r2)
    assert rc == api.RSQ_OK
    h = hashlib.sha256(r1.to_numpy(np.uint8, l1).tobytes() + r2.to_numpy(np.uint8, l2).tobytes()).hexdigest()
    r1.free()
    r2.free()
    return n, h


whole = api.Simulator(prof, ref, 0)
whole.prepare(7, 0, 30.0)                                     # warm-up (code objects, allocations)
t0 = time.perf_counter()
winfo = whole.prepare(7, 0, 30.0)
whole_s = time.perf_counter() - t0
out = {"config": "pre-pass of configs[3] (Drosophila-sized, 144.9 Mb, 1007 sequences), ranks emulated in one process on one GPU", "whole_prepare_s": whole_s, "worlds": {}}
for world in (1, 2, 4, 8):
    ranks = [Rank() for _ in range(world)]
    infos, ranges, rounds = sharding.sharded_prepare_in_process(ranks, 7, 0, 30.0)
    per_rank = [round(sum(r.seconds.values()), 4) for r in ranks]
    # spot check: the middle 300 blocks of the last rank's range simulate to the same text as after the whole pre-pass
    lo, hi = ranges[-1]
    a, b = lo + (hi - lo) // 2, min(hi, lo + (hi - lo) // 2 + 300)
    same = text_hash(ranks[-1].sim, a, b) == text_hash(whole, a, b)
    same = same and np.array_equal(ranks[0].sim.thresholds(), whole.thresholds())
    out["worlds"][str(world)] = {"per_rank_s": per_rank, "slowest_rank_s": max(per_rank), "stages_of_slowest": {k: round(v, 4) for k, v in ranks[int(np.argmax(per_rank))].seconds.items()},
                                 "chain_exchange_rounds": rounds, "equal_to_whole_pre_pass": bool(same)}
    for r in ranks:
        r.sim.close()
print(json.dumps(out))
