#!/usr/bin/env python3
"""How fast N processes write their shares of a job: into ONE pair of output files at their offsets (what a drop-in for `reseq illuminaPE -1 a.fq -2 b.fq` must
do) against one pair of files per process.  Every process simulates the bench workload (10 M pairs, 7.4 GB of text) on GPU 0 with the text kept in HBM
(rsq_sim_job_generate), all wait for each other, then write (rsq_sim_job_write).  Modes: `shared` = buffered pwrite into one pair of files, `shared_direct` = the same
with option job_write_direct (whole 4 KB blocks around the page cache, files allocated beforehand), `split` = a pair of files per process.
Usage: python tools/measure_rank_writes.py [processes] [directory, default /dev/shm] [pairs per process, default 10000000]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = r"""
import os, sys, time, json
sys.path.insert(0, os.environ["RSQ_ROOT"])
import torch                                        # the HIP runtime torch ships is the one the library binds to
from reseq_amd import api
rank, world, work, mode = int(os.environ["RANK"]), int(os.environ["WORLD"]), os.environ["WORK"], os.environ["MODE"]
if mode == "shared_direct":
    api.set_option("job_write_direct", 1)
prof, ref = api.Profile(os.path.join(work, "p0.rsqp")), api.Reference(os.path.join(work, "ref.fa"), 11)
sim = api.Simulator(prof, ref, 0)
info = sim.prepare(11, int(os.environ["PAIRS"]))
n, b1, b2 = sim.job_generate(1, info.total_blocks + 1)
if mode.startswith("shared") and rank == 0:                # the files exist at their final size, blocks allocated, before anybody writes
    for name, size in (("shared_1.fq", b1 * world), ("shared_2.fq", b2 * world)):
        with open(os.path.join(work, name), "wb") as f:
            try:
                os.posix_fallocate(f.fileno(), 0, size)
            except OSError:
                f.truncate(size)
open(os.path.join(work, f"ready{mode}{rank}"), "w").close()
while not all(os.path.exists(os.path.join(work, f"ready{mode}{r}")) for r in range(world)):
    time.sleep(0.001)
if mode.startswith("shared"):
    p1, p2, o1, o2 = os.path.join(work, "shared_1.fq"), os.path.join(work, "shared_2.fq"), rank * b1, rank * b2      # every process has the same job: equal sizes
else:
    p1, p2, o1, o2 = os.path.join(work, f"split{rank}_1.fq"), os.path.join(work, f"split{rank}_2.fq"), 0, 0
t0 = time.perf_counter()
sim.job_write(p1, o1, p2, o2)
dt = time.perf_counter() - t0
print(json.dumps({"rank": rank, "bytes": b1 + b2, "write_s": dt}))
"""


def main():
    from reseq_amd import synth
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    where = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
    pairs = sys.argv[3] if len(sys.argv) > 3 else "10000000"
    work = tempfile.mkdtemp(prefix="rsq_rw_", dir=where)
    synth.write_profile(os.path.join(work, "p0.rsqp"), synth.make_profile(synth.P0, seed=103741084))
    synth.write_fasta(os.path.join(work, "ref.fa"), synth.make_reference(2, [4_641_652], gc=0.508, names=["synthEcoli0 len=4641652"]))
    fs = subprocess.run(["df", "-T", where], capture_output=True, text=True).stdout.strip().split("\n")[-1]
    out = {"processes": world, "directory": where, "file_system": fs, "pairs_per_process": int(pairs)}
    for mode in ("shared", "shared_direct", "split"):
        env = dict(os.environ, RSQ_ROOT=ROOT, WORLD=str(world), WORK=work, MODE=mode, PAIRS=pairs)
        procs = [subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
        res = []
        for p in procs:
            so, se = p.communicate(timeout=1200)
            if p.returncode:
                raise SystemExit(se[-2000:])
            res.append(json.loads(so.strip().split("\n")[-1]))
        total, slowest = sum(r["bytes"] for r in res), max(r["write_s"] for r in res)
        out[mode] = {"bytes": total, "slowest_write_s": round(slowest, 3), "gbytes_per_s_job": round(total / slowest / 1e9, 2), "per_process_s": [round(r["write_s"], 3) for r in res]}
        for f in os.listdir(work):
            if f.endswith(".fq") or f.startswith("ready"):
                os.remove(os.path.join(work, f))
        os.sync()
    print(json.dumps(out))
    for f in os.listdir(work):
        os.remove(os.path.join(work, f))
    os.rmdir(work)


if __name__ == "__main__":
    main()
