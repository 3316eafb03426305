#!/usr/bin/env python3
"""BASELINE.json configs[4] (human-sized reference, phased VCF with substitutions and short
insertions / deletions, methylation BED, coverage 30) on ONE GPU at a chosen scale (default 1/10: 310 Mb in 24 sequences, 0.4 M
substitutions + 40 k insertions / deletions of at most 20 bases on two alleles, 2 M unmethylated regions with Beta(0.5, 0.5)
methylation, about 31 M pairs).  Prints one JSON line: sizes, pre-pass and generation times per stage, a checksum, and that a
second batching writes the same bytes (pairs, total bytes, SHA-256 of the first 48000 blocks).  Not a bench line."""
import hashlib
import math
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from reseq_amd import api, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
# `snv`: substitutions only (the commonest kind of call set: the sieve takes the allele-copy route, k_sieve_finish<1, 8>), no methylation file; `--lib path`: another build
snv_only = "snv" in sys.argv[2:]
if "--lib" in sys.argv:
    api.use_library(sys.argv[sys.argv.index("--lib") + 1])
for k, a in enumerate(sys.argv):                                     # --option name=value: rsq_set_option (e.g. trace_prepare=1: stage times of the pre-pass on stderr)
    if a == "--option":
        api.set_option(*[(n, int(v)) for n, v in [sys.argv[k + 1].split("=")]][0])
from reseq_amd import workloads  # noqa: E402

tmp = tempfile.mkdtemp(prefix="rsq_c5_")
ppath = os.path.join(tmp, "p0.rsqp")
workloads.p0_profile(ppath)
t0 = time.perf_counter()
job_in = workloads.human_sized(tmp, scale, snv_only)                 # the inputs' one definition (bench.py's other_configs leg runs the same at scale 0.1)
fpath, vpath, bpath, lengths = job_in["fasta"], job_in["vcf"], job_in["bed"], job_in["lengths"]
total, n_sub, n_indel, n_regions = int(sum(lengths)), job_in["substitutions"], job_in["indels"], job_in["regions"]
t_make = time.perf_counter() - t0

# `loads N`: N processes load the job's inputs at the same time, as the ranks of an N-GPU job on one host do (here all of them onto this one GPU): seconds per process and stage
if "loads" in sys.argv[2:]:
    import subprocess
    n_proc = int(sys.argv[sys.argv.index("loads") + 1])
    code = ("import sys, time, json; sys.path.insert(0, %r); from reseq_amd import api; api.set_option('trace_load', %d); t0 = time.perf_counter(); st = {}; t = time.perf_counter()\n"
            "prof, ref = api.Profile(%r), api.Reference(%r, 7); st['fasta_and_replace_n'] = round(time.perf_counter() - t, 2); t = time.perf_counter()\n"
            "ref.read_variants(%r); st['vcf'] = round(time.perf_counter() - t, 2); t = time.perf_counter()\n"
            "sim = api.Simulator(prof, ref, 0); st['create_simulator_pack_upload'] = round(time.perf_counter() - t, 2); t = time.perf_counter()\n"
            "sim.read_methylation(%r); st['methylation_bed'] = round(time.perf_counter() - t, 2)\n"
            "print(json.dumps({'load_s': round(time.perf_counter() - t0, 2), 'stages': st}))\n") % (ROOT, int("trace" in sys.argv), ppath, fpath, vpath, bpath)
    result = {}
    for n in sorted({1, n_proc}):
        procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True) for _ in range(n)]
        result[str(n)] = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    # one load per host (rsq_sim_export_reference / rsq_sim_import_reference): process 0 loads as above and exports to /dev/shm, the others wait for the file and
    # import it -- seconds from the common start until every process holds its simulator
    shared = os.path.join("/dev/shm", f"rsq_c5_packed_{os.getpid()}")
    code_shared = ("import sys, os, time, json; sys.path.insert(0, %r); from reseq_amd import api; rank = int(sys.argv[1]); t0 = time.perf_counter(); st = {}; t = t0\n"
                   "prof = api.Profile(%r)\n"
                   "if rank == 0:\n"
                   "    ref = api.Reference(%r, 7); ref.read_variants(%r); sim = api.Simulator(prof, ref, 0); sim.read_methylation(%r); st['load'] = round(time.perf_counter() - t, 2); t = time.perf_counter()\n"
                   "    sim.export_reference(%r); st['export'] = round(time.perf_counter() - t, 2)\n"
                   "else:\n"
                   "    sim = api.Simulator(prof, None, 0); st['create'] = round(time.perf_counter() - t, 2); t = time.perf_counter()\n"
                   "    while not os.path.exists(%r): time.sleep(0.002)\n"
                   "    st['wait_for_the_loader'] = round(time.perf_counter() - t, 2); t = time.perf_counter()\n"
                   "    sim.import_reference(%r); st['import'] = round(time.perf_counter() - t, 2)\n"
                   "print(json.dumps({'ready_s': round(time.perf_counter() - t0, 2), 'stages': st, 'file_bytes': os.path.getsize(%r)}))\n") % (ROOT, ppath, fpath, vpath, bpath, shared, shared, shared, shared)
    procs = [subprocess.Popen([sys.executable, "-c", code_shared, str(r)], stdout=subprocess.PIPE, text=True) for r in range(n_proc)]
    one_load = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    os.unlink(shared)
    print(json.dumps({"config": f"configs[4] human-sized at scale {scale}: the load of N ranks at once on one host", "host_threads": os.cpu_count(), "loads": result,
                      "one_load_per_host": {str(n_proc): one_load}}))
    sys.exit(0)

t0 = time.perf_counter()
load_stages = {}
t1 = time.perf_counter()
prof, ref = api.Profile(ppath), api.Reference(fpath, 7)
load_stages["fasta_and_replace_n"] = round(time.perf_counter() - t1, 2)
t1 = time.perf_counter()
alleles = ref.read_variants(vpath)
load_stages["vcf"] = round(time.perf_counter() - t1, 2)
t1 = time.perf_counter()
sim = api.Simulator(prof, ref, 0)
load_stages["create_simulator_pack_upload"] = round(time.perf_counter() - t1, 2)
t1 = time.perf_counter()
if not snv_only:
    sim.read_methylation(bpath)
load_stages["methylation_bed"] = round(time.perf_counter() - t1, 2)
t_load = time.perf_counter() - t0
# `prepare_sweep chunk:warmup,chunk:warmup,...`: the pre-pass timed under several chunk lengths / run-ups of the systematic-error chains, nothing else
if "prepare_sweep" in sys.argv[2:]:
    for combo in sys.argv[sys.argv.index("prepare_sweep") + 1].split(","):
        chunk, warmup = (int(v) for v in combo.split(":"))
        api.set_option("chain_chunk", chunk)
        api.set_option("chain_warmup", warmup)
        sim.take_options()
        t0 = time.perf_counter()
        sim.prepare(7, 0, 30.0)
        print(json.dumps({"chain_chunk": chunk, "chain_warmup": warmup, "prepare_s": round(time.perf_counter() - t0, 3)}), flush=True)
    sys.exit(0)
t0 = time.perf_counter()
info = sim.prepare(7, 0, 30.0)
t_prep = time.perf_counter() - t0
nb = info.total_blocks
out = {}
r1 = r2 = None
# `batches a,b`: blocks of 1000 start positions per rsq_sim_pairs call of the two runs (default 24000 and 12000; rsq_sim_job_generate takes about 12 M pairs per call: 120000 at coverage 30)
batches = [int(x) for x in sys.argv[sys.argv.index("batches") + 1].split(",")] if "batches" in sys.argv[2:] else [24000, 12000]
# the checksum covers the blocks up to a border EVERY batching has a call ending at -- the least common multiple of the batch sizes (the whole job if that lies beyond
# it): the text is hundreds of GB at full scale, and a run that hashed nothing must not report the hash of nothing (round 5's file did)
hash_border = math.lcm(*batches)
if hash_border >= nb:
    hash_border = nb
for name, batch in [(f"batch_{b}", b) for b in batches]:
    h1, h2 = hashlib.sha256(), hashlib.sha256()
    n = nbytes = hashed_to = hashed_bytes = 0
    t_gpu = 0.0
    kernel_ms = {}
    for lo in range(1, nb + 1, batch):
        hi = min(nb + 1, lo + batch)
        if r1 is None:
            _, l1, l2, _ = sim.pairs_device(lo, hi, None, None)
            r1, r2 = api.DeviceArray(0, int(l1 * 1.5) + 4096), api.DeviceArray(0, int(l2 * 1.5) + 4096)
        t1 = time.perf_counter()
        k, l1, l2, rc = sim.pairs_device(lo, hi, r1, r2)
        t_gpu += time.perf_counter() - t1
        if rc != api.RSQ_OK:
            raise SystemExit(f"rc {rc}: {api.lib().rsq_last_error().decode()}")
        for key in ("slot_table", "variant_templates", "sieve", "sieve_screen", "sieve_emit", "fill_reads", "format_write", "scan"):
            if sim.last_kernel_launches(key):
                kernel_ms[key] = kernel_ms.get(key, 0.0) + sim.last_kernel_ms(key)
        n += k
        nbytes += l1 + l2
        if hi <= hash_border + 1:
            h1.update(r1.to_numpy(np.uint8, l1).tobytes())
            h2.update(r2.to_numpy(np.uint8, l2).tobytes())
            hashed_to = hi - 1
            hashed_bytes += l1 + l2
    out[name] = {"pairs": n, "fastq_bytes": nbytes, "gpu_s": t_gpu, "kernel_ms": {k: round(v, 1) for k, v in kernel_ms.items()}}
    assert hashed_to == hash_border and hashed_bytes > 0, (name, hashed_to, hash_border)
    out[name]["sha256_first_blocks"] = {"blocks": hash_border, "bytes": hashed_bytes, "r1:r2": h1.hexdigest() + ":" + h2.hexdigest()}
# `python tools/run_config5.py <scale> job`: the whole range as ONE rank's share -- generated once with the text kept in HBM (rsq_sim_job_generate), then written to
# files in /dev/shm by the library's writer threads (rsq_sim_job_write): seconds and GB/s of both, and that the files hold the bytes of the batched run
job = None
if "job" in sys.argv[2:]:
    import shutil
    free = shutil.disk_usage("/dev/shm").free
    t1 = time.perf_counter()
    n, b1, b2 = sim.job_generate(1, nb + 1, 24000)
    t_gen = time.perf_counter() - t1
    job = {"pairs": n, "fastq_bytes": b1 + b2, "generate_s": round(t_gen, 3), "pairs_per_s": n / t_gen, "dev_shm_free_bytes": free}
    if b1 + b2 < 0.8 * free:
        p1, p2 = "/dev/shm/rsq_c5_1.fq", "/dev/shm/rsq_c5_2.fq"
        for threads in (1, 2, 4):
            t1 = time.perf_counter()
            sim.job_write(p1, 0, p2, 0, threads)
            t_w = time.perf_counter() - t1
            job[f"write_{threads}_threads_per_file"] = {"seconds": round(t_w, 3), "gbytes_per_s": round((b1 + b2) / t_w / 1e9, 2)}
        h1, h2 = hashlib.sha256(), hashlib.sha256()
        first = out[f"batch_{batches[0]}"]
        with open(p1, "rb") as f1, open(p2, "rb") as f2:
            ok_size = os.path.getsize(p1) == b1 and os.path.getsize(p2) == b2 and b1 + b2 == first["fastq_bytes"]
        job["sizes_equal_batched_run"] = bool(ok_size and n == first["pairs"])
        os.remove(p1)
        os.remove(p2)
    sim.job_free()
# `python tools/run_config5.py <scale> shard`: the pre-pass of the same job sharded over 2, 4, 8 ranks (the ranks' simulators in this process,
# sharding.sharded_prepare_in_process): seconds per rank, and that a rank's blocks then simulate to the same text as after the whole pre-pass
sharded = None
if "shard" in sys.argv[2:]:
    from reseq_amd import sharding


    class Rank:
        def __init__(self):
            self.sim = api.Simulator(prof, ref, 0)
            if not snv_only:
                self.sim.read_methylation(bpath)
            self.seq_len = lengths
            self.seconds = 0.0
            self.by_call = {}

        def _timed(self, f, *a):
            t = time.perf_counter()
            r = f(*a)
            dt = time.perf_counter() - t
            self.seconds += dt
            self.by_call[f.__name__] = round(self.by_call.get(f.__name__, 0.0) + dt, 3)
            return r

        def ref_seq_bias(self):
            return self.sim.ref_seq_bias(len(lengths))

        def prepare_plan(self, *a):
            return self._timed(self.sim.prepare_plan, *a)

        def bias_partials(self, lo, hi):
            return self._timed(self.sim.bias_partials, lo, hi)

        def prepare_normalization(self, a, b):
            return self._timed(self.sim.prepare_normalization, a, b)

        def prepare_sys_errors(self, lo, hi, st):
            return self._timed(self.sim.prepare_sys_errors, lo, hi, st)

        def prepare_finish(self):
            return self._timed(self.sim.prepare_finish)


    def text_hash(s, lo, hi):
        _, l1, l2, _ = s.pairs_device(lo, hi, None, None)
        a, b = api.DeviceArray(0, l1 + 4096), api.DeviceArray(0, l2 + 4096)
        k, l1, l2, rc = s.pairs_device(lo, hi, a, b)
        assert rc == api.RSQ_OK
        h = hashlib.sha256(a.to_numpy(np.uint8, l1).tobytes() + b.to_numpy(np.uint8, l2).tobytes()).hexdigest()
        a.free()
        b.free()
        return k, h


    sharded = {}
    for world in (2, 4, 8):
        ranks = [Rank() for _ in range(world)]
        _, ranges, rounds = sharding.sharded_prepare_in_process(ranks, 7, 0, 30.0)
        lo, hi = ranges[world // 2]
        mid = lo + (hi - lo) // 2
        same = text_hash(ranks[world // 2].sim, mid, min(hi, mid + 200)) == text_hash(sim, mid, min(hi, mid + 200))
        lo, hi = ranges[-1]
        same = same and text_hash(ranks[-1].sim, lo, min(hi, lo + 200)) == text_hash(sim, lo, min(hi, lo + 200))      # the first blocks behind a shard border
        sharded[str(world)] = {"per_rank_s": [round(r.seconds, 3) for r in ranks], "last_rank_by_call_s": ranks[-1].by_call, "chain_exchange_rounds": rounds, "equal_to_whole_pre_pass": bool(same)}
        for r in ranks:
            r.sim.close()

print(json.dumps({"config": f"configs[4] human-sized at scale {scale}, 1 GPU" + (", substitutions only, no methylation" if snv_only else ""), "sharded_prepare": sharded, "job": job, "reference_bp": total, "sequences": len(lengths), "alleles": alleles,
                  "substitutions_requested": n_sub, "indels_requested": n_indel, "methylation_regions_requested": n_regions, "total_blocks": nb,
                  "pairs_from_coverage_30": info.total_pairs, "make_inputs_s": round(t_make, 1), "load_s": round(t_load, 2), "load_stages_s": load_stages, "prepare_s": round(t_prep, 2),
                  "runs": out, "pairs_per_s_gpu": max(r["pairs"] / r["gpu_s"] for r in out.values()),
                  # pairs and bytes of every run, and the checksum of the text up to the border all batchings share
                  "batching_invariant": len({(r["pairs"], r["fastq_bytes"], r["sha256_first_blocks"]["r1:r2"]) for r in out.values()}) == 1}))
