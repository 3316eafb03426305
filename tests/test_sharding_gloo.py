"""The N > 1 path on CPU: two processes over gloo, each simulating its block range with the host emulation of the
kernels; the concatenated shards must equal the single-rank output byte for byte, and the job totals must add up."""
import os
import pathlib
import subprocess
import sys

import pytest

from reseq_amd import sharding

HERE = pathlib.Path(__file__).resolve().parent


def test_partition_covers_every_block_once():
    for total, world in ((1, 1), (7, 2), (8, 3), (5, 8), (1000, 8)):
        parts = sharding.partition_blocks(total, world)
        assert len(parts) == world and parts[0][0] == 1 and parts[-1][1] == total + 1
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert all(lo <= hi for lo, hi in parts)
    # balanced by weight: the heavy first block gets a rank of its own
    assert sharding.partition_blocks(4, 2, [10, 1, 1, 1])[0] == (1, 2)
    assert sharding.batches(3, 10, 4) == [(3, 7), (7, 10)]


SIMULATE_WORKER = r"""
import os, sys, pathlib
sys.path.insert(0, os.environ["RSQ_TESTS"]); sys.path.insert(0, os.environ["RSQ_ROOT"])
import torch.distributed as dist
import parity_cases as P
from backends import EmuBackend
from reseq_amd import simulate, synth

class Emu:                      # the host emulation behind the interface simulate.run_rank drives (the GPU run uses simulate.GpuBackend)
    def __init__(self, ppath, fpath, seqs, vcf=None, packed_from=None):
        if packed_from:             # the reference another rank of the "host" packed (simulate.load_once_per_host)
            if os.environ.get("RSQ_FAIL_IMPORT") == os.environ["RANK"]:
                raise IOError("cannot map the packed reference (the test's)")
            self.b = EmuBackend(ppath, None, 0)
            self.b.import_reference(packed_from)
        else:
            if os.environ.get("RSQ_FAIL_LOAD") == os.environ["RANK"]:
                raise IOError("reference file not found (the test's)")
            self.b = EmuBackend(ppath, fpath, 0, None, vcf_path=vcf) if vcf else EmuBackend(ppath, fpath, 0)
        self.seq_len = [len(c) for _, c in seqs]
        self.n_seqs = len(self.seq_len)
        self.can_shard_prepare = True
    def export_reference(self, path):
        self.b.export_reference(path)
    def close(self):
        self.b.close()
    def prepare(self, *a):
        return self.b.prepare(*a)
    def prepare_plan(self, *a):
        return self.b.prepare_plan(*a)
    def bias_partials(self, lo, hi):
        if os.environ.get("RSQ_FAIL_BIAS") == os.environ["RANK"]:
            raise MemoryError("no memory for the bias sums (the test's)")
        return self.b.bias_partials(lo, hi)
    def prepare_normalization(self, sums, maxes):
        self.b.prepare_normalization(sums, maxes)
    def prepare_sys_errors(self, lo, hi, in_state):
        if os.environ.get("RSQ_FAIL_CHAINS") == os.environ["RANK"]:
            raise RuntimeError("the chains failed (the test's)")
        return self.b.prepare_sys_errors(lo, hi, in_state)
    def prepare_finish(self):
        return self.b.prepare_finish()
    def ref_seq_bias(self):
        return self.b.ref_seq_bias(self.n_seqs)
    def job_generate(self, lo, hi, batch_blocks):          # what rsq_sim_job_generate / rsq_sim_job_write do, for the host emulation: the text kept, then put in place
        from reseq_amd import sharding
        self.text, n = [bytearray(), bytearray()], 0
        for a, b in sharding.batches(lo, hi, batch_blocks or 2000):
            fr, t1, t2 = self.b.pairs(a, b)
            n += len(fr)
            self.text[0] += t1
            self.text[1] += t2
        return n, len(self.text[0]), len(self.text[1])
    def job_compress(self):                                # rsq_sim_job_compress: gzip members of 1 MB of text
        import gzip
        self.text = [bytearray(b"".join(gzip.compress(bytes(t[k:k + (1 << 20)]), 6) for k in range(0, len(t), 1 << 20))) for t in self.text]
        return len(self.text[0]), len(self.text[1])
    def job_write(self, path1, offset1, path2, offset2):
        if os.environ.get("RSQ_FAIL_WRITE") == os.environ["RANK"]:
            raise IOError("no space left on the device (the test's)")
        for path, offset, text in ((path1, offset1, self.text[0]), (path2, offset2, self.text[1])):
            fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
            os.pwrite(fd, bytes(text), offset)
            os.close(fd)
    def adapter_only_pairs(self, first, n):
        return self.b.adapter_only_pairs(first, n)
    def job_slice(self, file, at, n, size):                # --gatherOutput: a fixed-size slice of the kept text as a tensor, a received one to its place
        import torch
        t = torch.zeros(size, dtype=torch.uint8)
        if n:
            t[:n] = torch.frombuffer(bytearray(self.text[file][at:at + n]), dtype=torch.uint8)
        return t
    def write_slice(self, tensor, n, path, offset):
        fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
        os.pwrite(fd, tensor[:n].numpy().tobytes(), offset)
        os.close(fd)
    def job_free(self):
        self.text = None

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["RSQ_PORT"], rank=rank, world_size=world)
work = pathlib.Path(os.environ["RSQ_WORK"])
(work / f"sim{rank}").mkdir(parents=True, exist_ok=True)
ppath, fpath, seqs = P.make_inputs(work / f"sim{rank}", "simjob", synth.TINY, [5000, 80, 3210])
tag = os.environ["RSQ_TAG"]
vcf = None
if os.environ.get("RSQ_VARIANTS"):           # substitutions, insertions, deletions on two alleles: extra start slots cross the shard borders too
    import numpy as np
    vcf = work / f"sim{rank}" / "simjob.vcf"
    P.write_vcf(vcf, seqs, P._mixed_variant_set(seqs, np.random.default_rng(5), 30, [999, 1000, 1999, 2000, 2999, 3000]))
if os.environ.get("RSQ_SHARED_LOAD"):          # the two ranks as the ranks of one host: rank 0 loads and exports, rank 1 imports (its FASTA path does not even exist)
    make = lambda packed_from: Emu(ppath, fpath if rank == 0 else str(work / "no_such.fa"), seqs, vcf if rank == 0 else None, packed_from=packed_from)
    backend = simulate.load_once_per_host(make, dist if world > 1 else None, "cpu", rank, world, 0, shm_dir=str(work))
    assert not list(work.glob("rsq_packed_reference_*")), "the exported file is gone once every rank has it"
else:
    backend = Emu(ppath, fpath, seqs, vcf)
pairs, _ = simulate.run_rank(backend, dist if world > 1 else None, rank, world, str(work / f"{tag}_1.fq"), str(work / f"{tag}_2.fq"), 7, 30000, 0.0, 1, "Job", 3,
                             split_output=bool(os.environ.get("RSQ_SPLIT")), gather_output=bool(os.environ.get("RSQ_GATHER")), gather_slice_bytes=200_000,
                             compress=tag.endswith("gz"))
if rank == 0:
    print("PAIRS", pairs)
if world > 1:
    dist.destroy_process_group()
"""


@pytest.mark.timeout(900)
@pytest.mark.parametrize("variants", ["", "1"])
def test_simulate_module_two_ranks_equal_one_rank(workdir, variants):
    """reseq_amd.simulate.run_rank over gloo with two ranks writes the same two FASTQ files (adapter-only pairs included) as one rank,
    without and with variants"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, RSQ_TESTS=str(HERE), RSQ_ROOT=str(HERE.parent), RSQ_WORK=str(workdir), RSQ_PORT=str(port), MASTER_ADDR="127.0.0.1", RSQ_VARIANTS=variants)
    one = subprocess.run([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK="0", WORLD_SIZE="1", RSQ_TAG="one"), capture_output=True, timeout=800)
    assert one.returncode == 0, one.stderr.decode()[-3000:]
    procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG="two"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    outs = [p.communicate(timeout=800) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-3000:]
    pairs_line = lambda out: [l for l in out.split(b"\n") if l.startswith(b"PAIRS")]          # gloo prints connection notes on stdout
    assert pairs_line(one.stdout) == pairs_line(outs[0][0]) and len(pairs_line(one.stdout)) == 1
    # --splitOutput: one pair of files per rank, whose concatenation in rank order is the single output
    procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG="split", RSQ_SPLIT="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    for p in procs:
        so, se = p.communicate(timeout=800)
        assert p.returncode == 0, se.decode()[-3000:]
    for k in (1, 2):
        assert (workdir / f"split_{k}.fq.part1of2").read_bytes() + (workdir / f"split_{k}.fq.part2of2").read_bytes() == (workdir / f"one_{k}.fq").read_bytes()
    for k in (1, 2):
        a, b = (workdir / f"one_{k}.fq").read_bytes(), (workdir / f"two_{k}.fq").read_bytes()
        assert a == b and a.count(b"\n") % 4 == 0 and b":0:Adapter:0:" in a
    assert not list(workdir.glob("*.rank*"))
    # --gatherOutput: the ranks' text gathered on rank 0 in slices of 200 kB (several rounds, the ranks' last slices of different lengths) and written by rank 0 alone
    procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG="gathered", RSQ_GATHER="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    for p in procs:
        so, se = p.communicate(timeout=800)
        assert p.returncode == 0, se.decode()[-3000:]
    for k in (1, 2):
        assert (workdir / f"gathered_{k}.fq").read_bytes() == (workdir / f"one_{k}.fq").read_bytes()
    # compressed outputs: every rank's share as gzip members at its offset of the file (the sizes exchanged are the compressed ones), the adapter-only pairs as a
    # member behind them; the decompressed files are the single run's -- also rank by rank with --splitOutput
    import gzip
    for tag, extra in (("twogz", {}), ("splitgz", {"RSQ_SPLIT": "1"})):
        procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG=tag, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                 for r in range(2)]
        for p in procs:
            so, se = p.communicate(timeout=800)
            assert p.returncode == 0, se.decode()[-3000:]
    for k in (1, 2):
        assert gzip.decompress((workdir / f"twogz_{k}.fq").read_bytes()) == (workdir / f"one_{k}.fq").read_bytes()
        parts = [(workdir / f"splitgz_{k}.fq.part{r}of2").read_bytes() for r in (1, 2)]
        assert gzip.decompress(parts[0]) + gzip.decompress(parts[1]) == (workdir / f"one_{k}.fq").read_bytes() == gzip.decompress(parts[0] + parts[1])
    # one load per host: rank 0 reads and packs, rank 1 takes the packed reference (variants included) from the shared directory -- the same two files
    procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG="shared", RSQ_SHARED_LOAD="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    for p in procs:
        so, se = p.communicate(timeout=800)
        assert p.returncode == 0, se.decode()[-3000:]
    for k in (1, 2):
        assert (workdir / f"shared_{k}.fq").read_bytes() == (workdir / f"one_{k}.fq").read_bytes()
    # a rank that fails leaves nobody waiting: it raises its own error, the other one says that another rank failed (simulate._agree)
    if not variants:
        for split in ("", "1"):
            procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG="fail" + split, RSQ_SPLIT=split, RSQ_FAIL_WRITE="1"),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
            errs = [p.communicate(timeout=300)[1].decode() for p in procs]
            assert all(p.returncode != 0 for p in procs)
            assert "another rank failed while writing" in errs[0] and "no space left on the device" in errs[1], errs
        # ... also in the pre-pass, whose phases stand in front of collectives (the bias sums' all-reduce, the chain states' all-gather)
        for switch, failing, said, own in (("RSQ_FAIL_BIAS", 0, "another rank failed while summing the coverage bias", "no memory for the bias sums"),
                                           ("RSQ_FAIL_CHAINS", 1, "another rank failed while running the systematic-error chains", "the chains failed")):
            procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG="failpre", **{switch: str(failing)}),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
            errs = [p.communicate(timeout=300)[1].decode() for p in procs]
            assert all(p.returncode != 0 for p in procs)
            assert said in errs[1 - failing] and own in errs[failing], errs
        # ... and in the shared load: the loader's failure reaches the rank that waits for its file, an importer's failure the loader
        for switch, failing, said, own in (("RSQ_FAIL_LOAD", 0, "another rank failed while loading and packing the reference", "reference file not found"),
                                           ("RSQ_FAIL_IMPORT", 1, "another rank failed while taking the packed reference", "cannot map the packed reference")):
            procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG="failload", RSQ_SHARED_LOAD="1", **{switch: str(failing)}),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
            errs = [p.communicate(timeout=300)[1].decode() for p in procs]
            assert all(p.returncode != 0 for p in procs)
            assert said in errs[1 - failing] and own in errs[failing], errs
            assert not list(workdir.glob("rsq_packed_reference_*"))


WORKER = r"""
import os, sys, pathlib, time
sys.path.insert(0, os.environ["RSQ_TESTS"]); sys.path.insert(0, os.environ["RSQ_ROOT"])
import torch, torch.distributed as dist
import parity_cases as P
from backends import EmuBackend
from reseq_amd import sharding, synth

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["RSQ_PORT"], rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
work = pathlib.Path(os.environ["RSQ_WORK"]) / f"rank{rank}"
work.mkdir(parents=True, exist_ok=True)
p = P.Pair(EmuBackend, work, "shard", synth.TINY, [5000, 80, 3210], seed=7, num_pairs=3000)
p.align_normalization()
tb = p.info["total_blocks"]
def pairs_fn(lo, hi):
    fr, a, b = p.b.pairs(lo, hi)
    return len(fr), a, b
t0 = time.perf_counter()
mine = sharding.partition_blocks(tb, world)[rank]
n, r1, r2 = sharding.simulate_shard(pairs_fn, mine, batch_blocks=2)
tot_pairs, tot_bytes, max_t = sharding.job_totals(dist, "cpu", n, len(r1) + len(r2), time.perf_counter() - t0)
shards = [None] * world
dist.all_gather_object(shards, (n, r1, r2))
if rank == 0:
    n_all, w1, w2 = sharding.simulate_shard(pairs_fn, (1, tb + 1), batch_blocks=tb)
    assert sum(s[0] for s in shards) == n_all == int(tot_pairs), (n_all, tot_pairs)
    assert b"".join(s[1] for s in shards) == w1
    assert b"".join(s[2] for s in shards) == w2
    assert int(tot_bytes) == len(w1) + len(w2) and max_t > 0
    assert all(s[0] > 0 for s in shards), "both ranks must have work in this case"
    o1, o2 = p.osim.create_reads(p.osim.sieve(1, tb + 1))          # and the whole equals the oracle
    assert o1 == w1 and o2 == w2
    print("SHARDING_OK", n_all)
p.close()
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.timeout(600)
def test_two_ranks_over_gloo_reproduce_the_single_rank_output(workdir):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RSQ_TESTS=str(HERE), RSQ_ROOT=str(HERE.parent), RSQ_WORK=str(workdir), RSQ_PORT=str(port), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=550) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-3000:]
    assert b"SHARDING_OK" in outs[0][0]


PREPASS_WORKER = r"""
import os, sys, pathlib
sys.path.insert(0, os.environ["RSQ_TESTS"]); sys.path.insert(0, os.environ["RSQ_ROOT"])
import numpy as np
import torch.distributed as dist
import parity_cases as P
from backends import EmuBackend
from reseq_amd import sharding, synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["RSQ_PORT"], rank=rank, world_size=world)
work = pathlib.Path(os.environ["RSQ_WORK"]) / f"pre{rank}"
work.mkdir(parents=True, exist_ok=True)
lengths = [9400, 80, 3210, 1000]                 # one sequence spans several ranks, one is too short for blocks, one is a single block
ppath, fpath, seqs = P.make_inputs(work, "prepass", synth.TINY, lengths)

class B:                                          # the interface sharding.sharded_prepare drives
    def __init__(self):
        self.b = EmuBackend(ppath, fpath, 0)
        self.seq_len = lengths
        self.states = []
    def ref_seq_bias(self):
        return self.b.ref_seq_bias(len(lengths))
    def prepare_sys_errors(self, lo, hi, in_state):
        out = self.b.prepare_sys_errors(lo, hi, in_state)
        self.states.append((list(in_state), out))
        return out
    def __getattr__(self, name):
        return getattr(self.b, name)

b = B()
info, (lo, hi), rounds = sharding.sharded_prepare(b, dist, "cpu", rank, world, 23, 20000, 0.0, 1, "Pre")
whole = EmuBackend(ppath, fpath, 0)
winfo = whole.prepare(23, 20000, 0.0, 1, "Pre")
assert info["total_pairs"] == winfo["total_pairs"] and info["bias_normalization"] == winfo["bias_normalization"]
assert np.array_equal(b.b.thresholds(), whole.thresholds()) and np.array_equal(b.b.norm_by_len(), whole.norm_by_len())
# the tracks over the positions the rank's reads can touch
first_block, covered = 1, 0
for seq, L in enumerate(lengths):
    if L < info["insert_to"]:
        continue
    nb = (L + 999) // 1000
    blo, bhi = max(first_block, lo), min(first_block + nb, hi)
    if blo < bhi:
        p_lo, t_hi = (blo - first_block) * 1000, min(L, min(L, (bhi - first_block) * 1000) + info["insert_to"])
        for strand in (0, 1):
            mine, ref = b.b.sys_errors(strand, seq, L), whole.sys_errors(strand, seq, L)
            sl = slice(p_lo, t_hi) if strand == 0 else slice(L - t_hi, L - p_lo)      # the reverse track is indexed L-1-position
            assert np.array_equal(mine[0][sl], ref[0][sl]) and np.array_equal(mine[1][sl], ref[1][sl]), (rank, seq, strand)
            covered += t_hi - p_lo
    first_block += nb
entered = [s[0] for s in b.states]
summary = [None] * world
dist.all_gather_object(summary, dict(rank=rank, range=(lo, hi), rounds=rounds, covered=covered, nonzero_in=any(any(e) for e in entered), calls=len(b.states)))
if rank == 0:
    assert all(s["covered"] > 0 for s in summary) and any(s["nonzero_in"] for s in summary), summary
    assert max(s["rounds"] for s in summary) >= 2, summary
    print("PREPASS_OK", summary)
whole.close(); b.b.close()
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.timeout(900)
def test_sharded_pre_passes_equal_the_whole_pre_pass(workdir):
    """four ranks over gloo: every rank computes its share of the bias sums and of the systematic-error chains (sharding.sharded_prepare);
    thresholds and normalisation equal a whole-genome pre-pass exactly, the tracks equal it over every position the rank's reads can touch,
    and chain states did cross shard borders"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RSQ_TESTS=str(HERE), RSQ_ROOT=str(HERE.parent), RSQ_WORK=str(workdir), RSQ_PORT=str(port), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, "-c", PREPASS_WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(4)]
    outs = [p.communicate(timeout=800) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-3000:]
    assert b"PREPASS_OK" in outs[0][0], outs[0][0][-2000:]
