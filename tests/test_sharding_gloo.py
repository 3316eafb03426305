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
from emu_ranks import EmuRankBackend          # the host emulation behind the interface simulate.run_rank drives (the GPU run uses simulate.GpuBackend)
from reseq_amd import simulate, synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["RSQ_PORT"], rank=rank, world_size=world)
work = pathlib.Path(os.environ["RSQ_WORK"])
(work / f"sim{rank}").mkdir(parents=True, exist_ok=True)
ppath, fpath, seqs = P.make_inputs(work / f"sim{rank}", "simjob", synth.TINY, [5000, 80, 3210])
tag = os.environ["RSQ_TAG"]
if os.environ.get("RSQ_SHARED_LOAD"):          # the two ranks as the ranks of one host: rank 0 loads and exports, rank 1 imports (its FASTA path does not even exist)
    make = lambda packed_from: EmuRankBackend(ppath, fpath if rank == 0 else str(work / "no_such.fa"), packed_from=packed_from)
    backend = simulate.load_once_per_host(make, dist if world > 1 else None, "cpu", rank, world, 0, shm_dir=str(work))
    assert not list(work.glob("rsq_packed_reference_*")), "the exported file is gone once every rank has it"
else:
    backend = EmuRankBackend(ppath, fpath)
pairs, _ = simulate.run_rank(backend, dist if world > 1 else None, rank, world, str(work / f"{tag}_1.fq"), str(work / f"{tag}_2.fq"), 7, 30000, 0.0, 1, "Job", 3,
                             split_output=bool(os.environ.get("RSQ_SPLIT")))
if rank == 0:
    print("PAIRS", pairs)
if world > 1:
    dist.destroy_process_group()
"""


@pytest.mark.timeout(900)
def test_a_failing_rank_leaves_nobody_waiting(workdir):
    """simulate.run_rank / load_once_per_host / sharded_prepare over gloo with two ranks: a rank that fails in any phase raises its own error, the other one says that
    another rank failed (simulate._agree) -- nobody stays in a collective.  (What the ranks write when nothing fails: tests/test_multi_gpu.py, through the launcher.)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, RSQ_TESTS=str(HERE), RSQ_ROOT=str(HERE.parent), RSQ_WORK=str(workdir), RSQ_PORT=str(port), MASTER_ADDR="127.0.0.1")

    def two_ranks(tag, **switches):
        procs = [subprocess.Popen([sys.executable, "-c", SIMULATE_WORKER], env=dict(base, RANK=str(r), WORLD_SIZE="2", RSQ_TAG=tag, **switches), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                 for r in range(2)]
        errs = [p.communicate(timeout=300)[1].decode() for p in procs]
        return procs, errs

    procs, errs = two_ranks("fine")
    assert all(p.returncode == 0 for p in procs), errs
    for split in ("", "1"):
        procs, errs = two_ranks("fail" + split, RSQ_SPLIT=split, RSQ_FAIL_WRITE="1")
        assert all(p.returncode != 0 for p in procs)
        assert "another rank failed while writing" in errs[0] and "no space left on the device" in errs[1], errs
    # ... also in the pre-pass, whose phases stand in front of collectives (the bias sums' all-reduce, the chain states' all-gather)
    for switch, failing, said, own in (("RSQ_FAIL_BIAS", 0, "another rank failed while summing the coverage bias", "no memory for the bias sums"),
                                       ("RSQ_FAIL_CHAINS", 1, "another rank failed while running the systematic-error chains", "the chains failed")):
        procs, errs = two_ranks("failpre", **{switch: str(failing)})
        assert all(p.returncode != 0 for p in procs)
        assert said in errs[1 - failing] and own in errs[failing], errs
    # ... and in the shared load: the loader's failure reaches the rank that waits for its file, an importer's failure the loader
    for switch, failing, said, own in (("RSQ_FAIL_LOAD", 0, "another rank failed while loading and packing the reference", "reference file not found"),
                                       ("RSQ_FAIL_IMPORT", 1, "another rank failed while taking the packed reference", "cannot map the packed reference")):
        procs, errs = two_ranks("failload", RSQ_SHARED_LOAD="1", **{switch: str(failing)})
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
