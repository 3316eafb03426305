"""seqToIllumina over several ranks ("shards by input record ranges"): rsq_fasta_count_records (host code of the library) against Python on files
with '>' in every place a record start is not; sharding.record_share; and reseq_amd.simulate.run_records_rank over gloo with 2 and 3 ranks -- the launcher's
exchanges, offsets and empty shares, with a stand-in for the device pipeline that numbers the records it is given (the pipeline itself runs on the GPU:
tests/test_parity_gpu.py::test_seq_to_illumina_in_shares_equals_the_single_run)."""
import os
import pathlib
import random
import subprocess
import sys

import pytest

from reseq_amd import api, sharding

HERE = pathlib.Path(__file__).resolve().parent


def starts_of(text):
    return [p for p in range(len(text)) if text[p:p + 1] == b">" and (p == 0 or text[p - 1:p] == b"\n")]


def random_fasta(rng, n, junk=True):
    out = []
    for i in range(n):
        rid = f">r{i} x>y" if junk and i % 3 == 0 else f">r{i}"
        out.append(rid + " 1;100;" + "N" * 5 + ";" + ">" * 5)           # '>' is a legal rate character (29 percent)
        seq = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 60)))
        out += [seq[k:k + 17] for k in range(0, len(seq), 17)] if i % 4 == 0 else [seq]
    return ("\n".join(out) + "\n").encode()


def test_count_records_against_python(tmp_path):
    rng = random.Random(5)
    for case in range(6):
        text = random_fasta(rng, rng.randint(1, 400))
        if case == 4:
            text = b"\n\n" + text
        if case == 5:
            text = text[:-1]                                                # no line end at the end of the file
        path = tmp_path / f"c{case}.fa"
        path.write_bytes(text)
        starts = starts_of(text)
        assert api.count_fasta_records(path) == (len(starts), starts[0])
        for _ in range(40):
            a = rng.randrange(len(text) + 1)
            b = rng.randrange(a, len(text) + 1)
            inside = [p for p in starts if a <= p < b]
            for threads in (1, 3):
                assert api.count_fasta_records(path, a, b, threads) == (len(inside), inside[0] if inside else b), (case, a, b)
    big = tmp_path / "big.fa"                                               # several pieces of 8 MB for the threads, starts on both sides of their borders
    text = random_fasta(rng, 150000, junk=False)
    big.write_bytes(text * 3)
    assert len(text) * 3 > 17 << 20
    assert api.count_fasta_records(big, 0, 0, 4) == (450000, 0)
    cut = (8 << 20) + 1
    want = [p for p in starts_of(text * 3) if cut <= p]
    assert api.count_fasta_records(big, cut, 0, 4) == (len(want), want[0])
    with pytest.raises(Exception, match="not a plain file"):
        api.count_fasta_records(tmp_path)


def test_record_shares_partition_the_file():
    rng = random.Random(7)
    for _ in range(200):
        text = random_fasta(rng, rng.randint(1, 30))
        starts, size = starts_of(text), len(text)
        world = rng.randint(1, 9)
        counts = []
        for r in range(world):
            lo, hi = sharding.record_stretch(size, r, world)
            inside = [p for p in starts if lo <= p < hi]
            counts.append((len(inside), inside[0] if inside else hi))
        at, records = 0, 0
        for r in range(world):
            begin, end, first = sharding.record_share(counts, size, r)
            if end > begin:
                assert begin == at and first == records and (begin in starts or begin == 0) and (end in starts or end == size)
                at = end
                records += sum(1 for p in starts if begin <= p < end)
            else:
                assert counts[r][0] == 0 or r == 0
        assert at == size and records == len(starts)


WORKER = r"""
import os, sys, pathlib
sys.path.insert(0, os.environ["RSQ_ROOT"])
import torch.distributed as dist
from reseq_amd import api, simulate

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["RSQ_PORT"], rank=rank, world_size=world)


class Numbering:
    # stands in for the simulator: the "FASTQ text" of a record is its index in the input and its id line; kept, then written at an offset like the job's text
    def error_model_file(self, input_path, output_path, from_=0, to=0, first_record=0, keep_text=False):
        assert keep_text and output_path is None
        if os.environ.get("RSQ_FAIL") == str(rank):
            raise RuntimeError("Template segment is 3 not 1 or 2: r 3;1;N;!")
        text = open(input_path, "rb").read()[from_:to]
        assert text[:1] == b">" or from_ == 0
        lines = [l for l in text.split(b"\n") if l[:1] == b">"]
        self.text = b"".join(b"%d %s\n" % (first_record + i, l) for i, l in enumerate(lines))
        return len(lines), len(self.text)

    def job_write(self, path, offset, path2, offset2):
        assert path2 is None
        with open(path, "r+b") as f:
            f.seek(offset)
            f.write(self.text)

    def job_free(self):
        self.text = None


work = pathlib.Path(os.environ["RSQ_WORK"])
records, _ = simulate.run_records_rank(Numbering(), dist if world > 1 else None, rank, world, str(work / "in.fa"), str(work / os.environ["RSQ_OUT"]), "cpu",
                                       bool(os.environ.get("RSQ_SPLIT")), count=api.count_fasta_records)
if rank == 0:
    print("RECORDS", records)
if world > 1:
    dist.destroy_process_group()
"""


def launch(workdir, world, out, **env):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, RSQ_ROOT=str(HERE.parent), RSQ_WORK=str(workdir), RSQ_PORT=str(port), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", RSQ_OUT=out, **env)
    procs = [subprocess.Popen([sys.executable, "-c", WORKER], env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    return procs, [p.communicate(timeout=300) for p in procs]


@pytest.mark.timeout(600)
def test_ranks_over_gloo_write_the_single_run(workdir):
    rng = random.Random(3)
    text = random_fasta(rng, 500)
    (workdir / "in.fa").write_bytes(text)
    lines = [l for l in text.split(b"\n") if l[:1] == b">"]
    want = b"".join(b"%d %s\n" % (i, l) for i, l in enumerate(lines))
    for world in (1, 2, 3):
        procs, outs = launch(workdir, world, f"out{world}.fq")
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, se.decode()[-3000:]
        assert b"RECORDS 500" in outs[0][0]
        assert (workdir / f"out{world}.fq").read_bytes() == want, world
    procs, outs = launch(workdir, 3, "split.fq", RSQ_SPLIT="1")
    assert all(p.returncode == 0 for p in procs), outs
    assert b"".join((workdir / f"split.fq.part{k}of3").read_bytes() for k in (1, 2, 3)) == want
    # two records only: the third rank's stretch holds no record start, and the second's share ends at the file's end
    (workdir / "in.fa").write_bytes(b">a 1;100;N;!\nA\n>b 2;100;" + b"N" * 300 + b";" + b"!" * 300 + b"\n" + b"A" * 300 + b"\n")
    procs, outs = launch(workdir, 3, "few.fq")
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-2000:] for o in outs]
    assert (workdir / "few.fq").read_bytes().startswith(b"0 >a 1;100;N;!\n1 >b 2;100;")
    # a rank that fails takes the others with it instead of leaving them in a collective
    (workdir / "in.fa").write_bytes(text)
    procs, outs = launch(workdir, 2, "failed.fq", RSQ_FAIL="1")
    assert procs[1].returncode != 0 and b"Template segment is 3" in outs[1][1]
    assert procs[0].returncode != 0 and b"another rank failed while simulating its records" in outs[0][1]
