"""Size-independent properties of the hot path at sizes the CPU oracle cannot check record by record (GPU only):
batching invariance (a checksum of checksums over block ranges), determinism in the seed, FASTQ well-formedness, mate
correspondence, pair counts against the request.  Workload: the bench's profile P0 (2x150) on a 2 Mb reference."""
import hashlib

import numpy as np
import pytest

from reseq_amd import api, synth

pytestmark = pytest.mark.gpu

GENOME = 4_641_652                             # BASELINE.json configs[1] at full size: E. coli-sized, 10 M pairs
PAIRS = 10_000_000
LENGTHS = [2_800_000, 500, GENOME - 2_800_000 - 500]


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    d = tmp_path_factory.mktemp("props")
    ppath, fpath = d / "p0.rsqp", d / "ref.fa"
    synth.write_profile(ppath, synth.make_profile(synth.P0, seed=103741084))
    seqs = synth.make_reference(4, LENGTHS, gc=0.45)       # a scaffold shorter than the longest insert in the middle
    synth.write_fasta(fpath, seqs)
    prof = api.Profile(str(ppath))
    ref = api.Reference(str(fpath), 1)
    yield prof, ref, seqs
    prof.close()
    ref.close()


def run(prof, ref, seed, batch_blocks):
    sim = api.Simulator(prof, ref, 0)
    info = sim.prepare(seed, PAIRS)
    digests, n_total, texts = [hashlib.sha256(), hashlib.sha256()], 0, None
    for lo in range(1, info.total_blocks + 1, batch_blocks):
        fr, r1, r2 = sim.pairs(lo, min(info.total_blocks + 1, lo + batch_blocks))
        n_total += len(fr)
        digests[0].update(r1)
        digests[1].update(r2)
        if texts is None:
            texts = (fr, r1, r2)
    sim.close()
    return info, n_total, digests[0].hexdigest(), digests[1].hexdigest(), texts


def test_batching_invariance_determinism_and_counts(world):
    prof, ref, seqs = world
    info, n_a, a1, a2, first = run(prof, ref, 11, 500)
    _, n_b, b1, b2, _ = run(prof, ref, 11, 173)                # other batch boundaries, same bytes
    assert (n_a, a1, a2) == (n_b, b1, b2)
    _, n_c, c1, _, _ = run(prof, ref, 12, 500)                 # another seed, other bytes, about the same count
    assert c1 != a1 and abs(n_c - n_a) < 0.01 * n_a
    # the 500-base scaffold has no unit (Simulator.cpp:1159)
    assert info.total_blocks == 2800 + (LENGTHS[2] + 999) // 1000
    assert abs(n_a - info.total_pairs) < 0.005 * info.total_pairs          # NB counts around the requested number
    fr, r1, r2 = first
    l1, l2 = r1.split(b"\n"), r2.split(b"\n")
    assert l1[-1] == b"" and len(l1) == 4 * len(fr) + 1 and len(l2) == len(l1)
    ids1, ids2 = l1[0::4][:-1], l2[0::4][:-1]
    assert all(x.startswith(b"@ReseqRead") for x in ids1[:1000])
    assert [x.split(b" ")[0] for x in ids1] == [x.split(b" ")[0] for x in ids2]       # same pair, same coordinates in both files
    assert all(s == b"+" for s in l1[2::4]) and all(len(s) == 150 for s in l1[1::4][:-1] if s) 
    seq_bytes = np.frombuffer(b"".join(l1[1::4]), np.uint8)
    assert set(np.unique(seq_bytes)) <= set(b"ACGTN")
    qual = np.frombuffer(b"".join(l1[3::4]), np.uint8)
    assert qual.min() >= 33 + 2 and qual.max() <= 33 + 41                              # P0 qualities 2..41, offset 33
    # fragments are sorted the way the reference's loops emit them: block, start, length
    key = fr["block"].astype(np.int64) * (1 << 40) + fr["start"].astype(np.int64) * (1 << 12) + fr["len"]
    assert np.all(np.diff(key) >= 0)
    assert np.all(fr["start"] + fr["len"] < np.where(fr["seq"] == 0, LENGTHS[0], LENGTHS[2]))
