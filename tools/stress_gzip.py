#!/usr/bin/env python3
"""Randomised check of the device's gzip (rsq_sim_gzip_device) against the host walk (tests/hostemu, byte-equal) and zlib's inflate: texts of many kinds -- FASTQ-like with
long and short reads and few or many quality values, random bytes, runs, lines that all begin with '@', no newline at all -- at lengths around the borders of segments
(32), rounds (8192) and pieces (65280).  Usage: python tools/stress_gzip.py [trials]"""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from reseq_amd import api, synth  # noqa: E402
from test_device_gzip import emu_gzip, members_of  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tmp = "/tmp/rsq_stress_gzip.rsqp"
synth.write_profile(tmp, synth.make_profile(synth.TINY, seed=5))
prof = api.Profile(tmp)
sim = api.Simulator(prof, None, 0)


def fastq(rng, n, read_len, n_quals, id_digits):
    recs, pos = [], 1000
    for i in range(n):
        pos += int(rng.integers(0, 7))
        seq = bytes(b"ACGT"[c] for c in rng.integers(0, 4, read_len))
        q = rng.integers(0, n_quals, read_len)
        if rng.random() < 0.5:
            q[rng.random(read_len) < 0.6] = n_quals - 1
        qual = bytes(int(x) + 35 for x in q)
        recs.append(b"@Read%0*d:%d:chr:%d %dM E%d\n%s\n+\n%s\n" % (id_digits, i, pos, pos + 350, read_len, int(rng.integers(0, 3)), seq, qual))
    return b"".join(recs)


bad = 0
for t in range(trials):
    rng = np.random.default_rng(9000 + t)
    kind = t % 8
    target = int(rng.choice([1, 3, 31, 32, 33, 8191, 8192, 8193, 65279, 65280, 65281, 130560, 200000, 70000 + int(rng.integers(0, 300000))]))
    if kind == 0:
        text = fastq(rng, target // 330 + 1, 150, 40, 6)
    elif kind == 1:
        text = fastq(rng, target // 80 + 1, 30, 5, 3)
    elif kind == 2:
        text = fastq(rng, target // 330 + 1, 150, 4, 9)
    elif kind == 3:
        text = rng.integers(0, 256, target, dtype=np.uint8).tobytes()
    elif kind == 4:
        text = (b"@" + bytes(rng.integers(65, 70, 40, dtype=np.uint8)) + b"\n") * (target // 42 + 1)
    elif kind == 5:
        text = bytes(rng.integers(65, 68, target, dtype=np.uint8))                       # no newline at all
    elif kind == 6:
        text = b"".join(bytes([int(rng.integers(33, 40))]) * int(rng.integers(1, 300)) for _ in range(target // 100 + 1))   # runs
    else:
        text = b"\n" * (target // 2) + b"@\n+\n" * (target // 8 + 1)
    text = text[:max(1, target)]
    device = sim.gzip(text)
    host = emu_gzip(text)
    ok = device == host and b"".join(m[1] for m in members_of(device)) == text and zlib.decompress(device[:len(members_of(device)[0][0])], 31) == members_of(device)[0][1]
    if not ok:
        bad += 1
    print(f"trial {t}: {'ok' if ok else 'MISMATCH'} kind {kind} bytes {len(text)} -> {len(device)}", flush=True)
print("mismatching trials:", bad)
sim.close()
prof.close()
sys.exit(1 if bad else 0)
