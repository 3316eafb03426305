#!/usr/bin/env python3
"""Wall time of `reseq seqToIllumina` (BASELINE.json configs[2] through the command line): N FASTA records of 150 bases with
"{id} {1|2};{fragment length};{dominant errors};{error rates}" headers in /dev/shm -> FASTQ in /dev/shm.  The first records are
checked against the oracle.  The run is bound by the host: reading and parsing ~470 bytes and writing ~340 bytes per record.
Usage: python tools/time_seq_to_illumina.py [records]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from reseq_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
L, BASE = 150, 100_000
tmp = tempfile.mkdtemp(prefix="rsq_s2i_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
ppath = os.path.join(tmp, "p0.rsqp")
arrays = synth.make_profile(synth.P0, seed=103741084)
synth.write_profile(ppath, arrays)
rec = synth.make_error_model_input(3, BASE, L, arrays, zero_frac=0.97)
rec["frag_len"] = np.clip(rec["frag_len"], 100, 999).astype(np.uint32)             # three digits: fixed-width records
# one record = ">r" 9 digits " " seg ";" 3 digits ";" dom ";" rate "\n" seq "\n"
W = 2 + 9 + 1 + 1 + 1 + 3 + 1 + L + 1 + L + 1 + L + 1
block = np.zeros((BASE, W), np.uint8)
block[:, 0:2] = np.frombuffer(b">r", np.uint8)
block[:, 11] = ord(" ")
block[:, 12] = rec["seg"] + ord("1")
block[:, 13] = ord(";")
fl = rec["frag_len"]
for k in range(3):
    block[:, 14 + k] = (fl // 10 ** (2 - k)) % 10 + ord("0")
block[:, 17] = ord(";")
block[:, 18:18 + L] = np.frombuffer(b"ACGTN", np.uint8)[rec["dom"]]
block[:, 18 + L] = ord(";")
block[:, 19 + L:19 + 2 * L] = synth.encode_sys_rate(rec["rate"])
block[:, 19 + 2 * L] = ord("\n")
block[:, 20 + 2 * L:20 + 3 * L] = np.frombuffer(b"ACGT", np.uint8)[rec["seqs"]]
block[:, 20 + 3 * L] = ord("\n")
inp, out = os.path.join(tmp, "in.fa"), os.path.join(tmp, "out.fq")
with open(inp, "wb") as f:
    for first in range(0, N, BASE):
        n = min(BASE, N - first)
        idx = np.arange(first, first + n)
        for k in range(9):
            block[:n, 2 + k] = (idx // 10 ** (8 - k)) % 10 + ord("0")
        block[:n].tofile(f)
exe = os.path.join(ROOT, "reseq_amd", "reseq")
times = []
for _ in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "seqToIllumina", "-i", inp, "-o", out, "-s", ppath, "--seed", "5", "--traceStages"] + sys.argv[2:], check=True, capture_output=True, text=True)
    times.append(time.perf_counter() - t0)
    stages = [l for l in r.stderr.splitlines() if l.startswith("stages of")]
# the first records against the oracle
import oracle_lib as O  # noqa: E402
K = 3000
oprof = O.Profile(ppath)
head = {k: v[:K] for k, v in rec.items()}
r = head["rate"].astype(np.int64)
head["rate"] = np.where(r > 86, r - r % 2, r).astype(np.uint8)
want = "".join(f"@r{i:09d} {cigar} E{nerr}\n" + "".join("ACGTN"[b] for b in seq) + "\n+\n" + qual.decode() + "\n"
               for i, (seq, qual, cigar, nerr, _t) in enumerate(O.error_model_only(oprof, 5, head, first_index=0)))
with open(out, "rb") as f:
    got = f.read(len(want)).decode()
# and a stretch in the middle (another block of the parallel parser; the record index selects the random stream, so order matters)
mid_ok = None
if N <= 10_000_000 and N > 2 * BASE:
    k0 = (N // 2 // BASE) * BASE + 777
    sel = {k: v[777:777 + 500] for k, v in rec.items()}
    r = sel["rate"].astype(np.int64)
    sel["rate"] = np.where(r > 86, r - r % 2, r).astype(np.uint8)
    want_mid = "".join(f"@r{k0 + i:09d} {cigar} E{nerr}\n" + "".join("ACGTN"[b] for b in seq) + "\n+\n" + qual.decode() + "\n"
                       for i, (seq, qual, cigar, nerr, _t) in enumerate(O.error_model_only(oprof, 5, sel, first_index=k0)))
    with open(out, "rb") as f:
        lines_seen, got_mid = 0, []
        for line in f:
            if lines_seen >= 4 * k0:
                got_mid.append(line)
                if len(got_mid) == 4 * 500:
                    break
            lines_seen += 1
    mid_ok = b"".join(got_mid).decode() == want_mid
in_bytes, out_bytes = os.path.getsize(inp), os.path.getsize(out)
print(json.dumps({"config": "configs[2] through `reseq seqToIllumina` (files in /dev/shm)", "records": N, "read_len": L, "wall_s": times, "stages": stages[-1] if stages else None, "reads_per_s_wall": N / min(times),
                  "input_bytes": in_bytes, "output_bytes": out_bytes, "first_records_equal_oracle": got == want, "checked_records": K, "records_in_the_middle_equal_oracle": mid_ok}))
for p in (inp, out, ppath):
    os.remove(p)
os.rmdir(tmp)
