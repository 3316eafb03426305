#!/usr/bin/env python3
"""Wall time of `reseq seqToIllumina` (BASELINE.json configs[2] through the command line): N FASTA records of 150 bases with
"{id} {1|2};{fragment length};{dominant errors};{error rates}" headers in /dev/shm -> FASTQ in /dev/shm.  The first records are
checked against the oracle.  The text is parsed on the device; the host reads ~470 bytes and writes ~325 bytes per record.
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
OUT = os.environ.get("RSQ_S2I_OUT")            # e.g. /dev/null: the pipeline without the file system on its output side
L, BASE = 150, 100_000
tmp = tempfile.mkdtemp(prefix="rsq_s2i_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
ppath = os.path.join(tmp, "p0.rsqp")
arrays = synth.make_profile(synth.P0, seed=103741084)
synth.write_profile(ppath, arrays)
rec = synth.make_error_model_input(3, BASE, L, arrays, zero_frac=0.97)
rec["frag_len"] = np.clip(rec["frag_len"], 100, 999).astype(np.uint32)             # three digits: fixed-width records
block = synth.fixed_width_fasta(rec)
inp, out = os.path.join(tmp, "in.fa"), OUT or os.path.join(tmp, "out.fq")
with open(inp, "wb") as f:
    for first in range(0, N, BASE):
        n = min(BASE, N - first)
        synth.number_rows(block[:n], first)
        block[:n].tofile(f)
exe = os.path.join(ROOT, "reseq_amd", "reseq")
times = []
for _ in range(2):
    if os.path.isfile(out):
        os.remove(out)                      # not part of the run: truncating 6 GB of tmpfs pages takes a third of a second
    t0 = time.perf_counter()
    r = subprocess.run([exe, "seqToIllumina", "-i", inp, "-o", out, "-s", ppath, "--seed", "5", "--traceStages"] + sys.argv[2:], check=True, capture_output=True, text=True)
    times.append(time.perf_counter() - t0)
    stages = [l for l in r.stderr.splitlines() if l.startswith("stages")]
# the first records against the oracle
import oracle_lib as O  # noqa: E402
K = 3000
oprof = O.Profile(ppath)
head = {k: v[:K] for k, v in rec.items()}
r = head["rate"].astype(np.int64)
head["rate"] = np.where(r > 86, r - r % 2, r).astype(np.uint8)
want = "".join(f"@r{i:09d} {cigar} E{nerr}\n" + "".join("ACGTN"[b] for b in seq) + "\n+\n" + qual.decode() + "\n"
               for i, (seq, qual, cigar, nerr, _t) in enumerate(O.error_model_only(oprof, 5, head, first_index=0)))
if OUT and OUT.endswith(".gz"):                       # the compressed output: size, and what one thread of zlib makes of the same text
    import gzip
    import zlib
    with gzip.open(OUT, "rb") as f:
        sample = f.read(64 << 20)
    t0 = time.perf_counter()
    zlib.compress(sample, 6)
    one_thread = len(sample) / (time.perf_counter() - t0)
    ok = sample.startswith(b"@r000000000 ")
    print(json.dumps({"config": "configs[2] through `reseq seqToIllumina`, output to " + OUT, "records": N, "wall_s": times, "stages": stages[-1] if stages else None, "flags": sys.argv[2:],
                      "compressed_bytes": os.path.getsize(OUT), "text_bytes_per_s_wall": N * 324 / min(times), "one_thread_of_zlib_level_6_bytes_per_s": one_thread, "first_record_as_expected": ok}))
    os.remove(inp), os.remove(ppath), os.remove(OUT), os.rmdir(tmp)
    sys.exit(0)
if OUT:
    print(json.dumps({"config": "configs[2] through `reseq seqToIllumina`, output to " + OUT, "records": N, "wall_s": times, "stages": stages[-1] if stages else None, "flags": sys.argv[2:]}))
    os.remove(inp), os.remove(ppath), os.rmdir(tmp)
    sys.exit(0)
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
                  "flags": sys.argv[2:], "input_bytes": in_bytes, "output_bytes": out_bytes, "first_records_equal_oracle": got == want, "checked_records": K, "records_in_the_middle_equal_oracle": mid_ok}))
for p in (inp, out, ppath):
    os.remove(p)
os.rmdir(tmp)
