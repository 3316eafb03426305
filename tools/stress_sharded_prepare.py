"""Randomised check of the sharded pre-pass on the device: 2 to 7 in-process ranks, with and without variants crowding the shard borders;
thresholds and every rank's FASTQ text must equal the whole pre-pass's (tests/parity_cases.py case_sharded_prepare)."""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as P
from backends import GpuBackend
ok = bad = skipped = 0
for seed in range(40, 64):
    for world in (2, 3, 5, 7):
        with tempfile.TemporaryDirectory() as d:
            try:
                P.case_sharded_prepare(GpuBackend, pathlib.Path(d), world=world, variants=bool(seed & 1), seed=seed)
                ok += 1
            except Exception as e:
                if "walk left the sequence" in str(e):
                    skipped += 1
                else:
                    bad += 1
                    print("FAIL", seed, world, type(e).__name__, str(e)[:200])
print("ok", ok, "skipped (walk-off sets)", skipped, "bad", bad)
