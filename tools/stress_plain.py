#!/usr/bin/env python3
"""Randomised parity stress without variants: random references (G/C content, several sequences, some without blocks), synthetic profiles
from different seeds, random pair counts and seeds, profile edits, optional methylation; fragments and FASTQ text of the device must equal
the oracle's.  Usage: python tools/stress_plain.py [n_trials] [gpu|emu] [tiny|p0]"""
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import parity_cases as P  # noqa: E402
from reseq_amd import synth  # noqa: E402

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = sys.argv[2] if len(sys.argv) > 2 else "gpu"
CFG = synth.P0 if len(sys.argv) > 3 and sys.argv[3] == "p0" else synth.TINY      # p0: the bench profile (2 x 150, inserts up to 1000)
if which == "gpu":
    from backends import GpuBackend as Backend
else:
    from backends import EmuBackend as Backend

bad = 0
with tempfile.TemporaryDirectory() as d:
    wd = pathlib.Path(d)
    for t in range(n_trials):
        rng = np.random.default_rng(5000 + t)
        n_seq = int(rng.integers(1, 6))
        lengths = [int(rng.integers(1001, 9000)) if rng.random() < 0.75 else int(rng.integers(30, 95)) for _ in range(n_seq)]
        if max(lengths) < 1001:
            lengths[-1] = 2345
        edits = {}
        r = rng.random()
        if r < 0.15:
            edits = {"no_indels": True}
        elif r < 0.3:
            edits = {"no_substitutions": True}
        elif r < 0.45:
            edits = {"error_multiplier": float(rng.choice([0.5, 2.0, 10.0]))}
        tag = f"pl{t}"
        cfg = dict(CFG)                                            # the indel draw's word-alone bound: other column orders, indel rates up to a few percent
        if rng.random() < 0.3:
            cfg["indel_columns_shuffled"] = True
        if rng.random() < 0.3:
            cfg["del_rate"] = cfg["del_rate"] * float(rng.choice([10.0, 300.0]))
            cfg["ins_rate"] = cfg["ins_rate"] * float(rng.choice([1.0, 100.0]))
        kw = dict(prof_seed=int(rng.integers(1, 1000)), ref_seed=int(rng.integers(1, 1000)), gc=float(rng.choice([0.25, 0.5, 0.7])))
        p = P.Pair(Backend, wd, tag, cfg, lengths, seed=int(rng.integers(1, 1 << 40)), num_pairs=int(rng.integers(500, 12000)), edits=edits or None,
                   ref_bias_mode=int(rng.choice([0, 1, 2])), **kw)
        try:
            if rng.random() < 0.3:
                names = [n.split(" ")[0] for n, _ in p.seqs]
                bed = wd / f"{tag}.bed"
                lines = []
                for si, L in enumerate(lengths):
                    if L > 1000:
                        a = int(rng.integers(0, L // 2))
                        lines.append(f"{names[si]}\t{a}\t{a + int(rng.integers(1, L // 3))}\t{rng.random():.3f}")
                bed.write_text("\n".join(lines) + "\n")
                p.b.read_methylation(bed)
                p.osim.read_methylation(bed)
            p.align_normalization()
            tb = p.info["total_blocks"]
            n, _ = P._compare_blocks(p, 1, tb + 1)
            ao = p.info["adapter_only_pairs"]
            if ao:
                o1, o2 = p.osim.adapter_only()
                b1, b2 = p.b.adapter_only_pairs(0, ao)
                assert o1 == b1 and o2 == b2
            print(f"trial {t}: ok  lengths {lengths} edits {edits} pairs {n} adapter-only {ao}")
        except AssertionError as e:
            bad += 1
            print(f"trial {t}: MISMATCH lengths {lengths} edits {edits} {kw}: {str(e)[:200]}")
        finally:
            p.close()
print("mismatching trials:", bad)
sys.exit(1 if bad else 0)
