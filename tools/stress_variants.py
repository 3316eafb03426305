#!/usr/bin/env python3
"""Randomised parity stress for the simulation with variants: random references (several sequences, some too short to get blocks), random
substitution / insertion / deletion sets of different densities, allele counts and seeds, with and without methylation; the device's
fragments and FASTQ text must equal the oracle's.  Usage: python tools/stress_variants.py [n_trials] [gpu|emu] [tiny|p0]"""
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
        rng = np.random.default_rng(1000 + t)
        n_seq = int(rng.integers(1, 5))
        lengths = [int(rng.integers(1001, 7000)) if rng.random() < 0.8 else int(rng.integers(40, 90)) for _ in range(n_seq)]
        if max(lengths) < 1001:
            lengths[0] = 3456
        density = int(rng.choice([6, 12, 25, 60]))
        samples = int(rng.choice([1, 1, 2, 3]))
        kind = int(rng.integers(0, 4))
        tag = f"st{t}"
        seqs = P.make_inputs(wd, tag, CFG, lengths, ref_seed=500 + t)[2]
        maker = (P._mixed_variant_set, P._complex_variant_set, lambda s, r, dd: P._substitution_set(s, r, dd, [0, 1, 999, 1000]),
                 lambda s, r, dd: P._mixed_variant_set(s, r, dd, ends=45))[kind]
        vs = maker(seqs, rng, max(density, 25) if kind == 1 else density)
        if samples > 1:
            vs = [(si, p0, rl, alt, "\t".join(["0|1", "1|0", "1|1", "0|0"][int(rng.integers(0, 4))] for _ in range(samples - 1)) + "\t" + gt) for si, p0, rl, alt, gt in vs
                  if "," not in alt]
        vcf = wd / f"{tag}.vcf"
        P.write_vcf(vcf, seqs, vs, samples=samples)
        p = P.Pair(Backend, wd, tag, CFG, lengths, seed=int(rng.integers(1, 1 << 30)), num_pairs=int(rng.integers(2000, 9000)), vcf=vcf, ref_seed=500 + t)
        try:
            if rng.random() < 0.4:
                names = [n.split(" ")[0] for n, _ in seqs]
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
            ofr, _ = P._compare_blocks_var(p, 1, p.info["total_blocks"] + 1)
            print(f"trial {t}: ok  lengths {lengths} kind {kind} density {density} alleles {p.ovars.contents.num_alleles} pairs {len(ofr)} starts inside insertions {int((ofr['sub'] > 0).sum())}")
        except RuntimeError as e:
            if "systematic-error walk left the sequence" not in str(e):
                raise
            try:                                                 # the oracle says the reference cannot simulate this set: the product has to say so too
                p.b.pairs(1, p.info["total_blocks"] + 1)
                bad += 1
                print(f"trial {t}: the oracle reports a walk off the sequence, the product does not")
            except Exception as e2:
                ok = "systematic-error walk left the sequence" in str(e2)
                bad += 0 if ok else 1
                print(f"trial {t}: walk off the sequence reported by both" if ok else f"trial {t}: unexpected product error {e2}")
        except AssertionError as e:
            bad += 1
            print(f"trial {t}: MISMATCH lengths {lengths} kind {kind} density {density}: {str(e)[:200]}")
        finally:
            p.close()
print("mismatching trials:", bad)
sys.exit(1 if bad else 0)
