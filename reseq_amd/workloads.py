"""The synthetic inputs of BASELINE.json's configs[2]-[4] as SURVEY.md section 8(d) restates them (no real genomes, call sets or profiles exist without a network):
one definition for bench.py's `other_configs` leg and for the full-size tools (tools/run_config4.py, tools/run_config5.py, tools/bench_error_model.py)."""
import os

import numpy as np

from . import synth

DROSOPHILA = [32_079_331, 28_110_227, 25_286_936, 23_542_271, 23_513_712, 7_350_000 + 3_667_352 - 1_000 * 600, 1_348_131]     # 143.7 Mb with the 1000 scaffolds of 600 bp
HUMAN = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
         101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
P0_SEED = 103741084                       # ProbabilityEstimatesTest.cpp:1006, SURVEY.md section 8(d)


def p0_profile(path, n_ref_seqs=1):
    arrays = synth.make_profile(synth.P0, seed=P0_SEED, n_ref_seqs=n_ref_seqs)
    synth.write_profile(path, arrays)
    return arrays


def seq_to_illumina_rows(n, arrays, seed=3, distinct=250_000):
    """configs[2]: n records of 150 bases as seqToIllumina's FASTA text, one fixed-width row per record (ids r000000000 ...): segment alternating, fragment lengths from
    P0's insert lengths (held to three digits), dominant errors from P0's marginals, 97 % of the rates 0.  `distinct` different records, repeated under their own ids
    up to n: a record's random streams are selected by its index in the input, so every record simulates to its own read (drawing 8 M x 150 x 3 values with numpy takes
    minutes).  Returns (byte matrix [n, row width], the distinct records)."""
    distinct = min(distinct, n) & ~1                                 # whole pairs of segments
    rec = synth.make_error_model_input(seed, distinct, 150, arrays, zero_frac=0.97)
    rec["frag_len"] = np.clip(rec["frag_len"], 100, 999).astype(np.uint32)
    rows = synth.fixed_width_fasta(rec)
    rows = np.tile(rows, (-(-n // distinct), 1))[:n]
    synth.number_rows(rows, 0)
    return rows, rec


def drosophila_sized(directory, scale=1.0):
    """configs[3]: 143.7 Mb in 7 sequences plus 1000 scaffolds of 600 bases (shorter than the longest insert: no units, Simulator.cpp:1159), GC 42 %.  Returns
    (fasta path, sequence lengths); simulate with P0 and coverage 30 (about 14.4 M pairs)."""
    lengths = [max(5000, int(n * scale)) for n in DROSOPHILA] + [600] * max(1, int(1000 * scale))
    path = os.path.join(directory, "drosophila_sized.fa")
    synth.write_fasta(path, synth.make_reference(5, lengths, gc=0.42))
    return path, lengths


def human_sized(directory, scale=0.1, snv_only=False):
    """configs[4] at `scale`: 3.1 Gb x scale in 24 sequences (GC 41 %), a phased VCF of 4 M x scale substitutions and 0.4 M x scale insertions / deletions of at most 20
    bases on two alleles (non-overlapping, as a normalised call set), a BED of 20 M x scale unmethylated regions with Beta(0.5, 0.5) methylation per allele.
    Returns dict(fasta, vcf, bed, lengths, substitutions, indels, regions); simulate with P0 and coverage 30."""
    lengths = [max(5000, int(n * scale)) for n in HUMAN]
    rng = np.random.default_rng(11)
    fpath, vpath, bpath = (os.path.join(directory, n) for n in ("human_sized.fa", "human_sized.vcf", "human_sized.bed"))
    seqs = synth.make_reference(9, lengths, gc=0.41)
    synth.write_fasta(fpath, seqs)
    total = int(sum(lengths))
    n_sub, n_indel, n_regions = int(4.0e6 * scale), 0 if snv_only else int(0.4e6 * scale), 0 if snv_only else int(20e6 * scale)
    names = [n.split(" ")[0] for n, _ in seqs]
    letters = np.frombuffer(b"ACGT", np.uint8)
    with open(vpath, "w") as f:
        f.write("##fileformat=VCFv4.2\n" + "".join(f"##contig=<ID={n},length={len(c)}>\n" for n, (_, c) in zip(names, seqs)))
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n")
        for si, (_, codes) in enumerate(seqs):
            L = len(codes)
            k = int((n_sub + n_indel) * L / total)
            pos = np.unique(rng.integers(1, L - 50, k))
            pos = pos[np.concatenate(([True], np.diff(pos) > 25))]                 # non-overlapping, as a normalised VCF would have them
            kinds = rng.random(len(pos)) < n_indel / (n_sub + n_indel)
            gts = rng.integers(0, 3, len(pos))
            lines = []
            for p0, is_indel, g in zip(pos.tolist(), kinds.tolist(), gts.tolist()):
                gt = ("0|1", "1|0", "1|1")[g]
                ref = chr(letters[codes[p0]])
                if not is_indel:
                    alt = chr(letters[(codes[p0] + 1 + p0 % 3) % 4])
                    lines.append(f"{names[si]}\t{p0 + 1}\t.\t{ref}\t{alt}\t.\tPASS\t.\tGT\t{gt}")
                elif p0 & 1:
                    ins = letters[rng.integers(0, 4, 1 + p0 % 20)].tobytes().decode()
                    lines.append(f"{names[si]}\t{p0 + 1}\t.\t{ref}\t{ref + ins}\t.\tPASS\t.\tGT\t{gt}")
                else:
                    dl = 1 + p0 % 20
                    lines.append(f"{names[si]}\t{p0 + 1}\t.\t{letters[codes[p0:p0 + dl + 1]].tobytes().decode()}\t{ref}\t.\tPASS\t.\tGT\t{gt}")
            f.write("\n".join(lines) + "\n")
    with open(bpath, "w") as f:
        for si, (_, codes) in enumerate(seqs):
            L = len(codes)
            k = int(n_regions * L / total)
            starts = np.unique(rng.integers(0, L - 200, k))
            if not len(starts):
                continue
            starts = starts[np.concatenate(([True], np.diff(starts) > 120))]
            lens = rng.integers(1, 100, len(starts))
            meth = rng.beta(0.5, 0.5, (len(starts), 2))
            f.write("".join(f"{names[si]}\t{a}\t{a + b}\t{m0:.4f}\t{m1:.4f}\n" for a, b, (m0, m1) in zip(starts.tolist(), lens.tolist(), meth.tolist())))
    return dict(fasta=fpath, vcf=vpath, bed=bpath, lengths=lengths, substitutions=n_sub, indels=n_indel, regions=n_regions)
