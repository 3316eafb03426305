"""The memory-mapped, multi-threaded readers of the methylation and variant files (rsq_host.cpp read_methylation_mapped / read_variants_mapped)
against the line readers they stand in for: the same arrays for the files tools write, and -- because everything else is handed back to the line
reader -- the same result or the same message for every other file.  Through the test-only host library, no GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

from backends import emu_lib

NAMES = ["chrA", "chrB", "chrC", "chrD", "chrE"]
LENS = np.array([5000, 300, 8000, 8000, 1200], np.uint32)


def parse_bed(path, alleles, cap=4096):
    L = emu_lib()
    L.emu_parse_methylation_columns.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p]
    n = len(NAMES)
    nr, cols = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    first, second, rate = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap * alleles)
    rc = L.emu_parse_methylation_columns(str(path).encode(), ("\n".join(NAMES) + "\n").encode(), LENS.ctypes.data, n, alleles, nr.ctypes.data, first.ctypes.data,
                                         second.ctypes.data, rate.ctypes.data, cap, cols.ctypes.data)
    if rc:
        return ("error", L.emu_last_error().decode())
    k = int(nr.sum())
    assert k <= cap
    return ("ok", nr.tolist(), cols.tolist(), first[:k].tolist(), second[:k].tolist(), rate[:k * alleles].tobytes())


def bed_text(rng, alleles, sequences=(0, 2, 3, 4), per_sequence=60, notation="plain", sep="\t"):
    lines = []
    for i in sequences:
        cols = alleles if rng.random() < 0.6 else 1
        at = int(rng.integers(0, 5))
        for _ in range(per_sequence):
            start = at + int(rng.integers(0, 12))
            end = start + int(rng.integers(1, 30))
            if end > LENS[i]:
                break
            vals = []
            for _ in range(cols):
                v = rng.random()
                vals.append({"plain": f"{v:.{int(rng.integers(1, 9))}f}", "long": f"{v:.17f}", "exp": f"{v:.5e}", "int": str(int(v > 0.5))}[notation])
            lines.append(sep.join([NAMES[i], str(start), str(end)] + vals))
            if rng.random() < 0.05:
                lines.append("")
            at = end
    return "\n".join(lines) + "\n"


@pytest.fixture
def both_readers(rsq_options):
    def run(path, alleles):
        rsq_options("serial_parse", 1)
        serial = parse_bed(path, alleles)
        rsq_options("serial_parse", 0)
        mapped = 0
        for stretch in (7, 190, 100000):
            rsq_options("parse_stretch", stretch)
            before = emu_lib().emu_get_option(b"mapped_parses")
            assert parse_bed(path, alleles) == serial, stretch
            mapped += emu_lib().emu_get_option(b"mapped_parses") - before
        rsq_options("parse_stretch", 0)
        return serial + (mapped,)
    return run


@pytest.mark.parametrize("notation,sep", [("plain", "\t"), ("plain", " "), ("long", "\t"), ("exp", "  \t"), ("int", "\t")])
def test_methylation_readers_agree_on_well_formed_files(tmp_path, both_readers, notation, sep):
    rng = np.random.default_rng(5)
    for alleles in (1, 2, 3):
        p = tmp_path / f"m{alleles}.bed"
        head = "track type=bedGraph\n\ntrack again\n" if alleles == 2 else ""
        p.write_text(head + bed_text(rng, alleles, notation=notation, sep=sep))
        res = both_readers(p, alleles)
        assert res[0] == "ok" and sum(res[1]) > 100 and res[1][1] == 0 and res[-1] == 3


def test_methylation_fast_path_is_taken_and_matches_bit_for_bit(tmp_path, rsq_options):
    """the value of every rate: 1 - strtod(text), for decimals of up to 15 digits computed as integer / power of ten"""
    rng = np.random.default_rng(9)
    p = tmp_path / "m.bed"
    text = bed_text(rng, 2, per_sequence=200)
    p.write_text(text)
    rsq_options("parse_stretch", 64)
    res = parse_bed(p, 2)
    rsq_options("parse_stretch", 0)
    rates = np.frombuffer(res[5])
    exp, at = [], 0
    for line in text.split("\n"):
        if line:
            vals = [1.0 - float(v) for v in line.split("\t")[3:]]
            exp += vals if len(vals) == 2 else vals * 2
    assert np.array_equal(rates, np.array(exp))


MALFORMED = {
    "overlap": "chrA\t10\t20\t0.5\nchrA\t19\t30\t0.5\n",
    "touching regions are fine": "chrA\t10\t20\t0.5\nchrA\t20\t30\t0.5\n",
    "start beyond the sequence": "chrB\t300\t301\t0.5\n",
    "end beyond the sequence": "chrB\t290\t301\t0.5\n",
    "end at the sequence end": "chrB\t290\t300\t0.5\n",
    "end before start": "chrA\t10\t10\t0.5\n",
    "rate above one": "chrA\t10\t20\t1.5\n",
    "negative rate": "chrA\t10\t20\t-0.5\n",
    "negative start": "chrA\t-10\t20\t0.5\n",
    "no rate": "chrA\t10\t20\n",
    "no rate but a blank": "chrA\t10\t20\t\n",
    "three alleles of two": "chrA\t10\t20\t0.5\t0.5\t0.5\n",
    "alleles change within a sequence": "chrA\t10\t20\t0.5\t0.5\nchrA\t30\t40\t0.5\n",
    "alleles differ between sequences": "chrA\t10\t20\t0.5\t0.5\nchrC\t30\t40\t0.5\n",
    "unknown sequence": "chrA\t10\t20\t0.5\nchrX\t30\t40\t0.5\nchrC\t30\t40\t0.5\n",
    "sequences out of order": "chrC\t10\t20\t0.5\nchrA\t30\t40\t0.5\nchrD\t1\t2\t0.25\n",
    "sequence comes back": "chrA\t10\t20\t0.5\nchrC\t30\t40\t0.5\nchrA\t50\t60\t0.5\n",
    "carriage returns": "chrA\t10\t20\t0.5\r\nchrA\t30\t40\t0.25\r\n",
    "number with a tail": "chrA\t10x\t20\t0.5\nchrA\t30\t40\t0.25zz\n",
    "text for a number": "chrA\tten\t20\t0.5\n",
    "not a number": "chrA\t10\t20\tnan\n",
    "hexadecimal": "chrA\t10\t20\t0x0.8p0\n",
    "huge position": "chrA\t99999999999999999999\t20\t0.5\n",
    "underflow": "chrA\t10\t20\t1e-400\n",
    "blank line": "chrA\t10\t20\t0.5\n \nchrA\t30\t40\t0.5\n",
    "track line later": "chrA\t10\t20\t0.5\ntrack x\nchrC\t30\t40\t0.5\n",
    "only track lines": "track a\ntrack b\n",
    "no final line end": "chrA\t10\t20\t0.5\nchrA\t30\t40\t0.25",
    "name only": "chrA\n",
    "plus sign": "chrA\t+10\t20\t+0.5\n",
    "leading zeros and a bare point": "chrA\t0010\t020\t.5\t1.\n",
    "many digits": "chrA\t10\t20\t0.1234567890123456789\t0.500000000000000000000\n",
}


@pytest.mark.parametrize("what", list(MALFORMED))
def test_methylation_readers_agree_on_everything_else(tmp_path, both_readers, what):
    """the mapped reader gives such files back to the line reader: same arrays, or same message"""
    for text in (MALFORMED[what],):
        p = tmp_path / "m.bed"
        p.write_bytes(text.encode())
        res = both_readers(p, 2)
        if what in ("touching regions are fine", "end at the sequence end", "leading zeros and a bare point", "many digits", "carriage returns", "number with a tail",
                    "alleles differ between sequences", "unknown sequence", "sequences out of order", "sequence comes back", "blank line", "track line later", "no final line end", "plus sign", "not a number",
                    "hexadecimal"):
            assert res[0] == "ok", res
            assert res[-1] == (3 if what in ("touching regions are fine", "end at the sequence end", "leading zeros and a bare point", "many digits", "alleles differ between sequences", "hexadecimal") else 0)
        else:
            assert res[0] == "error", (what, res)


def test_methylation_malformed_line_deep_inside_a_large_file(tmp_path, both_readers):
    """the anomaly is in a piece of its own thread; the line reader's message names the line"""
    rng = np.random.default_rng(3)
    lines = bed_text(rng, 2, per_sequence=150).split("\n")
    bad = "chrC\t7000\t6000\t0.5"
    at = max(i for i, l in enumerate(lines) if l.startswith("chrC"))
    lines.insert(at + 1, bad)
    p = tmp_path / "m.bed"
    p.write_text("\n".join(lines))
    res = both_readers(p, 2)
    assert res[0] == "error" and bad in res[1]


# ------------------------------------------------------------------------------------------------ variants (VCF), through the product library (host code)
def _vcf_inputs(tmp_path, seed=4, lengths=(6000, 150, 9000, 4000)):
    import parity_cases as P
    from reseq_amd import synth
    rng = np.random.default_rng(seed)
    seqs = [(f"seq{i} some description", rng.integers(0, 4, L).astype(np.uint8)) for i, L in enumerate(lengths)]
    fa = tmp_path / "ref.fa"
    synth.write_fasta(fa, seqs)
    return P, seqs, fa, rng


def _read_vcf(fa, vcf, n_seqs):
    from reseq_amd import api
    ref = api.Reference(str(fa))
    try:
        alleles = ref.read_variants(str(vcf))
        return ("ok", alleles, [ref.variants(i) for i in range(n_seqs)])
    except api.RsqError as e:
        return ("error", str(e))
    finally:
        ref.close()


@pytest.fixture
def both_vcf_readers(rsq_options):
    from reseq_amd import api

    def run(fa, vcf, n_seqs):
        rsq_options("serial_parse", 1)
        serial = _read_vcf(fa, vcf, n_seqs)
        rsq_options("serial_parse", 0)
        mapped = 0
        for stretch in (11, 300, 1 << 20):
            rsq_options("parse_stretch", stretch)
            before = api.get_option("mapped_parses")
            assert _read_vcf(fa, vcf, n_seqs) == serial, stretch
            mapped += api.get_option("mapped_parses") - before
        rsq_options("parse_stretch", 0)
        return serial + (mapped,)
    return run


@pytest.mark.parametrize("samples", [1, 2])
def test_variant_readers_agree_on_sorted_files(tmp_path, both_vcf_readers, samples):
    P, seqs, fa, rng = _vcf_inputs(tmp_path)
    variants = P._mixed_variant_set(seqs, rng, 9, ends=30)
    if samples == 2:
        variants = [(si, pos, rl, alt, gt + "\t" + ["0|0", "1|1", "0/1"][k % 3]) for k, (si, pos, rl, alt, gt) in enumerate(variants)]
    # two alternatives in one record, alleles carrying the second
    si, pos = 2, 8990
    variants = [v for v in variants if not (v[0] == si and v[1] >= pos - 8)] + [(si, pos, 1, "ACGT"[(seqs[si][1][pos] + 1) % 4] + "," + "ACGT"[(seqs[si][1][pos] + 2) % 4] + "TT", "2|1" + ("\t1|2" if samples == 2 else ""))]
    variants.sort(key=lambda v: (v[0], v[1]))
    vcf = tmp_path / "v.vcf"
    P.write_vcf(vcf, seqs, variants, samples)
    res = both_vcf_readers(fa, vcf, len(seqs))
    assert res[0] == "ok" and res[1] == 2 * samples and sum(len(v) for v in res[2]) > 1500 and not res[2][1] and res[-1] == 3


def _edit(vcf, fn):
    lines = vcf.read_text().split("\n")
    head = [l for l in lines if l.startswith("#")]
    recs = [l for l in lines if l and not l.startswith("#")]
    vcf.write_bytes(("\n".join(head + fn(recs)) + "\n").encode())


VCF_EDITS = {
    "two records swapped": lambda r: r[:700] + [r[701], r[700]] + r[702:],
    "a sequence comes back": lambda r: r[1:] + r[:1],
    "overlap": lambda r: r[:500] + [r[500], "\t".join(r[500].split("\t")[:1] + [str(int(r[500].split("\t")[1]))] + r[500].split("\t")[2:])] + r[501:],
    "unknown contig": lambda r: r[:900] + [r[900].replace("seq", "chr", 1)] + r[901:],
    "REF differs": lambda r: r[:300] + ["\t".join(r[300].split("\t")[:3] + ["ACGTACGTAC"] + r[300].split("\t")[4:])] + r[301:],
    "N in ALT": lambda r: r[:300] + ["\t".join(r[300].split("\t")[:4] + ["N"] + r[300].split("\t")[5:])] + r[301:],
    "nine columns": lambda r: r[:1200] + ["\t".join(r[1200].split("\t")[:9])] + r[1201:],
    "letters in the genotype": lambda r: r[:100] + [r[100].rsplit("\t", 1)[0] + "\t.|1"] + r[101:],
    "three alleles in the genotype": lambda r: r[:100] + [r[100].rsplit("\t", 1)[0] + "\t0|1|1"] + r[101:],
    "one allele in the genotype": lambda r: r[:100] + [r[100].rsplit("\t", 1)[0] + "\t1"] + r[101:],
    "alternative that does not exist": lambda r: r[:100] + [r[100].rsplit("\t", 1)[0] + "\t0|3"] + r[101:],
    "position past the end": lambda r: r[:-1] + ["\t".join(r[-1].split("\t")[:1] + ["999999"] + r[-1].split("\t")[2:])],
    "carriage returns": lambda r: [l + "\r" for l in r],
    "empty lines": lambda r: r[:50] + ["", ""] + r[50:],
    "comment between records": lambda r: r[:50] + ["#comment"] + r[50:],
    "the same record twice": lambda r: r[:640] + [r[640]] + r[640:],
    "more than twenty complaints": lambda r: [l.rsplit("\t", 1)[0] + "\tx|1" if 200 <= k < 240 else l for k, l in enumerate(r)],
}


@pytest.mark.parametrize("what", list(VCF_EDITS))
def test_variant_readers_agree_on_everything_else(tmp_path, both_vcf_readers, what):
    P, seqs, fa, rng = _vcf_inputs(tmp_path)
    vcf = tmp_path / "v.vcf"
    P.write_vcf(vcf, seqs, P._mixed_variant_set(seqs, rng, 9))
    _edit(vcf, VCF_EDITS[what])
    res = both_vcf_readers(fa, vcf, len(seqs))
    if what in ("carriage returns", "empty lines"):
        assert res[0] == "ok" and res[-1] == 3
    else:
        assert res[0] == "error" and res[-1] == 0, res


def test_variant_readers_other_files(tmp_path, both_vcf_readers):
    """gzip, no final line end, a header that does not name the reference's sequences: the line reader's"""
    import gzip
    P, seqs, fa, rng = _vcf_inputs(tmp_path)
    vcf = tmp_path / "v.vcf"
    P.write_vcf(vcf, seqs, P._mixed_variant_set(seqs, rng, 9))
    plain = both_vcf_readers(fa, vcf, len(seqs))
    gz = tmp_path / "v.vcf.gz"
    gz.write_bytes(gzip.compress(vcf.read_bytes()))
    res = both_vcf_readers(fa, gz, len(seqs))
    assert res[:3] == plain[:3] and res[-1] == 0
    cut = tmp_path / "cut.vcf"
    cut.write_bytes(vcf.read_bytes()[:-1])
    res = both_vcf_readers(fa, cut, len(seqs))
    assert res[:3] == plain[:3] and res[-1] == 0
    other = tmp_path / "other.vcf"
    other.write_text(vcf.read_text().replace("##contig=<ID=seq1,", "##contig=<ID=seq9,"))
    res = both_vcf_readers(fa, other, len(seqs))
    assert res[0] == "error" and "Contigs at position 1" in res[1]


# ------------------------------------------------------------------------------------------------ random damage
def _damage(rng, text):
    """one to three random edits of a text file: a character changed, removed or inserted, a line removed, doubled or moved"""
    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 6))
        if kind < 3 and text:
            at = int(rng.integers(0, len(text)))
            alphabet = "0123456789.\t \n-+eE\rxNACGT|/:chrseq"
            c = alphabet[int(rng.integers(0, len(alphabet)))]
            text = text[:at] + (c + text[at + 1:] if kind == 0 else text[at + 1:] if kind == 1 else c + text[at:])
        else:
            lines = text.split("\n")
            if len(lines) < 3:
                continue
            a, b = int(rng.integers(0, len(lines) - 1)), int(rng.integers(0, len(lines) - 1))
            if kind == 3:
                del lines[a]
            elif kind == 4:
                lines.insert(b, lines[a])
            else:
                lines.insert(b, lines.pop(a))
            text = "\n".join(lines)
    return text


def test_methylation_readers_agree_on_randomly_damaged_files(tmp_path, both_readers):
    rng = np.random.default_rng(77)
    good = bed_text(rng, 2, per_sequence=25)
    outcomes = {"ok": 0, "error": 0, "mapped": 0}
    for k in range(250 * int(os.environ.get("RSQ_FUZZ", "1"))):          # RSQ_FUZZ=20: a longer run by hand
        p = tmp_path / "d.bed"
        p.write_bytes(_damage(rng, good).encode())
        res = both_readers(p, 2)                                    # asserts that the mapped reader (three piece lengths) gives what the line reader gives
        outcomes[res[0]] += 1
        outcomes["mapped"] += res[-1] > 0
    assert outcomes["ok"] > 30 and outcomes["error"] > 30 and outcomes["mapped"] > 10, outcomes


def test_variant_readers_agree_on_randomly_damaged_files(tmp_path, both_vcf_readers):
    P, seqs, fa, rng = _vcf_inputs(tmp_path, lengths=(1500, 150, 2200))
    vcf = tmp_path / "v.vcf"
    P.write_vcf(vcf, seqs, P._mixed_variant_set(seqs, rng, 25))
    good = vcf.read_text()
    outcomes = {"ok": 0, "error": 0, "mapped": 0}
    for k in range(120 * int(os.environ.get("RSQ_FUZZ", "1"))):
        p = tmp_path / "d.vcf"
        p.write_bytes(_damage(rng, good).encode())
        res = both_vcf_readers(fa, p, len(seqs))
        outcomes[res[0]] += 1
        outcomes["mapped"] += res[-1] > 0
    assert outcomes["ok"] > 5 and outcomes["error"] > 30, outcomes
