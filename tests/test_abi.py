"""The C-ABI library loads, exports every symbol include/reseq_amd.h declares, its host-only entry points work
without a GPU, and the simulation entry points fail loudly (no CPU fallback) when no device is visible."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, read_fasta
from reseq_amd import api

HEADER = os.path.join(ROOT, "include", "reseq_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(api.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), n
    assert sorted(api.SYMBOLS) == names                     # the Python binding covers the whole header


def test_version_and_error_strings():
    L = api.lib()
    assert b"gfx950" in L.rsq_version()
    h = C.c_void_p()
    assert L.rsq_profile_load(b"/nonexistent/profile.rsqp", C.byref(h)) == api.RSQ_EIO
    assert b"cannot open" in L.rsq_last_error()


def test_reference_loading_matches_the_reference_tests_fixture():
    ref = api.Reference(os.path.join(GOLDEN, "reference-test.fa"))
    exp = read_fasta(os.path.join(GOLDEN, "reference-test.fa"))
    assert ref.num_sequences() == 2
    assert [ref.sequence_length(i) for i in range(2)] == [500, 501]        # ReferenceTest.cpp:275-276
    for i in range(2):
        assert np.array_equal(ref.codes(i), exp[i][1])
    ref.close()


def test_replace_n_is_seeded_and_complete(workdir):
    path = workdir / "withn.fa"
    path.write_text(">a\nACGTNNNNACGTRYKM\nnnAC\n>b\nNNNN\n")
    r1, r2, r3 = (api.Reference(path, s) for s in (5, 5, 6))
    a, b, c = r1.codes(0), r2.codes(0), r3.codes(0)
    assert a.max() <= 3 and np.array_equal(a, b) and not np.array_equal(a, c)
    assert a[:4].tolist() == [0, 1, 2, 3] and a[8:12].tolist() == [0, 1, 2, 3]
    raw = api.Reference(path).codes(0)
    assert (raw == 4).sum() == 10                               # IUPAC codes other than ACGT load as N
    for r in (r1, r2, r3):
        r.close()


def test_profile_loading_and_edits(tiny_profile_path):
    p = api.Profile(tiny_profile_path)
    assert p.max_read_length() == 30
    p.change_error_rate(2.0)
    p.remove_substitution_errors()
    p.remove_indel_errors()
    with pytest.raises(api.RsqError):
        p.change_error_rate(0.0)
    p.close()


def test_truncated_profile_is_rejected(tiny_profile_path, workdir):
    data = open(tiny_profile_path, "rb").read()
    bad = workdir / "truncated.rsqp"
    bad.write_bytes(data[: len(data) // 2])
    with pytest.raises(api.RsqError) as e:
        api.Profile(bad)
    assert e.value.code == api.RSQ_EIO


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_a_gpu(tiny_profile_path):
    p = api.Profile(tiny_profile_path)
    with pytest.raises(api.RsqError) as e:
        api.Simulator(p, None, 0)
    assert e.value.code == api.RSQ_ENODEV
    assert "no CPU fallback" in str(e.value)
    p.close()


def test_read_kernel_compiles_for_a_profile_without_a_gpu(tiny_profile_path, workdir):
    """rsq_profile_compile_read_kernel: the library carries the read kernels' source and hiprtc compiles it with the profile's plan as literals -- host only (the
    compiler cross-compiles), so the build of the run-time route is checked here; the code object holds the kernel, and a second request comes from the kernel cache"""
    import subprocess
    api.set_kernel_cache_dir(str(workdir / "kernel_cache"))
    try:
        p = api.Profile(tiny_profile_path)
        for kind, var, name in ((0, False, "rsq_spec_fill_reads"), (0, True, "rsq_spec_fill_reads"), (1, False, "rsq_spec_fill_records")):
            out = workdir / f"spec_{kind}_{int(var)}.hsaco"
            size, seconds = p.compile_read_kernel(kind, with_variants=var, out_path=str(out))
            assert size == out.stat().st_size > 10000 and seconds > 0
            syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--symbols", str(out)], capture_output=True, text=True, check=True).stdout
            assert any(l.split()[-1] == name and " FUNC " in l for l in syms.splitlines() if l.strip()), syms[-2000:]
            size2, seconds2 = p.compile_read_kernel(kind, with_variants=var)
            assert size2 == size and seconds2 == 0.0                       # from the cache directory
        assert len(list((workdir / "kernel_cache").glob("rsq_spec_*.hsaco"))) == 3
        with pytest.raises(api.RsqError) as e:                            # the compiler's complaint reaches the caller
            p.compile_read_kernel(0, arch="gfx000")
        assert "failed" in str(e.value)
        p.close()
    finally:
        api.set_kernel_cache_dir(str(workdir / "kernel_cache_session"))     # what the session's fixture had set up is gone; any directory of this run does


def test_gzip_fasta_loads_like_plain(workdir):
    """SeqAn opens .gz references transparently (SURVEY.md section 8(b) inputs); so does rsq_ref_load_fasta"""
    import gzip
    plain = os.path.join(GOLDEN, "reference-test.fa")
    packed = workdir / "reference-test.fa.gz"
    with open(plain, "rb") as f, gzip.open(packed, "wb") as g:
        g.write(f.read())
    a, b = api.Reference(plain), api.Reference(str(packed))
    assert a.num_sequences() == b.num_sequences() == 2
    for i in range(2):
        assert np.array_equal(a.codes(i), b.codes(i))
    a.close()
    b.close()


def test_mapped_fasta_reader_reads_what_the_line_reader_reads(workdir, rsq_options):
    """large plain FASTA files are memory-mapped and converted by several threads in stretches (rsq_host.cpp read_fasta_mapped); here the
    stretches are a few bytes, so that headers, line ends, "\\r\\n", blanks and '>' inside a line fall on stretch borders"""
    rng = np.random.default_rng(3)
    letters = "ACGTacgtNnRYKMUu"
    texts = []
    for trial in range(12):
        parts = ["\n\r\n"] if trial % 3 == 0 else []
        for s in range(int(rng.integers(1, 5))):
            parts.append(f">seq{s} trial {trial} with > inside" + ("\r\n" if trial & 1 else "\n"))
            for _ in range(int(rng.integers(0, 9))):
                line = "".join(letters[k] for k in rng.integers(0, len(letters), int(rng.integers(0, 70))))
                if rng.random() < 0.2:
                    line = line[:5] + " \t" + line[5:] + ">"
                if rng.random() < 0.1:
                    line += "\rA"                                       # a carriage return inside a line is a character (N), not a line end
                parts.append(line + ("\r\n" if trial & 1 else "\n"))
        text = "".join(parts)
        if trial % 4 == 2:
            text = text.rstrip("\n")                                    # no newline at the end of the file
        if trial % 4 == 3 and text.endswith("\r\n"):
            text = text[:-1]                                            # ends with a lone carriage return
        texts.append(text)
    for i, text in enumerate(texts):
        path = workdir / f"tricky{i}.fa"
        path.write_bytes(text.encode())
        rsq_options("serial_fasta", 1)
        a = api.Reference(str(path))
        rsq_options("serial_fasta", 0)
        for stretch in (1, 7, 64, 1 << 20):
            rsq_options("fasta_stretch", stretch)
            b = api.Reference(str(path))
            assert a.num_sequences() == b.num_sequences(), (i, stretch)
            for k in range(a.num_sequences()):
                assert a.sequence_name(k) == b.sequence_name(k), (i, stretch, k)
                assert np.array_equal(a.codes(k), b.codes(k)), (i, stretch, k)
            b.close()
        rsq_options("fasta_stretch", 0)
        a.close()
    # text before the first header: refused by both readers
    bad = workdir / "bad.fa"
    bad.write_bytes(b"ACGT\n>s\nACGT\n")
    for env in ({"serial_fasta": 1}, {"fasta_stretch": 5}):
        for k, v in env.items():
            rsq_options(k, v)
        with pytest.raises(api.RsqError) as e:
            api.Reference(str(bad))
        assert "FASTA header" in str(e.value)
        for k in env:
            rsq_options(k, 0)


def test_bzip2_fasta_in_and_out(workdir):
    """the reference also reads and writes .bz2 (SeqAn with BZip2, ReferenceTest.BZip2); libbz2 is bound at run time (rsq_textio.h)"""
    import bz2
    plain = os.path.join(GOLDEN, "reference-test.fa")
    packed = workdir / "reference-test.fa.bz2"
    packed.write_bytes(bz2.compress(open(plain, "rb").read()))
    a, b = api.Reference(plain), api.Reference(str(packed))
    assert a.num_sequences() == b.num_sequences() == 2
    for i in range(2):
        assert np.array_equal(a.codes(i), b.codes(i))
    out = workdir / "again.fa.bz2"
    b.write_fasta(out)
    assert out.read_bytes()[:3] == b"BZh"
    c = api.Reference(str(out))
    for i in range(2):
        assert np.array_equal(a.codes(i), c.codes(i))
    for r in (a, b, c):
        r.close()


def test_replace_n_cli_mode_and_write_fasta(workdir):
    """`reseq replaceN -r in -R out --seed s` (main.cpp:611-692): ReadFasta, ReplaceN, WriteFasta -- host code only, no GPU needed"""
    import subprocess
    src = workdir / "n_in.fa"
    src.write_text(">chr1 some description\nACGTNNNNACGT" + "N" * 120 + "GGCC\n>chr2\nNNAC\n")
    out = workdir / "n_out.fa.gz"
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reseq_amd", "reseq")
    subprocess.run([exe, "replaceN", "-r", str(src), "-R", str(out), "--seed", "4"], check=True, capture_output=True)
    a = api.Reference(str(src), 4)
    b = api.Reference(str(out))
    assert b.num_sequences() == 2
    for i in range(2):
        assert np.array_equal(a.codes(i), b.codes(i)) and b.codes(i).max() <= 3
    again = workdir / "n_again.fa"
    b.write_fasta(again)
    lines = again.read_text().split("\n")
    assert lines[0] == ">chr1 some description" and len(lines[1]) == 70 and lines[1].startswith("ACGT")
    assert subprocess.run([exe, "replaceN", "-r", str(src)], capture_output=True).returncode == 1
    a.close()
    b.close()


def test_query_profile_cli_mode(workdir, tiny_profile_path):
    """`reseq queryProfile -s profile [-r ref] --maxLenDeletion --maxReadLength --refSeqBias -` (main.cpp:481-610); host code only"""
    import subprocess
    from reseq_amd import synth
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reseq_amd", "reseq")
    r = subprocess.run([exe, "queryProfile", "-s", tiny_profile_path, "--maxReadLength", "--maxLenDeletion"], capture_output=True, text=True)
    assert r.returncode == 0 and "maxReadLength: 30" in r.stdout and "maxLenDeletion: " in r.stdout
    arrays = synth.make_profile(synth.TINY, seed=5, n_ref_seqs=2)
    arrays["frag.ref_seq_bias"] = np.array([1.5, 0.25])
    ppath = workdir / "two_seq.rsqp"
    synth.write_profile(ppath, arrays)
    fa = workdir / "two_seq.fa"
    synth.write_fasta(fa, synth.make_reference(1, [300, 200], names=["chrA first sequence", "chrB"]))
    r = subprocess.run([exe, "queryProfile", "-s", str(ppath), "-r", str(fa), "--refSeqBias", "-"], capture_output=True, text=True)
    lines = [l.split("\t") for l in r.stdout.strip().split("\n")]
    assert r.returncode == 0 and [l[0] for l in lines] == ["chrA", "chrB"]
    assert np.allclose([float(l[1]) for l in lines], arrays["frag.ref_seq_bias"], rtol=1e-5)
    assert subprocess.run([exe, "queryProfile", "-s", tiny_profile_path], capture_output=True).returncode == 1      # no output option selected
    # -S: the loaded profile as ReSeq's own pair of files (rsq_profile_save_reseq), which loads again and answers the same
    out = workdir / "written.reseq"
    r = subprocess.run([exe, "queryProfile", "-s", tiny_profile_path, "-S", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and out.exists() and (workdir / "written.reseq.ipf").exists(), r.stderr
    r = subprocess.run([exe, "queryProfile", "-s", str(out), "--maxReadLength", "--maxLenDeletion"], capture_output=True, text=True)
    assert r.returncode == 0 and "maxReadLength: 30" in r.stdout and "maxLenDeletion: " in r.stdout


def test_the_references_own_fixtures_load(workdir):
    """The reference's test data as it lies in its test/ directory (byte-identical copies under tests/golden/): reference-test.fa.gz and .bz2 written by the reference's
    authors load like the plain file (ReferenceTest GZip / BZip2), every IUPAC code of reference-special-chars.fa loads as N (ReferenceTest.cpp:263-266), and
    rsq_ref_sequence_name returns ReferenceIdFirstPart (ReferenceTest.cpp:328-329) -- the string that goes into every read id."""
    ka = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))
    plain = api.Reference(os.path.join(GOLDEN, "reference-test.fa"))
    assert [plain.sequence_name(i) for i in range(2)] == ka["reference_ids"]["first_part"]
    assert [plain.sequence_length(i) for i in range(2)] == ka["reference_ids"]["lengths"]
    import bz2
    import gzip
    for packed, opener in (("reference-test.fa.gz", gzip.open), ("reference-test.fa.bz2", bz2.open)):
        r = api.Reference(os.path.join(GOLDEN, packed))
        with opener(os.path.join(GOLDEN, packed), "rb") as f:           # the packed fixtures name their sequences "NC_000913.3:1-500 ...", the plain one "NC_000913.3_1-500 ..."
            names = [line[1:].split()[0].decode() for line in f if line.startswith(b">")]
        assert r.num_sequences() == 2 and [r.sequence_name(i) for i in range(2)] == names and names[0].startswith("NC_000913.3")
        for i in range(2):
            assert np.array_equal(plain.codes(i), r.codes(i)), (packed, i)
        r.close()
    plain.close()
    g = ka["reference_special_chars"]
    special = api.Reference(os.path.join(GOLDEN, g["file"]))
    assert special.num_sequences() == 1 and special.sequence_length(0) == g["n_bases"] and (special.codes(0) == 4).all()
    special.close()


def test_variants_of_the_reference_sequence_known_answers(workdir):
    """the VCF that encodes the variants of ReferenceTest.cpp:283-326 (a deletion with its anchor base, two insertions, a substitution, on two alleles) loads as exactly
    the Reference::Variant list the test builds by hand; the templates themselves are checked on the device (tests/test_parity_gpu.py)"""
    ka = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))["reference_sequence_with_variants"]
    vcf = workdir / "reference_sequence_known_answers.vcf"
    write_known_answer_vcf(vcf, ka)
    r = api.Reference(os.path.join(GOLDEN, "reference-test.fa"))
    assert r.read_variants(vcf) == 2
    assert r.variants(0) == [tuple(v) for v in ka["variants"] + [ka["added_variant"]]] and r.variants(1) == []
    r.close()


def write_known_answer_vcf(path, ka):
    lines = ["##fileformat=VCFv4.2", "##contig=<ID=NC_000913.3_1-500,length=500>", "##contig=<ID=NC_000913.3_10000-10500,length=501>",
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">', "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample"]
    lines += [f"{chrom}\t{pos}\t.\t{ref}\t{alt}\t.\tPASS\t.\tGT\t{gt}" for chrom, pos, ref, alt, gt in ka["vcf_records"]]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def test_partition_blocks_is_the_launchers_rule():
    """rsq_partition_blocks (what `reseq illuminaPE --gpus N` cuts its workers' ranges with) against reseq_amd.sharding.partition_blocks (the N-process launcher's):
    the same bounds for any weights -- uniform, biased per sequence, zeros, more workers than blocks, no blocks at all"""
    from reseq_amd import sharding
    rng = np.random.default_rng(17)
    cases = [(0, 3, None), (1, 1, None), (5, 8, None), (4800, 8, None)]
    for _ in range(200):
        n = int(rng.integers(1, 400))
        kind = rng.integers(0, 4)
        if kind == 0:
            w = rng.random(n)
        elif kind == 1:                                      # a few sequences, each with its bias
            w = np.repeat(rng.choice([0.25, 1.0, 2.0, 7.5], size=8), rng.integers(1, 80, size=8))[:n]
            n = len(w)
        elif kind == 2:
            w = rng.random(n) * (rng.random(n) < 0.3)        # many blocks without weight
        else:
            w = np.full(n, 0.1)                              # sums that are not exact in binary
        cases.append((n, int(rng.integers(1, 17)), w))
    for n, world, w in cases:
        want = sharding.partition_blocks(n, world, None if w is None else [float(x) for x in w])
        got = api.partition_blocks(n, world, w)
        assert got == want, (n, world)
        assert got[0][0] == 1 and got[-1][1] == n + 1 and all(a[1] == b[0] for a, b in zip(got, got[1:]))


def test_bgzf_end_of_file_member():
    """rsq_gzip_eof_member: the 28 bytes of the SAM specification's BGZF end-of-file block -- a complete gzip member of no text with the "BC" extra field"""
    import zlib
    eof = api.gzip_eof_member()
    assert len(eof) == 28 and eof[:4] == b"\x1f\x8b\x08\x04" and eof[12:14] == b"BC" and int.from_bytes(eof[16:18], "little") == 27
    d = zlib.decompressobj(31)
    assert d.decompress(eof) == b"" and d.eof and d.unused_data == b""
    assert zlib.decompress(zlib.compress(b"x") , 15) == b"x"
