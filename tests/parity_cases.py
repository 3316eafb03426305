"""Parity cases shared by the CPU emulation run (`-m "not gpu"`) and the GPU run (`-m gpu`).

Every case builds the same inputs for the oracle (oracle/liboracle.so) and for a backend (tests/backends.py) and
compares: bit-exact for bytes / integers / indices (systematic-error tracks, fragment lists, FASTQ text, CIGARs),
relative 1e-11 for the double-precision bias-normalisation results whose summation order differs (tree reduction
on the device, sequential in the reference).
"""
import numpy as np

import oracle_lib as O
from reseq_amd import synth

NORM_RTOL = 1e-11      # a14: device tree-reduction vs sequential sum of ~1e6 positive doubles


def make_inputs(workdir, tag, cfg, ref_lengths, prof_seed=5, ref_seed=1, gc=0.5):
    ppath = workdir / f"{tag}.rsqp"
    fpath = workdir / f"{tag}.fa"
    if not ppath.exists():
        synth.write_profile(ppath, synth.make_profile(cfg, seed=prof_seed, n_ref_seqs=len(ref_lengths)))
    seqs = synth.make_reference(ref_seed, ref_lengths, gc=gc)
    if not fpath.exists():
        synth.write_fasta(fpath, seqs)
    return str(ppath), str(fpath), seqs


class Pair:
    """oracle + backend on the same inputs"""

    def __init__(self, backend_cls, workdir, tag, cfg, ref_lengths, seed, num_pairs=0, coverage=0.0, base_identifier="", edits=None, ref_bias_mode=0,
                 ref_bias_file=None, vcf=None, backend_profile=None, **kw):
        self.ppath, self.fpath, self.seqs = make_inputs(workdir, tag, cfg, ref_lengths, **kw)
        self.oprof = O.Profile(self.ppath)
        if edits:
            L = O.lib()
            if edits.get("error_multiplier", 1.0) != 1.0:
                L.orc_profile_change_error_rate(self.oprof.h, edits["error_multiplier"])
            if edits.get("no_substitutions"):
                L.orc_profile_remove_substitution_errors(self.oprof.h)
            if edits.get("no_indels"):
                L.orc_profile_remove_indel_errors(self.oprof.h)
        self.oref = O.Reference(self.seqs)
        self.ovars = None
        if vcf:
            import ctypes as C
            err = C.create_string_buffer(2048)
            self.ovars = O.lib().orc_read_variants(str(vcf).encode(), self.oref.h, err, len(err))
            if not self.ovars:
                raise RuntimeError(err.value.decode())
        self.osim = O.Sim(self.oprof, self.oref, seed, num_pairs, coverage, base_identifier.encode(), ref_bias_mode, ref_bias_file, variants=self.ovars)
        bprof = backend_profile or self.ppath             # the backend may read the same profile from ReSeq's own files
        self.b = backend_cls(bprof, self.fpath, 0, edits, vcf_path=vcf) if vcf else backend_cls(bprof, self.fpath, 0, edits)
        if ref_bias_file:
            self.b.set_ref_bias_file(ref_bias_file)
        self.info = self.b.prepare(seed, num_pairs, coverage, ref_bias_mode, base_identifier)

    def align_normalization(self):
        """stage-wise parity: give the backend the oracle's thresholds so that the sieve sees identical doubles"""
        self.b.set_normalization(self.osim.bias_normalization(), self.osim.thresholds())

    def close(self):
        self.b.close()
        self.osim.close()
        if self.ovars:
            O.lib().orc_variants_free(self.ovars)
        self.oref.close()
        self.oprof.close()


def case_reference_packing(backend_cls, workdir):
    ppath, fpath, seqs = make_inputs(workdir, "pack", synth.TINY, [777, 33, 1025, 64])
    b = backend_cls(ppath, fpath)
    for i, (_, codes) in enumerate(seqs):
        assert np.array_equal(b.codes(i, len(codes)), codes)
    b.close()


def case_prepass(backend_cls, workdir):
    # three sequences, the middle one shorter than the longest insert (no unit: Simulator.cpp:1159)
    p = Pair(backend_cls, workdir, "prepass", synth.TINY, [4100, 70, 2999], seed=11, num_pairs=4000)
    try:
        assert p.info["total_pairs"] == p.osim.total_pairs()
        assert p.info["adapter_only_pairs"] == p.osim.adapter_only_pairs()
        assert p.info["total_blocks"] == p.osim.total_blocks() == 5 + 3
        np.testing.assert_allclose(p.b.norm_by_len(), p.osim.norm_by_len(), rtol=NORM_RTOL, atol=0)
        np.testing.assert_allclose(p.b.thresholds(), p.osim.thresholds(), rtol=NORM_RTOL, atol=0)
        assert abs(p.info["bias_normalization"] / p.osim.bias_normalization() - 1) < NORM_RTOL
        for seq in (0, 2):
            n = len(p.seqs[seq][1])
            for strand in (0, 1):
                od, orate = p.osim.sys_errors(strand, seq)
                bd, brate = p.b.sys_errors(strand, seq, n)
                assert np.array_equal(od, bd) and np.array_equal(orate, brate), (seq, strand)
        arr = synth.make_profile(synth.TINY, seed=5)
        for seg in (0, 1):
            ptr = arr[f"adapters.{seg}.seq_ptr"]
            for a in range(len(ptr) - 1):
                n = int(ptr[a + 1] - ptr[a])
                o = p.osim.adapter_sys_errors(seg, a, n)
                bd, brate = p.b.adapter_sys_errors(seg, a, n)
                assert np.array_equal(o[0], bd) and np.array_equal(o[1], brate), (seg, a)
        assert p.info["passes"] >= 2
    finally:
        p.close()


def _compare_blocks(p, lo, hi):
    ofr = p.osim.sieve(lo, hi)
    o1, o2 = p.osim.create_reads(ofr)
    bfr, b1, b2 = p.b.pairs(lo, hi)
    assert len(ofr) == len(bfr)
    assert ofr.tobytes() == bfr.tobytes()          # start positions, lengths, strands, duplicates, ids: bit-identical
    assert o1 == b1
    assert o2 == b2
    return len(ofr), o1


def case_sieve_and_reads_tiny(backend_cls, workdir):
    p = Pair(backend_cls, workdir, "tiny_e2e", synth.TINY, [5000, 80, 3210], seed=7, num_pairs=3000)
    try:
        p.align_normalization()
        tb = p.info["total_blocks"]
        n_all, text = _compare_blocks(p, 1, tb + 1)
        assert 2000 < n_all < 4000
        # batching by block range does not change anything
        n_a, _ = _compare_blocks(p, 1, 4)
        n_b, _ = _compare_blocks(p, 4, tb + 1)
        assert n_a + n_b == n_all
        assert _compare_blocks(p, 3, 3)[0] == 0               # empty range
        # the profile exercises indels, adapters, several tiles and variable read lengths
        assert b"D" in text and b"I" in text and b"S" in text and b"H" in text
        assert b":1102:" in text and b":2308:" in text
        lens = {len(l) for l in text.split(b"\n")[1::4]}
        assert len(lens) > 1
    finally:
        p.close()


def case_sieve_dense_thresholds(backend_cls, workdir):
    """zero thresholds so small that nearly every cell passes: the sieve's running product of thresholds is restarted several times
    within the fragment lengths (segments), and one threshold is exactly zero (a length that always passes)"""
    p = Pair(backend_cls, workdir, "dense_thr", synth.TINY, [2500], seed=19, num_pairs=2500)
    try:
        thr = p.osim.thresholds()
        to = thr.shape[1]
        thr[0, :, 1] = np.geomspace(3e-1, 1e-9, to)
        thr[0, :, 0] = np.sqrt(thr[0, :, 1])
        thr[0, to // 2, :] = 0.0
        p.osim.set_normalization(p.osim.bias_normalization(), thr)
        p.b.set_normalization(p.osim.bias_normalization(), thr)
        n_all, _ = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert n_all > 1000
    finally:
        p.close()


def case_sharded_prepare(backend_cls, workdir, world=3, variants=False, seed=23):
    """the pre-passes of a sharded job (rsq_sim_prepare_plan .. rsq_sim_prepare_finish, sharding.sharded_prepare_in_process): `world` simulators
    each compute their share; thresholds equal a whole pre-pass exactly, and every rank's block range simulates to the whole simulator's text.
    variants: with insertions / deletions on two alleles, dense enough that variants sit within five bases of each other across the shard
    borders (the variants' own systematic errors are then computed per share too)"""
    from reseq_amd import sharding
    lengths = [9400, 80, 3210, 1000]
    tag = "shardprepv" if variants else "shardprep"
    ppath, fpath, seqs = make_inputs(workdir, tag, synth.TINY, lengths)
    kw = {}
    if variants:
        vcf = workdir / f"{tag}.vcf"
        borders = [x + d for x in range(1000, 9000, 1000) for d in (-6, -3, 0, 2, 5)]
        write_vcf(vcf, seqs, _mixed_variant_set(seqs, np.random.default_rng(seed), 14, borders))
        kw = dict(vcf_path=vcf)
    whole = backend_cls(ppath, fpath, **kw)
    ranks = [backend_cls(ppath, fpath, **kw) for _ in range(world)]
    try:
        winfo = whole.prepare(seed, 20000, 0.0, 1, "Pre")
        for b in ranks:
            b.seq_len = lengths
            b.ref_seq_bias_ = b.ref_seq_bias
            b.ref_seq_bias = lambda b=b: b.ref_seq_bias_(len(lengths))
        infos, ranges, rounds = sharding.sharded_prepare_in_process(ranks, seed, 20000, 0.0, 1, "Pre")
        assert rounds >= 2 and all(lo < hi for lo, hi in ranges)
        for b, info, (lo, hi) in zip(ranks, infos, ranges):
            assert info["total_pairs"] == winfo["total_pairs"]
            assert np.array_equal(b.thresholds(), whole.thresholds())
            got, want = b.pairs(lo, hi), whole.pairs(lo, hi)
            assert len(got[0]) == len(want[0]) > 0 and got[1] == want[1] and got[2] == want[2]
    finally:
        whole.close()
        for b in ranks:
            b.close()


def case_profile_from_reseq_archive(backend_cls, workdir):
    """the product reads the profile from `.reseq` + `.reseq.ipf` Boost text archives (rsq_profile_archive.cpp), the oracle from the
    RSQP container those were made from: pre-pass results, fragments and FASTQ text must not differ"""
    import archive_fixtures as af
    from reseq_amd import api
    lengths = [5000, 80, 3210]
    arrays = synth.make_profile(synth.TINY, seed=5, n_ref_seqs=len(lengths))
    stats = workdir / "tiny_e2e_archive.reseq"
    if not stats.exists():
        af.write_profile_archives(str(stats), arrays)
    bprof = str(stats)
    if backend_cls.name != "gpu":                  # the host emulation takes containers only: convert with the product's reader
        bprof = str(workdir / "tiny_e2e_archive.rsqp")
        prof = api.Profile(str(stats))
        prof.save(bprof)
        prof.close()
    p = Pair(backend_cls, workdir, "tiny_e2e", synth.TINY, lengths, seed=7, num_pairs=3000, backend_profile=bprof, edits=dict(error_multiplier=1.7))
    try:
        p.align_normalization()
        n_all, text = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert 2000 < n_all < 4000 and b"I" in text and b"D" in text
    finally:
        p.close()


def case_sieve_own_thresholds(backend_cls, workdir):
    """the other direction: the oracle takes the backend's own pre-pass results (tree-reduced sums)"""
    p = Pair(backend_cls, workdir, "tiny_e2e", synth.TINY, [5000, 80, 3210], seed=21, num_pairs=2500, base_identifier="Sim")
    try:
        p.osim.set_normalization(p.info["bias_normalization"], p.b.thresholds())
        n, text = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert n > 1500 and text.startswith(b"@Sim1_1:")
    finally:
        p.close()


def case_dense_coverage(backend_cls, workdir):
    """very high coverage: a quarter of all (start, length) cells carry fragments, duplicates are common and the sieve's
    candidate queue has to be flushed in the middle of a wave's positions"""
    p = Pair(backend_cls, workdir, "tiny_dense", synth.TINY, [3000], seed=17, num_pairs=60000)
    try:
        p.align_normalization()
        n, text = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert 50000 < n < 70000
    finally:
        p.close()


def case_adapter_only(backend_cls, workdir):
    p = Pair(backend_cls, workdir, "tiny_e2e", synth.TINY, [5000, 80, 3210], seed=3, num_pairs=30000)
    try:
        n = p.info["adapter_only_pairs"]
        assert n == p.osim.adapter_only_pairs() and n > 10
        o1, o2 = p.osim.adapter_only()
        b1, b2 = p.b.adapter_only_pairs(0, n)
        assert o1 == b1 and o2 == b2
        # split in two calls
        c1, c2 = p.b.adapter_only_pairs(0, 5)
        d1, d2 = p.b.adapter_only_pairs(5, n - 5)
        assert c1 + d1 == o1 and c2 + d2 == o2
    finally:
        p.close()


def case_p0_reads(backend_cls, workdir):
    """HiSeq-shaped profile (K = 40 qualities, 2x150): a 30 kb reference, about 1500 pairs"""
    p = Pair(backend_cls, workdir, "p0_small", synth.P0, [30000], seed=11, num_pairs=1500, prof_seed=103741084, ref_seed=2)
    try:
        p.align_normalization()
        n, text = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert 1000 < n < 2000
        recs = text.split(b"\n")
        assert all(len(s) == 150 for s in recs[1::4])
    finally:
        p.close()


def case_p0_tiles(backend_cls, workdir, n_tiles, num_pairs=1500, expect_image_tiles=1):
    """P0 with several tiles: their tables do not fit one 160 KiB image, so the read kernels serve one tile per workgroup (reads binned by the
    tile they draw); fragments, read ids (the tile is part of them) and FASTQ text equal the oracle's, for pairs, adapter-only pairs and
    seqToIllumina records"""
    cfg = synth.p0_with_tiles(n_tiles)
    tag = f"p0_tiles{n_tiles}"
    p = Pair(backend_cls, workdir, tag, cfg, [30000], seed=11, num_pairs=num_pairs, prof_seed=21, ref_seed=2)
    try:
        plan = p.b.fill_plan()
        assert plan["mask"] != 0, plan                          # screened draws, not the double-precision route
        if expect_image_tiles is not None:
            assert plan["image_tiles"] == expect_image_tiles, plan
        p.align_normalization()
        n, text = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert 0.6 * num_pairs < n < 1.4 * num_pairs
        tiles_seen = {l.split(b":")[4] for l in text.split(b"\n")[0::4] if l}
        assert len(tiles_seen) >= min(n_tiles, 3) and tiles_seen <= {b"%d" % t for t in cfg["tiles"]}
        # the same in two calls: the bins of a call depend on the call's pairs only
        half = 1 + p.info["total_blocks"] // 2
        f1, a1, a2 = p.b.pairs(1, half)
        f2, b1, b2 = p.b.pairs(half, p.info["total_blocks"] + 1)
        assert a1 + b1 == text
    finally:
        p.close()
    exp = _error_model(backend_cls, workdir, tag, cfg, 600, 150, seed=99, prof_seed=21, zero_frac=0.9)
    assert len({e[4] for e in exp}) >= min(n_tiles, 3)


def case_more_quality_values_than_the_screen_is_built_for(backend_cls, workdir):
    """60 quality values: above the 48 the screened single-precision draws are instantiated for, so the read kernels take the double-precision route
    (LogArrayResult::Draw as written, every row from device memory); the plan says so (rsq_last_warning after rsq_sim_create), the output is the oracle's"""
    cfg = dict(synth.TINY, name="TINYq60", qual_from=2, qual_to=62)
    p = Pair(backend_cls, workdir, "tiny_q60", cfg, [4000, 2500], seed=17, num_pairs=2500, prof_seed=9)
    try:
        plan = p.b.fill_plan()
        assert plan["mask"] == 0, plan
        p.align_normalization()
        n, text = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert n > 1500
        assert max(max(l) for l in text.split(b"\n")[3::4] if l) > 33 + 48          # qualities above the 48th value occur
    finally:
        p.close()
    exp = _error_model(backend_cls, workdir, "tiny_q60", cfg, 300, 30, seed=5, prof_seed=9, zero_frac=0.7)
    assert len(exp) == 300


def case_indel_columns_shuffled(backend_cls, workdir):
    """the indel draw decided by the random word alone (DevTable::sure_range) when "no indel" is a middle column of its tables and one
    insertion is frequent: the range has a lower and an upper end"""
    cfg = dict(synth.P0, name="P0s", indel_columns_shuffled=True)
    p = Pair(backend_cls, workdir, "p0_shuffled", cfg, [20000], seed=13, num_pairs=1000, prof_seed=7, ref_seed=3)
    try:
        arrays = synth.make_profile(cfg, seed=7)
        par0 = arrays["tab.indels.0.0.par0"].tolist()
        assert 0 < par0.index(0) < len(par0) - 1, par0
        p.align_normalization()
        n, text = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert 600 < n < 1400
        cigars = [l.split(b" ")[1] for l in text.split(b"\n")[0::4] if l]
        assert sum(b"I" in c for c in cigars) > 20
    finally:
        p.close()


def case_ragged_tables(backend_cls, workdir, cfg_base=None, lengths=(5000, 3210), num_pairs=3000):
    """Every quality, base-call and indel table over its own value ranges (rows cut off both ends, differently per table) and an empty table in two families: the read
    kernel's layout writes all tables of a family over the family's common ranges with edge rows repeated (FamilyGeo) -- the output must stay the oracle's, whose Draw
    clamps per table (AdjustIndeces)"""
    cfg = dict(cfg_base or synth.TINY, ragged_tables=True)
    cfg["name"] = cfg["name"] + "rag"
    arrays = synth.make_profile(cfg, seed=19)
    lims = {tuple(map(tuple, arrays[k].tolist())) for k in arrays if k.startswith("tab.quality.") and k.endswith(".limits")}
    assert len(lims) > 3, "the quality tables must differ in their ranges"
    p = Pair(backend_cls, workdir, cfg["name"].lower(), cfg, list(lengths), seed=29, num_pairs=num_pairs, prof_seed=19)
    try:
        assert p.b.fill_plan()["mask"] != 0
        p.align_normalization()
        n, text = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
        assert n > num_pairs // 2
    finally:
        p.close()
    exp = _error_model(backend_cls, workdir, cfg["name"].lower(), cfg, 400, cfg["read_len_max"], seed=5, prof_seed=19, zero_frac=0.7)
    assert len(exp) == 400


def case_packed_reference(backend_cls, workdir):
    """One load per host (rsq_sim_export_reference / rsq_sim_import_reference): a simulator that was given another simulator's packed reference -- variants of every
    kind on two alleles and methylation regions included -- instead of the files writes the same fragments and FASTQ bytes; a file packed for another profile,
    a damaged file and a second import are refused"""
    import pytest
    ppath, fpath, seqs = make_inputs(workdir, "packed", synth.TINY, [6200, 80, 3100])
    vcf = workdir / "packed.vcf"
    write_vcf(vcf, seqs, _mixed_variant_set(seqs, np.random.default_rng(5), 30, [999, 1000, 1999, 2000]))
    names = [n.split(" ")[0] for n, _ in seqs]
    for with_variants in (True, False):
        bed = workdir / f"packed_{int(with_variants)}.bed"                  # a rate per allele: two with the phased sample, one without variants
        second = (lambda r: f"\t{r}") if with_variants else (lambda r: "")
        bed.write_text(f"{names[0]}\t100\t900\t0.3{second(0.6)}\n{names[0]}\t2000\t5000\t0.0{second(0.5)}\n{names[2]}\t50\t2000\t0.0{second(1.0)}\n")
        a = backend_cls(ppath, fpath, 0, None, vcf_path=str(vcf)) if with_variants else backend_cls(ppath, fpath, 0)
        b = backend_cls(ppath, None, 0)
        try:
            a.read_methylation(bed)
            packed = workdir / f"packed_{int(with_variants)}.ref"
            a.export_reference(packed)
            b.import_reference(packed)
            ia, ib = a.prepare(7, 3000), b.prepare(7, 3000)
            assert ia == ib
            fa, a1, a2 = a.pairs(1, ia["total_blocks"] + 1)
            fb, b1, b2 = b.pairs(1, ib["total_blocks"] + 1)
            assert len(fa) > 2000 and fa.tobytes() == fb.tobytes() and a1 == b1 and a2 == b2
            if with_variants:
                assert b"_allele1" in a1
            with pytest.raises(Exception, match="has a reference already"):
                b.import_reference(packed)
        finally:
            a.close()
            b.close()
    other = workdir / "packed_other.rsqp"
    synth.write_profile(other, synth.make_profile(dict(synth.TINY, name="TINYr32", read_len_max=32), seed=5, n_ref_seqs=3))
    c = backend_cls(str(other), None, 0)
    try:
        with pytest.raises(Exception, match="packed for another profile"):
            c.import_reference(workdir / "packed_1.ref")
    finally:
        c.close()
    data = (workdir / "packed_1.ref").read_bytes()
    (workdir / "cut.ref").write_bytes(data[:len(data) // 2])
    d = backend_cls(ppath, None, 0)
    try:
        with pytest.raises(Exception, match="incomplete|damaged"):
            d.import_reference(workdir / "cut.ref")
    finally:
        d.close()
    # a file someone else may have written (the launcher's /dev/shm): every crafted record below is refused by name, none reaches an array index.
    # Layout: "RSQREF1\0", then {u32 tag, u32 element size, u64 count, bytes padded to 8} (rsq_pack.h refio)
    import struct
    records, at = [], 8
    while True:
        tag, size, count = struct.unpack_from("<IIQ", data, at)
        records.append((tag, size, count, at + 16))
        if tag == 20:                                                      # kEnd
            break
        at += 16 + (count * size + 7) // 8 * 8
    place = {tag: (size, count, body) for tag, size, count, body in records}

    def crafted(tag, index, value, fmt):
        size, count, body = place[tag]
        out = bytearray(data)
        struct.pack_into(fmt, out, body + (index % count) * size, value)
        return bytes(out)

    def header(tag, size=None, count=None):
        out = bytearray(data)
        old_size, old_count, body = place[tag]
        struct.pack_into("<IQ", out, body - 12, old_size if size is None else size, old_count if count is None else count)
        return bytes(out)
    tampered = {
        "count times size wraps": (header(7, size=1 << 31, count=(1 << 33) + 1), "damaged record"),              # kWords
        "element size 0": (header(7, size=0), "damaged record"),
        "name offsets decrease": (crafted(3, 1, 1 << 40, "<Q"), "sequence tables"),                                # kNamePtr
        "id offsets past the ids": (crafted(5, -1, 1 << 40, "<Q"), "sequence tables"),                             # kIdPtr
        "variant offsets past the variants": (crafted(10, -1, 1 << 30, "<I"), "variant tables"),                   # kVarPtr
        "variant offsets decrease": (crafted(10, 1, 0xFFFFFFF0, "<I"), "variant tables"),
        "allele map offsets past the map": (crafted(13, -1, 1 << 30, "<I"), "variant tables"),                     # kAlleleMapPtr
        "extra start offsets past the list": (crafted(15, -1, 1 << 30, "<I"), "variant tables"),                   # kExtraSeqPtr
        "a variant behind its sequence": (crafted(9, 0, 0x7FFFFFFF, "<I"), "a variant outside"),                   # kVariants[0].pos
        "a map entry of no variant": (crafted(12, 0, 0x7FFFFFFF, "<I"), "allele map entry"),                       # kAlleleMap[0].pos
        "methylation offsets past the regions": (crafted(16, -1, 1 << 30, "<I"), "methylation tables"),            # kMethPtr
        "a region behind its sequence": (crafted(18, 0, 0x7FFFFFFF, "<I"), "methylation region"),                  # kMethSecond
        "fewer conversion rates than regions": (header(19, count=place[19][1] - 1), "methylation tables|damaged|incomplete"),
    }
    for what, (blob, message) in tampered.items():
        (workdir / "tampered.ref").write_bytes(blob)
        d = backend_cls(ppath, None, 0)
        try:
            with pytest.raises(Exception, match=message):
                d.import_reference(workdir / "tampered.ref")
        finally:
            d.close()
    # the export neither follows a link left at its temporary name nor writes over a file there
    a = backend_cls(ppath, fpath, 0)
    try:
        victim = workdir / "victim.txt"
        victim.write_text("untouched")
        (workdir / "linked.ref.writing").symlink_to(victim)
        with pytest.raises(Exception, match="cannot write"):
            a.export_reference(workdir / "linked.ref")
        assert victim.read_text() == "untouched" and not (workdir / "linked.ref").exists()
        (workdir / "linked.ref.writing").unlink()
        a.export_reference(workdir / "linked.ref")
        assert ((workdir / "linked.ref").stat().st_mode & 0o777) == 0o600
    finally:
        a.close()


def case_profile_edits(backend_cls, workdir):
    for edits in ({"error_multiplier": 3.0}, {"no_substitutions": True}, {"no_indels": True}, {"no_substitutions": True, "no_indels": True}):
        p = Pair(backend_cls, workdir, "tiny_e2e", synth.TINY, [5000, 80, 3210], seed=5, num_pairs=800, edits=edits)
        try:
            p.align_normalization()
            n, text = _compare_blocks(p, 1, 4)
            assert n > 100
            if edits.get("no_indels"):
                assert all(b"D" not in l.split(b" ")[1] and b"I" not in l.split(b" ")[1] for l in text.split(b"\n")[0::4] if l)
            if edits.get("no_substitutions") and edits.get("no_indels"):
                assert all(l.endswith(b" E0") for l in text.split(b"\n")[0::4] if l)
        finally:
            p.close()


def _error_model(backend_cls, workdir, tag, cfg, n, read_len, seed, prof_seed, zero_frac):
    ppath, _, _ = make_inputs(workdir, tag, cfg, [100], prof_seed=prof_seed)
    arrays = synth.make_profile(cfg, seed=prof_seed)
    rec = synth.make_error_model_input(9, n, read_len, arrays, zero_frac=zero_frac)
    oprof = O.Profile(ppath)
    exp = O.error_model_only(oprof, seed, rec, first_index=17)
    b = backend_cls(ppath, None)
    b.prepare(seed)
    got = b.error_model(rec, first_index=17)
    assert len(got) == len(exp) == n
    for i, (e, g) in enumerate(zip(exp, got)):
        assert e == g, i
    if hasattr(b, "error_model_fastq"):                       # the text of the same records formatted on the device (rsq_sim_error_model_fastq)
        ids = [(b"read%d/" % i) + b"x" * (i % 41) + (b" extra words" if i % 5 == 0 else b"") for i in range(n)]
        want = b"".join(b"@" + ids[i] + b" " + e[2].encode() + b" E%d\n" % e[3] + bytes(b"ACGTN"[c] for c in e[0]) + b"\n+\n" + e[1] + b"\n"
                        for i, e in enumerate(exp))
        assert b.error_model_fastq(rec, ids, first_index=17) == want
    if hasattr(b, "error_model_fasta"):                       # the same through the FASTA text, parsed on the device (rsq_sim_error_model_fasta)
        _error_model_fasta(b, oprof, seed, rec)
    b.close()
    oprof.close()
    return exp


def fasta_of_records(rec, ids, wrap_every=0, line_end="\n"):
    """seqToIllumina's input for the records: ">{id} {1|2};{fragment length};{dominant errors};{error rates}" (Simulator.cpp:2423-2485), every
    wrap_every-th sequence wrapped over lines of 20"""
    lines = []
    for i, rid in enumerate(ids):
        seq = "".join("ACGT"[c] for c in rec["seqs"][i])
        dom = "".join("ACGTN"[c] for c in rec["dom"][i])
        rate = synth.encode_sys_rate(rec["rate"][i]).tobytes().decode()
        lines.append(f">{rid} {int(rec['seg'][i]) + 1};{int(rec['frag_len'][i])};{dom};{rate}")
        lines += [seq[k:k + 20] for k in range(0, len(seq), 20)] if wrap_every and i % wrap_every == 0 else [seq]
    return (line_end.join(lines) + line_end).encode()


def _error_model_fasta(b, oprof, seed, rec):
    rec = dict(rec)
    r = rec["rate"].astype(np.int64)                          # what survives the file: odd percents above 86 become the even one below (Simulator.cpp:2439-2442)
    rec["rate"] = np.where(r > 86, r - r % 2, r).astype(np.uint8)
    n = len(rec["seg"])
    ids = [f"read {i}/x" if i % 7 == 0 else f"r{i}" for i in range(n)]
    exp = O.error_model_only(oprof, seed, rec, first_index=17)
    want = b"".join(b"@" + ids[i].encode() + b" " + e[2].encode() + b" E%d\n" % e[3] + bytes(b"ACGTN"[c] for c in e[0]) + b"\n+\n" + e[1] + b"\n" for i, e in enumerate(exp))
    text = fasta_of_records(rec, ids, wrap_every=3)
    got, k, used = b.error_model_fasta(text, first_index=17)
    assert (k, used) == (n, len(text))
    assert got == want
    assert b.error_model_fasta(fasta_of_records(rec, ids, wrap_every=2, line_end="\r\n"), first_index=17)[0] == want
    for skew in (1, 7, 9):                                   # text that does not begin on a 16-byte boundary of device memory (the record kernel stages whole 16-byte words)
        assert b.error_model_fasta(text, first_index=17, skew=skew)[0] == want, skew
    # in blocks that end anywhere: what the call leaves over goes in front of the next block
    for block in (len(text) // 3 + 1, 4096):
        out, first, rest, pos = [], 17, b"", 0
        while True:
            piece = rest + text[pos:pos + block]
            pos += block
            final = pos >= len(text)
            got, k, used = b.error_model_fasta(piece, first_index=first, final=final)
            out.append(got)
            first += k
            rest = piece[used:]
            if final:
                break
        assert first == 17 + n and not rest and b"".join(out) == want, block


def case_error_model_tiny(backend_cls, workdir):
    exp = _error_model(backend_cls, workdir, "em_tiny", synth.TINY, 400, 30, seed=13, prof_seed=5, zero_frac=0.7)
    assert len({e[4] for e in exp}) == 3                      # all three tiles drawn
    assert any("D" in e[2] for e in exp) and any("I" in e[2] for e in exp)


def case_error_model_templates_beyond_the_staging(backend_cls, workdir):
    """records of 1.2 KB and 3 KB: the 256 records of a workgroup of k_fasta_records no longer fit the 144 KB of LDS their text is staged in, the lanes read HBM"""
    _error_model(backend_cls, workdir, "em_tiny", synth.TINY, 700, 400, seed=3, prof_seed=5, zero_frac=0.5)
    _error_model(backend_cls, workdir, "em_tiny", synth.TINY, 300, 1000, seed=4, prof_seed=5, zero_frac=0.8)


def case_error_model_long_templates(backend_cls, workdir):
    # templates longer than the profile's read length (the tail of the template is never reached)
    _error_model(backend_cls, workdir, "em_tiny", synth.TINY, 64, 75, seed=2, prof_seed=5, zero_frac=0.5)


def case_error_model_p0(backend_cls, workdir):
    exp = _error_model(backend_cls, workdir, "em_p0", synth.P0, 300, 150, seed=99, prof_seed=103741084, zero_frac=0.97)
    assert all(len(e[0]) == 150 for e in exp)


def case_error_model_p0_groups(backend_cls, workdir, n=7000):
    """k_fill_records reads a record's template, dominant errors and rates in 8-byte groups held between steps: templates of 149, 150 and 151 bases (the
    last group partly filled; the template one base short of / beyond the profile's read length) with a systematic error rate at 40 % of the positions,
    group borders included; 3 x n records"""
    for read_len in (149, 150, 151):
        exp = _error_model(backend_cls, workdir, "em_p0", synth.P0, n, read_len, seed=100 + read_len, prof_seed=103741084, zero_frac=0.6)
        assert sum(e[3] > 0 for e in exp) > n // 20                 # reads with errors
        assert all(len(e[0]) == 150 for e in exp)


def case_sys_error_profile_round_trip(backend_cls, workdir):
    """--writeSysError / --readSysError (Simulator.cpp:2562-2653, Simulator.h:326-335): the profile text equals the oracle's, and a
    simulation that reads it back produces the oracle's reads from the same file"""
    p = Pair(backend_cls, workdir, "sysprof", synth.TINY, [4100, 2999], seed=19, num_pairs=2500)
    try:
        path = workdir / f"sysprof_{backend_cls.name}.fq"
        p.b.create_sys_error_profile(31, path)
        text = path.read_bytes()
        assert text == O.create_sys_error_profile(p.oprof, p.oref, 31)
        recs = text.split(b"\n")
        assert recs[0].endswith(b" reverse") and recs[4].endswith(b" forward") and len(recs) == 4 * 4 + 1
        assert set(recs[1]) <= set(b"ACGTN") and len(recs[1]) == len(recs[3]) == 4100
        # drawn with another seed than the simulation: reading it back must change the tracks, identically on both sides
        p.info = p.b.prepare(19, 2500)
        before = p.b.sys_errors(0, 0, 4100)
        p.b.read_sys_errors(path)
        p.osim.load_sys_errors(text)
        for seq, n in ((0, 4100), (1, 2999)):
            for strand in (0, 1):
                od, orate = p.osim.sys_errors(strand, seq)
                bd, brate = p.b.sys_errors(strand, seq, n)
                assert np.array_equal(od, bd) and np.array_equal(orate, brate)
        assert not np.array_equal(before[1], p.b.sys_errors(0, 0, 4100)[1])
        p.align_normalization()
        assert _compare_blocks(p, 1, p.info["total_blocks"] + 1)[0] > 1500
    finally:
        p.close()


def case_sys_error_profile_rejects_wrong_reference(backend_cls, workdir):
    """LoadSysErrorRecord's length check (Simulator.cpp:762-766); a sequence without a unit consumes no record, so a profile
    written for a reference with a too-short sequence in front is rejected, as in the reference"""
    import pytest
    p = Pair(backend_cls, workdir, "sysprof_bad", synth.TINY, [80, 4100], seed=19, num_pairs=500)
    try:
        path = workdir / f"sysprof_bad_{backend_cls.name}.fq"
        p.b.create_sys_error_profile(31, path)
        p.b.prepare(19, 500)
        with pytest.raises(Exception, match="does not match reference sequence"):
            p.b.read_sys_errors(path)
        with pytest.raises(RuntimeError, match="does not match reference sequence"):
            p.osim.load_sys_errors(path.read_bytes())
    finally:
        p.close()


def case_ref_bias_modes(backend_cls, workdir):
    """--refBias no|draw|file (UpdateRefSeqBias, FragmentDistributionStats.cpp:3352-3500)"""
    lengths = [3000, 2600, 2800]
    bias_file = workdir / "bias.tsv"
    ppath, fpath, seqs = make_inputs(workdir, "refbias", synth.TINY, lengths)
    names = [n.split(" ")[0] for n, _ in seqs]
    bias_file.write_text(f">{names[1]} some description\t0.5\n{names[0]}\t2.25\n{names[2]} 1.0\n\n")
    for mode, kw in ((1, {}), (2, {}), (3, {"ref_bias_file": bias_file})):
        p = Pair(backend_cls, workdir, "refbias", synth.TINY, lengths, seed=23, num_pairs=1500, ref_bias_mode=mode, **kw)
        try:
            ob, bb = p.osim.ref_seq_bias(), p.b.ref_seq_bias(3)
            assert np.array_equal(ob, bb), (mode, ob, bb)
            if mode == 1:
                assert list(bb) == [1.0, 1.0, 1.0]
            if mode == 3:
                assert list(bb) == [2.25, 0.5, 1.0]
            assert abs(p.info["bias_normalization"] / p.osim.bias_normalization() - 1) < NORM_RTOL
            p.align_normalization()
            assert _compare_blocks(p, 1, p.info["total_blocks"] + 1)[0] > 800
        finally:
            p.close()
    import pytest
    bias_file.write_text(f"{names[0]}\t2.25\n")
    with pytest.raises(Exception, match="[Cc]ould not find bias|errors"):
        Pair(backend_cls, workdir, "refbias", synth.TINY, lengths, seed=23, num_pairs=1500, ref_bias_mode=3, ref_bias_file=bias_file)


def case_methylation(backend_cls, workdir):
    """--methylation without variants (CTConversion, Simulator.cpp:1925-2002,2219-2247; ReadMethylation, Reference.cpp:1177-1310):
    C->T inside the listed regions with probability 1 - methylation, shared by the duplicates of a site; the reverse walk never
    reaches region 0 and read_pos is 16 bits wide, both as in the reference"""
    lengths = [70000, 90, 4000]
    p = Pair(backend_cls, workdir, "meth", synth.TINY, lengths, seed=29, num_pairs=60000)
    try:
        names = [n.split(" ")[0] for n, _ in p.seqs]
        bed = workdir / "meth.bed"
        bed.write_text("track name=test\n\n"
                       f"{names[0]}\t0\t400\t0.0\n{names[0]}\t420\t421\t0.5\n{names[0]} 900 1500 0.25\n\n{names[0]}\t66500\t69000\t0.0\n"
                       f"{names[2]}\t100\t3900\t1.0\n")
        p.b.read_methylation(bed)
        p.osim.read_methylation(bed)
        p.align_normalization()
        tb = p.info["total_blocks"]
        n_a, text_a = _compare_blocks(p, 1, 3)                 # regions 0..2: fully unmethylated start, a single base, a quarter methylated
        n_b, _ = _compare_blocks(p, 66, 71)                    # 65 kb after the previous region: read_pos wraps in the forward walk
        n_c, text_c = _compare_blocks(p, 71, tb + 1)           # sequence 2: methylation 1.0, nothing converts
        assert n_a > 500 and n_b > 500 and n_c > 500
        # forward-strand first mates that start inside [0, 400) are fully converted there: no C survives in the template part
        plain = Pair(backend_cls, workdir, "meth", synth.TINY, lengths, seed=29, num_pairs=60000)
        try:
            plain.align_normalization()
            _, text_plain_a = _compare_blocks(plain, 1, 3)
            _, text_plain_c = _compare_blocks(plain, 71, tb + 1)
        finally:
            plain.close()
        assert text_c == text_plain_c                          # methylation 1.0 == no file
        assert text_a != text_plain_a and text_a.count(b"C") < text_plain_a.count(b"C")
        assert len(text_a.split(b"\n")) == len(text_plain_a.split(b"\n"))
    finally:
        p.close()
    import pytest
    for bad, msg in ((f"{names[0]}\t10\t5\t0.5\n", "Third field is smaller"), (f"{names[0]}\t10\t20\t0.5\n{names[0]}\t15\t30\t0.5\n", "overlapping"),
                     (f"{names[0]}\t10\t20\t1.5\n", "not between 0 and 1"), (f"{names[0]}\t10\t20\n", "alleles specified"),
                     (f"{names[0]}\t10\t70001\t0.5\n", "larger than sequence length")):
        bed.write_text(bad)
        q = Pair(backend_cls, workdir, "meth", synth.TINY, lengths, seed=29, num_pairs=100)
        try:
            with pytest.raises(Exception, match=msg):
                q.b.read_methylation(bed)
            with pytest.raises(RuntimeError, match=msg):
                q.osim.read_methylation(bed)
        finally:
            q.close()


# ------------------------------------------------------------------------------------------------ variants
def write_vcf(path, seqs, variants, samples=1):
    """variants: [(sequence index, 0-based position, alt letters, genotype string like "0|1")]; REF is read from seqs"""
    lines = ["##fileformat=VCFv4.2"] + [f"##contig=<ID={n.split(' ')[0]},length={len(c)}>" for n, c in seqs]
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i + 1}" for i in range(samples)))
    for si, pos, ref_len, alt, gt in variants:
        name = seqs[si][0].split(" ")[0]
        ref = "".join("ACGT"[b] for b in seqs[si][1][pos:pos + ref_len])
        lines.append(f"{name}\t{pos + 1}\t.\t{ref}\t{alt}\t.\tPASS\t.\tGT\t{gt}")
    path.write_text("\n".join(lines) + "\n")


def _substitution_set(seqs, rng, density, special):
    """random substitutions plus hand-placed ones: neighbours, block borders, sequence ends, two alts at one position"""
    out = []
    for si, (_, codes) in enumerate(seqs):
        L = len(codes)
        if L < 200:
            continue
        pos = set(int(x) for x in rng.choice(L, size=L // density, replace=False))
        pos |= {x for x in special if x < L} | {L - 1 - x for x in (0, 1, 2, 9, 10, 19, 20, 29)}
        for p0 in sorted(pos):
            alt = "ACGT"[(int(codes[p0]) + 1 + int(rng.integers(0, 3))) % 4]
            out.append((si, p0, 1, alt, ["0|1", "1|0", "1|1"][int(rng.integers(0, 3))]))
    return out


def _compare_variant_sys_errors(p, seqs):
    for seq in seqs:
        for i in range(p.ovars.contents.n[seq]):
            for strand in (0, 1):
                d, r = p.osim.var_sys_errors(strand, seq, i)
                got = p.b.variant_sys_errors(seq, i, strand)
                assert len(got) == len(d) and np.array_equal(got, d.astype(np.uint16) | (r.astype(np.uint16) << 8)), (seq, i, strand)


def _compare_blocks_var(p, lo, hi):
    ofr = p.osim.sieve_var(lo, hi)
    o1, o2 = p.osim.create_reads_var(ofr)
    bfr, b1, b2 = p.b.pairs(lo, hi)
    assert len(ofr) == len(bfr)
    for name in ("seq", "start", "len", "dup", "strand", "block", "number"):
        assert np.array_equal(ofr[name], bfr[name]), name
    assert np.array_equal(ofr["allele"], bfr["allele"])
    assert o1 == b1
    assert o2 == b2
    return ofr, o1


def case_variants_substitutions(backend_cls, workdir):
    """-V with substitutions (Simulator.cpp:2249-2357 with VariantsLoaded): per-allele sieve (possible alleles, ChooseAlleles, GC and
    surroundings of the allele, counts with NumAlleles), templates of the allele, the variants' own systematic errors
    (SetSystematicErrorVariantsForward / Reverse) spliced by GetSysErrorFromBlock, `_allele<a>` read ids.  The oracle runs the
    reference's general bookkeeping (oracle_variants.hpp); the product reads per-allele copies of the reference."""
    lengths = [6200, 80, 3100]
    rng = np.random.default_rng(17)
    seqs = make_inputs(workdir, "vars", synth.TINY, lengths)[2]
    special = [0, 1, 2, 5, 9, 10, 11, 19, 20, 21, 29, 30, 31, 995, 996, 997, 998, 999, 1000, 1001, 1002, 1003, 1999, 2000, 2001, 2002, 2004, 2999, 3000]
    subs = _substitution_set(seqs, rng, 40, special)
    # a second alternative at positions that already carry one, on the other allele ("1|2")
    subs_two = []
    for si, p0, rl, alt, gt in subs:
        if p0 % 7 == 3:
            other = next(c for c in "ACGT" if c != alt and c != "ACGT"[seqs[si][1][p0]])
            subs_two.append((si, p0, rl, f"{alt},{other}", "1|2"))
        else:
            subs_two.append((si, p0, rl, alt, gt))
    vcf = workdir / "subs.vcf"
    write_vcf(vcf, seqs, subs_two)
    p = Pair(backend_cls, workdir, "vars", synth.TINY, lengths, seed=23, num_pairs=9000, vcf=vcf)
    try:
        assert p.info["total_blocks"] == p.osim.total_blocks()
        np.testing.assert_allclose(p.b.thresholds(), p.osim.thresholds(), rtol=NORM_RTOL, atol=0)      # thresholds for two alleles
        if hasattr(p.b, "variant_sys_errors"):
            _compare_variant_sys_errors(p, (0, 2))
        p.align_normalization()
        tb = p.info["total_blocks"]
        ofr, text = _compare_blocks_var(p, 1, tb + 1)
        assert len(ofr) > 7000 and set(np.unique(ofr["allele"])) == {0, 1}
        assert b"_allele0:" in text and b"_allele1:" in text
        part, _ = _compare_blocks_var(p, 2, 4)                 # batching by block range
        assert len(part) and np.array_equal(part["start"], ofr["start"][(ofr["block"] >= 2) & (ofr["block"] < 4)])
        # the variants change the output: the same run without them differs
        plain = Pair(backend_cls, workdir, "vars", synth.TINY, lengths, seed=23, num_pairs=9000)
        try:
            plain.align_normalization()
            assert len(plain.osim.sieve(1, tb + 1)) != len(ofr) or plain.osim.create_reads(plain.osim.sieve(1, tb + 1))[0] != text
        finally:
            plain.close()
    finally:
        p.close()


def _mixed_variant_set(seqs, rng, density, special=(), ends=0):
    """substitutions, insertions (1..6 bases, with or without a substituted first base) and deletions (1..4 bases), non-overlapping;
    ends > 0: every third position of the first and the last `ends` positions gets one as well (surroundings that wrap around)"""
    out = []
    for si, (_, codes) in enumerate(seqs):
        L = len(codes)
        if L < 200:
            continue
        last = -1
        cand = set(int(x) for x in rng.choice(np.arange(0, L - 8), size=L // density, replace=False)) | {x for x in special if x < L - 8}
        if ends:
            cand |= {int(x) for x in list(range(0, ends)) + list(range(L - 8 - ends, L - 8)) if rng.random() < 0.34}
        for p0 in sorted(cand):
            if p0 <= last:
                continue
            kind = int(rng.integers(0, 4))
            gt = ["0|1", "1|0", "1|1"][int(rng.integers(0, 3))]
            ref = "ACGT"[codes[p0]]
            other = "ACGT"[(int(codes[p0]) + 1 + int(rng.integers(0, 3))) % 4]
            ins = "".join("ACGT"[b] for b in rng.integers(0, 4, int(rng.integers(1, 7))))
            if kind == 0:
                out.append((si, p0, 1, other, gt))
                last = p0
            elif kind == 1:
                out.append((si, p0, 1, ref + ins, gt))
                last = p0
            elif kind == 2:
                out.append((si, p0, 1, other + ins, gt))          # substitution and insertion in one variant
                last = p0
            else:
                dl = int(rng.integers(1, 5))
                out.append((si, p0, dl + 1, ref, gt))
                last = p0 + dl
    return out


def case_variants_indels(backend_cls, workdir, density=18, seed=31, tag="indels", lengths=(6200, 80, 3100), samples=1, ends=0):
    """-V with insertions and deletions: starts inside inserted bases as extra slots of the sieve, every cell read off the allele's
    coordinate map (rsq_variants.h) against the oracle's incremental restatement of the reference's bookkeeping, templates with the
    allele's variants, the error walk through insertions and deletions, end positions shifted by the allele's length changes in the
    read ids.  ends > 0: variants crowd the first and the last positions of every sequence (surroundings that wrap around the ends)"""
    lengths = list(lengths)
    rng = np.random.default_rng(seed)
    seqs = make_inputs(workdir, tag, synth.TINY, lengths)[2]
    special = [0, 3, 9, 10, 11, 20, 29, 30, 990, 995, 998, 999, 1000, 1001, 1990, 1999, 2000, 2995, 2999, 3000]
    vs = _mixed_variant_set(seqs, rng, density, special, ends=ends)
    if samples > 1:
        vs = [(si, p0, rl, alt, "\t".join(["0|1", "1|0", "1|1", "0|0"][int(rng.integers(0, 4))] for _ in range(samples - 1)) + "\t" + gt) for si, p0, rl, alt, gt in vs]
    vcf = workdir / f"{tag}.vcf"
    write_vcf(vcf, seqs, vs, samples=samples)
    p = Pair(backend_cls, workdir, tag, synth.TINY, lengths, seed=seed, num_pairs=9000, vcf=vcf)
    try:
        np.testing.assert_allclose(p.b.thresholds(), p.osim.thresholds(), rtol=NORM_RTOL, atol=0)
        if hasattr(p.b, "variant_sys_errors"):
            _compare_variant_sys_errors(p, [i for i, n in enumerate(lengths) if n >= 200])
        p.align_normalization()
        tb = p.info["total_blocks"]
        ofr, text = _compare_blocks_var(p, 1, tb + 1)
        assert len(ofr) > 7000
        assert (ofr["sub"] > 0).sum() > 20                    # fragments that start inside inserted bases
        shift = ofr["end"].astype(np.int64) - ofr["start"] - ofr["len"]
        assert shift.min() < 0 < shift.max()                  # insertions shorten, deletions lengthen the reference span
        part, _ = _compare_blocks_var(p, 2, 5)                # batching by block range
        assert len(part) == int(((ofr["block"] >= 2) & (ofr["block"] < 5)).sum())
    finally:
        p.close()


def _complex_variant_set(seqs, rng, density):
    """long insertions (20..60 bases) and deletions (10..40), records with two alternatives (insertion | substitution, deletion |
    insertion), neighbouring variants, variants at the first positions"""
    out = []
    ins = lambda n: "".join("ACGT"[b] for b in rng.integers(0, 4, n))
    for si, (_, codes) in enumerate(seqs):
        L = len(codes)
        if L < 200:
            continue
        last = -1
        for p0 in sorted(set(int(x) for x in rng.choice(np.arange(0, L - 60), size=L // density, replace=False)) | {0, 1, L - 61}):
            if p0 <= last:
                continue
            ref = "ACGT"[codes[p0]]
            kind = int(rng.integers(0, 6))
            if kind == 0:
                out.append((si, p0, 1, ref + ins(int(rng.integers(20, 61))), "0|1"))
                last = p0
            elif kind == 1:
                dl = int(rng.integers(10, 41))
                out.append((si, p0, dl + 1, ref, "1|0"))
                last = p0 + dl
            elif kind == 2:
                out.append((si, p0, 1, f"{ref + ins(3)},{'ACGT'[(codes[p0] + 1) % 4]}", "1|2"))
                last = p0
            elif kind == 3:
                dl = int(rng.integers(1, 4))
                refs = "".join("ACGT"[codes[p0 + k]] for k in range(dl + 1))
                out.append((si, p0, dl + 1, f"{ref},{refs + ins(2)}", "1|2"))
                last = p0 + dl
            elif kind == 4:
                out.append((si, p0, 1, "ACGT"[(codes[p0] + 2) % 4], "1|1"))
                out.append((si, p0 + 1, 1, "ACGT"[codes[p0 + 1]] + ins(2), "0|1"))
                last = p0 + 1
            else:
                out.append((si, p0, 2, ref, "1|1"))
                last = p0 + 1
    return out


def case_variants_complex(backend_cls, workdir):
    """insertions longer than a surrounding and than a read, long deletions, two alternatives in one record, neighbouring variants"""
    lengths = [5300, 2400]
    rng = np.random.default_rng(401)
    seqs = make_inputs(workdir, "vcomplex", synth.TINY, lengths, ref_seed=91)[2]
    vcf = workdir / "vcomplex.vcf"
    write_vcf(vcf, seqs, _complex_variant_set(seqs, rng, 25))
    p = Pair(backend_cls, workdir, "vcomplex", synth.TINY, lengths, seed=2, num_pairs=7000, vcf=vcf, ref_seed=91)
    try:
        if hasattr(p.b, "variant_sys_errors"):
            _compare_variant_sys_errors(p, [0, 1])
        p.align_normalization()
        ofr, _ = _compare_blocks_var(p, 1, p.info["total_blocks"] + 1)
        assert len(ofr) > 6000 and ofr["sub"].max() >= 40
    finally:
        p.close()


def case_variants_walk_off_sequence(backend_cls, workdir):
    """Variants every few bases right at a sequence end: the walk of GetSysErrorFromBlock (skipped and late variants included) uses up more
    reference positions than the template has and leaves the sequence -- the reference follows a NULL next_block_ there.  The oracle raises,
    and so does the product, instead of reading whatever lies behind the track."""
    import pytest
    lengths = [2705, 4757]
    rng = np.random.default_rng(1011)
    # the draws tools/stress_variants.py makes before this variant set (its trial 11)
    n_seq = int(rng.integers(1, 5))
    [int(rng.integers(1001, 7000)) if rng.random() < 0.8 else int(rng.integers(40, 90)) for _ in range(n_seq)]
    int(rng.choice([6, 12, 25, 60])), int(rng.choice([1, 1, 2, 3])), int(rng.integers(0, 3))
    seqs = make_inputs(workdir, "walkoff", synth.TINY, lengths, ref_seed=511)[2]
    vcf = workdir / "walkoff.vcf"
    write_vcf(vcf, seqs, _mixed_variant_set(seqs, rng, 6))
    p = Pair(backend_cls, workdir, "walkoff", synth.TINY, lengths, seed=1, num_pairs=9000, vcf=vcf, ref_seed=511)
    try:
        p.align_normalization()
        tb = p.info["total_blocks"]
        walks_off = []
        for b in range(1, tb + 1):
            try:
                p.osim.create_reads_var(p.osim.sieve_var(b, b + 1))
            except RuntimeError as e:
                assert "systematic-error walk left the sequence" in str(e)
                walks_off.append(b)
        assert walks_off and len(walks_off) < tb, walks_off
        for b in walks_off:
            with pytest.raises(Exception, match="systematic-error walk left the sequence"):
                p.b.pairs(b, b + 1)
        fine = [b for b in range(1, tb + 1) if b not in walks_off]
        _compare_blocks_var(p, fine[0], fine[0] + 1)           # away from the sequence end the same set simulates fine
    finally:
        p.close()


def case_coverage_driven(backend_cls, workdir):
    """the number of pairs from --coverage (configs 3-4 are coverage driven) and from the profile's corrected coverage
    (Simulator.cpp:2713-2743, CoveragePropLostFromAdapters :90-104, CoverageToNumberPairs :106-108)"""
    lengths = [5000, 80, 3210]
    for coverage in (12.5, 0.0):                          # 0 = keep the coverage of the original data (TINY: 8x)
        p = Pair(backend_cls, workdir, "tiny_e2e", synth.TINY, lengths, seed=41, num_pairs=0, coverage=coverage)
        try:
            assert p.info["total_pairs"] == p.osim.total_pairs() and p.info["adapter_only_pairs"] == p.osim.adapter_only_pairs()
            want = (coverage or synth.TINY["corrected_coverage"]) * sum(lengths) / 2 / 29.0
            assert 0.9 * want < p.info["total_pairs"] + p.info["adapter_only_pairs"] < 1.25 * want
            p.align_normalization()
            n, _ = _compare_blocks(p, 1, p.info["total_blocks"] + 1)
            assert 0.85 * p.info["total_pairs"] < n < 1.15 * p.info["total_pairs"]
        finally:
            p.close()


def case_p0_variants(backend_cls, workdir, kind):
    """configs[4] in small: the HiSeq-shaped profile (K = 40 qualities, 2x150, inserts up to 1000) on a 30 kb reference with variants on
    two alleles -- `subs`: substitutions (allele copies of the reference), `indels`: insertions and deletions (per-cell bookkeeping,
    templates with variants), `meth`: indels together with --methylation (one conversion rate per allele)"""
    lengths = [30000]
    tag = f"p0_var_{kind}"
    rng = np.random.default_rng({"subs": 211, "indels": 223, "meth": 227}[kind])
    kw = dict(prof_seed=103741084, ref_seed=2)
    seqs = make_inputs(workdir, tag, synth.P0, lengths, **kw)[2]
    special = [0, 9, 10, 29, 30, 149, 150, 999, 1000, 1001, 14999, 15000]
    vs = _substitution_set(seqs, rng, 70, special) if kind == "subs" else _mixed_variant_set(seqs, rng, 70, special)
    vcf = workdir / f"{tag}.vcf"
    write_vcf(vcf, seqs, vs)
    p = Pair(backend_cls, workdir, tag, synth.P0, lengths, seed=53, num_pairs=1500, vcf=vcf, **kw)
    try:
        np.testing.assert_allclose(p.b.thresholds(), p.osim.thresholds(), rtol=NORM_RTOL, atol=0)
        if hasattr(p.b, "variant_sys_errors"):
            _compare_variant_sys_errors(p, [0])
        if kind == "meth":
            name = seqs[0][0].split(" ")[0]
            bed = workdir / f"{tag}.bed"
            bed.write_text(f"{name}\t0\t400\t0.0\t0.5\n{name}\t420\t421\t0.5\t0.0\n{name}\t900\t14000\t0.25\t0.75\n{name}\t15000\t29900\t0.1\t0.9\n")
            p.b.read_methylation(bed)
            p.osim.read_methylation(bed)
        p.align_normalization()
        tb = p.info["total_blocks"]
        ofr, text = _compare_blocks_var(p, 1, tb + 1)
        assert 1000 < len(ofr) < 2000 and set(np.unique(ofr["allele"])) == {0, 1}
        lens = {len(l) for l in text.split(b"\n")[1::4] if l}
        assert lens == {150}
        if kind != "subs":
            shift = ofr["end"].astype(np.int64) - ofr["start"] - ofr["len"]
            assert shift.min() < 0 < shift.max()
        part, _ = _compare_blocks_var(p, 7, 19)               # batching by block range
        assert len(part) == int(((ofr["block"] >= 7) & (ofr["block"] < 19)).sum())
    finally:
        p.close()


def case_variants_with_loaded_sys_errors(backend_cls, workdir):
    """--readSysError together with -V: the variants' own errors are drawn against the LOADED tracks (their error-region state follows
    the file's rates: SetSystematicErrorVariants* run after ReadSystematicErrors, Simulator.cpp:983-986,1232-1234)"""
    lengths = [5200, 3100]
    rng = np.random.default_rng(71)
    seqs = make_inputs(workdir, "vsys", synth.TINY, lengths)[2]
    vcf = workdir / "vsys.vcf"
    write_vcf(vcf, seqs, _mixed_variant_set(seqs, rng, 20, [0, 999, 1000]))
    donor = Pair(backend_cls, workdir, "vsys", synth.TINY, lengths, seed=5, num_pairs=100)      # a profile drawn with another seed
    prof = workdir / "vsys_profile.fq"
    try:
        donor.b.create_sys_error_profile(99, prof)
    finally:
        donor.close()
    p = Pair(backend_cls, workdir, "vsys", synth.TINY, lengths, seed=73, num_pairs=6000, vcf=vcf)
    try:
        before = [p.osim.var_sys_errors(0, 0, i) for i in range(p.ovars.contents.n[0])]
        p.b.read_sys_errors(prof)
        p.osim.load_sys_errors(prof.read_bytes())
        after = [p.osim.var_sys_errors(0, 0, i) for i in range(p.ovars.contents.n[0])]
        assert any(not np.array_equal(a[1], b[1]) or not np.array_equal(a[0], b[0]) for a, b in zip(before, after))
        if hasattr(p.b, "variant_sys_errors"):
            _compare_variant_sys_errors(p, [0, 1])
        p.align_normalization()
        ofr, _ = _compare_blocks_var(p, 1, p.info["total_blocks"] + 1)
        assert len(ofr) > 4000
    finally:
        p.close()


def case_variants_with_methylation(backend_cls, workdir):
    """--methylation together with -V: CTConversion with variants (Simulator.cpp:2004-2217) on the templates of the allele, with one
    conversion rate per allele (two columns) or one for all (one column); substitutions, insertions and deletions inside and between
    the regions; and a substitution-only set, which takes the same path once methylation is loaded"""
    lengths = [6200, 80, 3100]
    for tag, maker, seed in (("vmeth", _mixed_variant_set, 61), ("vmeth_subs", None, 67)):
        rng = np.random.default_rng(seed)
        seqs = make_inputs(workdir, tag, synth.TINY, lengths)[2]
        vs = maker(seqs, rng, 20, [0, 5, 399, 400, 401, 419, 420, 421, 899, 900, 1499, 1500]) if maker else _substitution_set(seqs, rng, 25, [0, 399, 400, 420, 899, 1500])
        vcf = workdir / f"{tag}.vcf"
        write_vcf(vcf, seqs, vs)
        names = [n.split(" ")[0] for n, _ in seqs]
        bed = workdir / f"{tag}.bed"
        bed.write_text(f"{names[0]}\t0\t400\t0.0\t0.5\n{names[0]}\t420\t421\t0.5\t0.0\n{names[0]}\t900\t1500\t0.25\t0.75\n{names[0]}\t2000\t6100\t0.1\t0.9\n"
                       f"{names[2]}\t100\t3000\t0.3\n")
        p = Pair(backend_cls, workdir, tag, synth.TINY, lengths, seed=seed, num_pairs=9000, vcf=vcf)
        try:
            p.b.read_methylation(bed)
            p.osim.read_methylation(bed)
            p.align_normalization()
            tb = p.info["total_blocks"]
            ofr, text = _compare_blocks_var(p, 1, tb + 1)
            assert len(ofr) > 7000
            plain = Pair(backend_cls, workdir, tag, synth.TINY, lengths, seed=seed, num_pairs=9000, vcf=vcf)
            try:
                plain.align_normalization()
                _, text_plain = _compare_blocks_var(plain, 1, tb + 1)
                assert text != text_plain and text.count(b"C") < text_plain.count(b"C")
            finally:
                plain.close()
        finally:
            p.close()


def case_variants_methylation_far_regions(backend_cls, workdir, seeds=(73, 79)):
    """CTConversion with variants where the reference's walk is more than geometry: regions more than 65535 bases apart (the 16-bit
    template position wraps and the walk goes on), variants of both kinds inside, at the borders of and between the regions, records with two
    alternatives at one position (the variant cursor stays behind on the second one), deletions inside regions, fragments that start inside inserted bases"""
    lengths = [140000, 2300]
    for seed in seeds:
        tag = f"vmeth_far{seed}"
        rng = np.random.default_rng(seed)
        seqs = make_inputs(workdir, tag, synth.TINY, lengths, ref_seed=seed)[2]
        vs = sorted(set((si, p0, rl, alt, gt) for si, p0, rl, alt, gt in _complex_variant_set(seqs, rng, 40) + _mixed_variant_set(seqs, rng, 35, [99, 100, 101, 699, 700, 66100, 66101, 131900])),
                    key=lambda v: (v[0], v[1]))
        kept, last = [], (-1, -1)
        for v in vs:                                              # the two generators' records must not overlap; none near a sequence end (the reference's error walk leaves the sequence there)
            if (v[0], v[1]) > last and 60 <= v[1] < lengths[v[0]] - 400:
                kept.append(v)
                last = (v[0], v[1] + v[2] + 1)
        vcf = workdir / f"{tag}.vcf"
        write_vcf(vcf, seqs, kept)
        names = [n.split(" ")[0] for n, _ in seqs]
        bed = workdir / f"{tag}.bed"
        bed.write_text(f"{names[0]}\t100\t700\t0.1\t0.6\n{names[0]}\t66100\t66800\t0.0\t0.3\n{names[0]}\t66830\t66831\t0.2\t0.2\n{names[0]}\t131900\t132600\t0.4\t0.0\n"
                       f"{names[1]}\t5\t2290\t0.25\n")
        p = Pair(backend_cls, workdir, tag, synth.TINY, lengths, seed=seed, num_pairs=200000, vcf=vcf, ref_seed=seed)
        try:
            p.b.read_methylation(bed)
            p.osim.read_methylation(bed)
            p.align_normalization()
            tb = p.info["total_blocks"]
            n = 0
            for lo, hi in ((1, 3), (66, 69), (131, 135), (141, tb + 1)):      # around the regions, and the second sequence
                ofr, _ = _compare_blocks_var(p, lo, min(hi, tb + 1))
                n += len(ofr)
            assert n > 10000
        finally:
            p.close()


def case_variants_rejected(backend_cls, workdir):
    """more alleles than Reference::Variant::kMaxAlleles (128) are refused with the reference's message (Reference.cpp:1016-1019)"""
    import pytest
    lengths = [3000]
    seqs = make_inputs(workdir, "vars_rej", synth.TINY, lengths)[2]
    vcf = workdir / "many.vcf"
    alt = "A" if seqs[0][1][100] != 0 else "C"
    write_vcf(vcf, seqs, [(0, 100, 1, alt, "\t".join(["0|1"] * 65))], samples=65)
    ppath, fpath, _ = make_inputs(workdir, "vars_rej", synth.TINY, lengths)
    with pytest.raises(Exception, match="only 128 alleles are supported"):
        backend_cls(ppath, fpath, 0, None, vcf_path=vcf)
