"""ReSeq's own profile files (`.reseq` / `.reseq.ipf`, Boost text archives) -> the prepared profile.

Checks, without a GPU (rsq_profile_load* / rsq_profile_save are host code of the product library):
  * the token grammar of the writer the tests use, against hand-written archives (the rules of rsq_archive.h);
  * product reader (reseq_amd/csrc/rsq_profile_archive.cpp) == oracle reader (oracle/reseq_archive.py) == the synthetic profile
    the archives were made from, array by array, bit for bit;
  * FullExpansion / GetResults / ImputeMissingValues on cases worked out by hand from the statements of
    ProbabilityEstimates.h:253-290, 386-479, 1004-1036;
  * AdapterStats::SumCounts / PrepareSimulation, ErrorStats::PrepareSimulation on hand-made counts;
  * the error paths.
Compatibility with a file written by Boost itself is NOT verified anywhere (no Boost, no sample profile in the image).
"""
import math

import numpy as np
import pytest

import archive_fixtures as af
from oracle_archive import ra
from reseq_amd import api, synth
from reseq_amd.container import read_container


# ------------------------------------------------------------------------------------------------ token grammar
def test_writer_token_grammar():
    head = "22 serialization::archive 17 "
    # vector of arithmetic items: no class info; count, item version, items
    assert ra.dumps("vector<u16>", [5, 6, 7]) == head + "3 0 5 6 7\n"
    # std::array: class info, then the C array inside (count + items, no item version)
    assert ra.dumps("array<u64,3>", [1, 2, 3]) == head + "0 0 3 1 2 3\n"
    # the shape of the well-known std::map<int,int>{{1,2},{3,4}} archive: container info, count, item version, pair info once
    assert ra.dumps("vector<pair<u32,u32>>", [(1, 2), (3, 4)]) == head + "0 0 2 0 0 0 1 2 3 4\n"
    # vector<bool> has no item version; the outer vector of vectors has class info
    assert ra.dumps("vector<vector<bool>>", [[True, False], [True]]) == head + "0 0 2 0 2 1 0 1 1\n"
    # strings: length, blank, bytes (blanks inside are kept); char-sized integers as numbers; doubles with 17 digits
    assert ra.dumps("vector<string>", ["adapter one", ""]) == head + "0 0 2 0 11 adapter one 0 \n"
    assert ra.dumps("u8", 33) == head + "33\n"
    assert ra.dumps("f64", 0.1) == head + "1.00000000000000006e-01\n"
    # a class type is announced once per archive, also when it recurs inside other containers
    v = ra.vect(2, [9, 8])
    assert ra.dumps("array<Vect<u64>,2>", [v, v]) == head + "0 0 2 0 0 0 0 2 2 0 9 8 2 2 0 9 8\n"
    # library versions up to 3 write no item version
    assert ra.dumps("vector<u16>", [5], library_version=3) == "22 serialization::archive 3 1 5\n"


def test_reader_inverts_writer():
    rng = np.random.default_rng(3)
    ipf = af.rebin(af.ipf_from_table(np.asarray([4, 1, 2]), [(3, 6), (0, 2), (7, 9)],
                                     [rng.random((3, 3)), rng.random((2, 3)), rng.random((2, 3))]), rng)
    text = ra.dumps("LogIPF<4>", ipf)
    back = ra.loads("LogIPF<4>", text)
    assert back["dim_indices_"] == ipf["dim_indices_"] and back["steps_"] == 37 and back["precision_"] == 0.01
    for a, b in zip(back["estimates_"]["dim2_"], ipf["estimates_"]["dim2_"]):
        assert np.array_equal(a, np.asarray(b))
    assert ra.dumps("LogIPF<4>", back) == text
    with pytest.raises(ValueError):
        ra.loads("LogIPF<4>", text[:-40])
    with pytest.raises(ValueError):
        ra.loads("LogIPF<4>", text + " 1")


# ---------------------------------------------------------------------------------------------- whole profiles
@pytest.fixture(scope="module")
def tiny_archives(tmp_path_factory, tiny_profile_arrays):
    d = tmp_path_factory.mktemp("archives")
    plain = str(d / "tiny.reseq")
    af.write_profile_archives(plain, tiny_profile_arrays)
    binned = str(d / "tiny_binned.reseq")
    af.write_profile_archives(binned, tiny_profile_arrays, rng=np.random.default_rng(99), ipf_path=str(d / "elsewhere.ipf"))
    return dict(dir=d, plain=plain, binned=binned, binned_ipf=str(d / "elsewhere.ipf"))


def _product_arrays(stats, ipf, out):
    p = api.Profile(stats, ipf_path=ipf)
    p.save(out)
    return read_container(out), p.warning


def _assert_same(got, want, what):
    for name, v in got.items():
        w = np.asarray(want[name])
        assert v.dtype == w.dtype or name.startswith("tab.") and v.dtype.kind == w.dtype.kind, (what, name, v.dtype, w.dtype)
        assert v.size == w.size and np.array_equal(v.ravel(), w.ravel()), (what, name)


def test_archive_profile_equals_the_profile_it_was_made_from(tiny_archives, tiny_profile_arrays):
    got, warning = _product_arrays(tiny_archives["plain"], None, str(tiny_archives["dir"] / "plain.rsqp"))
    assert "fitted tables" not in warning and warning.startswith("read as Boost text archives of library version 17")
    assert set(tiny_profile_arrays) - set(got) == {"frag.sur_bias_separated"}
    _assert_same(got, tiny_profile_arrays, "product vs source")
    _assert_same(got, ra.load_profile(tiny_archives["plain"]), "product vs oracle")


def test_the_product_writes_the_archives_the_fixture_writes(tiny_archives, tiny_profile_arrays, tmp_path):
    """rsq_profile_save_reseq: a loaded profile goes out as ReSeq's own pair of files.  From the container of the TINY profile the product writes, token for token, what
    the tests' own writer (oracle/reseq_archive.py through tests/archive_fixtures.py) writes for it; the pair loads to the same tables bit for bit; a profile that came
    from archives with re-binned fits and imputed rows goes out and comes back unchanged as well (its tables are what PrepareResult left)."""
    from reseq_amd import synth
    src = str(tmp_path / "tiny.rsqp")
    synth.write_profile(src, tiny_profile_arrays)
    p = api.Profile(src)
    out = str(tmp_path / "written.reseq")
    p.save_reseq(out, creation_time=1600000000)
    p.close()
    assert open(out, "rb").read() == open(tiny_archives["plain"], "rb").read()
    assert open(out + ".ipf", "rb").read() == open(tiny_archives["plain"] + ".ipf", "rb").read()
    got, warning = _product_arrays(out, None, str(tmp_path / "back.rsqp"))
    assert "does not follow the recalled token rules" not in warning
    _assert_same(got, tiny_profile_arrays, "written and read back vs source")
    # a profile from re-binned fits with imputed rows
    binned = api.Profile(tiny_archives["binned"], ipf_path=tiny_archives["binned_ipf"])
    binned.save(str(tmp_path / "binned.rsqp"))
    binned.save_reseq(str(tmp_path / "binned_again.reseq"), ipf_path=str(tmp_path / "binned_again.fit"))
    binned.close()
    again, _ = _product_arrays(str(tmp_path / "binned_again.reseq"), str(tmp_path / "binned_again.fit"), str(tmp_path / "binned_back.rsqp"))
    # Rows that ImputeMissingValues filled change the columns' mean likelihoods, by which GetResults sorts them: such a table comes back with its columns in another
    # order (the same distribution; the original binary would sort the same way).  Compared with the columns in the order of their outcome values.
    def by_outcome(arrays):
        out = dict(arrays)
        for name in arrays:
            if name.startswith("tab.") and name.endswith(".par0"):
                par0 = np.asarray(arrays[name])
                order = np.argsort(par0, kind="stable")
                out[name] = par0[order]
                flat = np.asarray(arrays[name[:-5] + ".dim2"])
                if len(par0):
                    out[name[:-5] + ".dim2"] = flat.reshape(-1, len(par0))[:, order].ravel()
        return out
    want = read_container(str(tmp_path / "binned.rsqp"))
    _assert_same(by_outcome(again), by_outcome(want), "re-binned profile written and read back")
    assert any(not np.array_equal(again[n], want[n]) for n in want if n.endswith(".par0"))      # (the case exists in the fixture)
    with pytest.raises(api.RsqError):
        api.Profile(src).save_reseq(str(tmp_path / "no_such_directory" / "x.reseq"))


def test_binned_tables_and_missing_rows_product_equals_oracle(tiny_archives, tiny_profile_arrays):
    got, _ = _product_arrays(tiny_archives["binned"], tiny_archives["binned_ipf"], str(tiny_archives["dir"] / "binned.rsqp"))
    want = ra.load_profile(tiny_archives["binned"], tiny_archives["binned_ipf"])
    _assert_same(got, want, "product vs oracle")
    # the fixture really did bin and knock out rows: tables differ from the source, statistics do not
    changed = [n for n in got if n.startswith("tab.") and (got[n].size != np.asarray(tiny_profile_arrays[n]).size
                                                           or not np.array_equal(got[n].ravel(), np.asarray(tiny_profile_arrays[n]).ravel()))]
    assert len(changed) > 100
    assert all(np.array_equal(got[n].ravel(), np.asarray(tiny_profile_arrays[n]).ravel()) for n in got if not n.startswith("tab."))
    # imputed rows exist: a quality table's position margin has rows that no index of the fit points to, and none is empty
    p = api.Profile(tiny_archives["binned"], ipf_path=tiny_archives["binned_ipf"])
    assert p.max_read_length() == synth.TINY["read_len_max"]


def test_rsqp_container_is_still_loaded(tiny_profile_path, tmp_path):
    p = api.Profile(tiny_profile_path)
    p.save(str(tmp_path / "again.rsqp"))
    a, b = read_container(tiny_profile_path), read_container(str(tmp_path / "again.rsqp"))
    _assert_same(b, a, "save(load(rsqp))")


# ------------------------------------------------------------------------------- PrepareResult, worked out by hand
def _hand_ipf():
    """LogIPF<4>: outcomes {7,3,5}; condition 1 seen at values {2,4}, condition 2 at {10,13}, condition 3 at {0}."""
    m0 = np.asarray([[0.5, 0.25, 1.0], [1.5, 0.75, 1.0]])        # margin (1,0): rows = condition 1, columns = outcome
    m1 = np.asarray([[2.0, 1.0, 0.5], [2.0, 3.0, 0.5]])          # margin (2,0)
    m2 = np.asarray([[1.0, 1.0, 4.0]])                           # margin (3,0)
    ipf = af.ipf_from_table(np.asarray([7, 3, 5]), [(0, 2), (0, 2), (0, 1)], [m0, m1, m2])
    ipf["dim_indices_"] = [[7, 3, 5], [2, 4], [10, 13], [0]]
    return ipf


def _hand_expected():
    # mean likelihood per outcome: m2 [1,1,4] + m1 [2,2,.5] + m0 [1,.5,1] = [4, 3.5, 5.5]  ->  ascending: 3, 7, 5
    par0 = [3, 7, 5]
    limits = [[2, 5], [10, 14], [0, 1]]
    rows0 = [[0.25, 0.5, 1.0],
             [0.25 * 1 / 2 + 0.75 * 1 / 2, 0.5 * 1 / 2 + 1.5 * 1 / 2, 1.0 * 1 / 2 + 1.0 * 1 / 2],       # value 3: no data
             [0.75, 1.5, 1.0]]
    lo, hi = [1.0, 2.0, 0.5], [3.0, 2.0, 0.5]
    # values 11 and 12 have no data.  ProbabilityEstimates.h:472: row = last*(gap-last)/(i-last) + next*(i-gap)/(i-last):
    # value 11 (one step from 10) takes 1/3 of row 10 and 2/3 of row 13 -- the nearer row gets the smaller weight
    rows1 = [lo, [l * 1 / 3 + h * 2 / 3 for l, h in zip(lo, hi)], [l * 2 / 3 + h * 1 / 3 for l, h in zip(lo, hi)], hi]
    rows2 = [[1.0, 1.0, 4.0]]
    return par0, limits, np.concatenate([np.ravel(rows0), np.ravel(rows1), np.ravel(rows2)])


def test_get_results_order_limits_imputation_oracle():
    t = ra.prepare_table(_hand_ipf(), 4)
    par0, limits, dim2 = _hand_expected()
    assert t["par0"].tolist() == par0 and t["limits"].tolist() == limits
    assert np.array_equal(t["dim2"], dim2)
    assert dim2[9 + 3] == 1.0 * 1 / 3 + 3.0 * 2 / 3 and dim2[9 + 3] > 2.3          # nearer to row 10 (1.0), yet closer to row 13 (3.0)


def test_get_results_ties_keep_index_order_and_empty_tables():
    ipf = af.ipf_from_table(np.asarray([9, 8, 7]), [(0, 1), (0, 1), (0, 1)], [np.asarray([[1.0, 1.0, 0.5]])] * 3)
    t = ra.prepare_table(ipf, 4)
    assert t["par0"].tolist() == [7, 9, 8]                        # 7 has the lowest mean; 9 and 8 tie and keep their order
    empty = ra.prepare_table(af.ipf_from_table(np.zeros(0, np.uint32), [(0, 0)] * 3, [np.zeros((0, 0))] * 3), 4)
    assert len(empty["par0"]) == 0 and empty["limits"].tolist() == [[0, 0]] * 3 and len(empty["dim2"]) == 0


def _hand_binned_ipf():
    """LogIPF<4> stored reduced.  Outcomes {1,2,3}: the first two share a bin (initial reduction), conditions un-binned except
    condition 1, whose three values collapse 3 -> 2 (initial) -> 1 (final)."""
    ipf = af.ipf_from_table(np.asarray([1, 2, 3]), [(0, 3), (5, 6), (0, 1)], [np.ones((3, 3)), np.ones((1, 3)), np.ones((1, 3))])
    ipf["initial_dim_indices_reduced_"] = [[0, 0, 1], [0, 1, 1], [0], [0]]
    ipf["dim_indices_reduced_"] = [[0, 1], [0, 0], [0], [0]]
    # stored margins over the bins: (1,0) is 1 x 2, (2,0) 1 x 2, (3,0) 1 x 2; the three others 1 x 1
    ipf["estimates_"] = dict(dim2_=[[8.0, 27.0], [2.0, 5.0], [1.0, 3.0], [1.0], [1.0], [1.0]], dim_size_=[2, 1, 1, 1])
    return ipf


def test_full_expansion_by_hand():
    t = ra.prepare_table(_hand_binned_ipf(), 4)
    half, third = math.pow(1.0 / 2, 1.0 / 3), math.pow(1.0 / 3, 1.0 / 3)     # (1/bins sharing)^(1/(N-1)), N = 4
    # outcome columns before sorting: [bin0, bin0, bin1] with weights [half, half, 1]
    m0 = [[8.0 * third * half, 8.0 * third * half, 27.0 * third * 1.0]] * 3   # the three values of condition 1 share ONE bin
    m1 = [[2.0 * 1.0 * half, 2.0 * 1.0 * half, 5.0 * 1.0 * 1.0]]
    m2 = [[1.0 * 1.0 * half, 1.0 * 1.0 * half, 3.0 * 1.0 * 1.0]]
    # means: outcomes 1 and 2 tie (index order kept), outcome 3 is larger
    assert t["par0"].tolist() == [1, 2, 3] and t["limits"].tolist() == [[0, 3], [5, 6], [0, 1]]
    assert np.array_equal(t["dim2"], np.concatenate([np.ravel(m0), np.ravel(m1), np.ravel(m2)]))


def test_hand_cases_through_the_product(tiny_archives, tiny_profile_arrays, tmp_path):
    st, pe = af.from_rsqp(tiny_profile_arrays)
    pe["indels_"][0][0] = _hand_ipf()
    pe["indels_"][1][5] = _hand_binned_ipf()
    pe["dom_error_"][3][4][4] = af.ipf_from_table(np.asarray([9, 8, 7]), [(0, 1), (0, 1), (0, 1)], [np.asarray([[1.0, 1.0, 0.5]])] * 3)
    ipf_path = str(tmp_path / "hand.ipf")
    ra.write_archive(ipf_path, "ProbabilityEstimates", pe)
    got, _ = _product_arrays(tiny_archives["plain"], ipf_path, str(tmp_path / "hand.rsqp"))
    par0, limits, dim2 = _hand_expected()
    assert got["tab.indels.0.0.par0"].tolist() == par0 and got["tab.indels.0.0.limits"].tolist() == limits
    assert np.array_equal(got["tab.indels.0.0.dim2"], dim2)
    want = ra.prepare_table(_hand_binned_ipf(), 4)
    assert np.array_equal(got["tab.indels.1.5.dim2"], want["dim2"]) and got["tab.indels.1.5.par0"].tolist() == [1, 2, 3]
    assert got["tab.dom_error.3.4.4.par0"].tolist() == [7, 9, 8]


# ---------------------------------------------------------------------------------- DataStats::PrepareProcessing by hand
def test_adapter_counts_by_hand(tiny_archives, tiny_profile_arrays, tmp_path):
    st, pe = af.from_rsqp(tiny_profile_arrays)
    ad = st["adapters_"]
    # three adapters per segment; segment 0: ACGT.., ACGA.., TTTT..: the first two share "ACG" (first difference at 3)
    ad["seqs_archive"] = [["ACGTAC", "ACGAAC", "TTTTTT"], ["GGGGGG", "GGCCCC", "GGCCAA"]]
    v = ra.vect

    def pair(table, first1=0, first2=0):
        return {"vec_": (first1, [v(first2, row) for row in table])}

    # counts_[a1][a2][length of adapter 1 seen][length of adapter 2 seen]
    zero = {"vec_": (0, [])}
    counts = [[zero, zero, zero] for _ in range(3)]
    counts[0][0] = pair([[1, 1, 1, 1, 1, 1, 1]] * 7)            # 7 x 7 ones starting at (0, 0)
    counts[1][2] = pair([[10, 20, 30]], first1=5, first2=3)      # lengths (5, 3..5)
    counts[2][1] = pair([[100], [200]], first1=0, first2=6)      # lengths (0..1, 6)
    ad["counts_"] = counts
    stats_path = str(tmp_path / "adapters.reseq")
    ra.write_archive(stats_path, "DataStats", st)
    got, _ = _product_arrays(stats_path, tiny_archives["plain"] + ".ipf", str(tmp_path / "adapters.rsqp"))
    # segment 0 thresholds (first length that tells an adapter from both neighbours in the list): a0: 3, a1: 3, a2: 0
    # segment 1: G6 vs GGC4: 2; GGC4 vs GGCCAA: max(2, 4) = 4; GGCCAA: 4
    # pair (0,0): lengths 3..6 x 2..6 of the 7 x 7 ones = 4 x 5 = 20
    # pair (1,2): length 5 >= 3; lengths 3,4,5 of adapter 2 against threshold 4: 20 + 30 = 50
    # pair (2,1): lengths 0,1 >= 0; length 6 >= 4: 300
    assert got["adapters.0.counts"].tolist() == [20, 50, 300]
    assert got["adapters.1.counts"].tolist() == [20, 300, 50]
    # AdapterStats::PrepareSimulation: below ceil(0.1 * 300) = 30 -> not simulated
    assert got["adapters.0.significant_counts"].tolist() == [0, 50, 300]
    assert got["adapters.1.significant_counts"].tolist() == [0, 300, 50]
    want = ra.prepare_stats(ra.read_archive(stats_path, "DataStats"))
    for name in ("adapters.0.counts", "adapters.1.significant_counts", "adapters.0.seqs", "adapters.1.seq_ptr"):
        assert np.array_equal(got[name], want[name])
    # ErrorStats::PrepareSimulation: the longest run of deletions after any call (from_rsqp stored up to max_len_deletion of them)
    assert int(got["errors.max_len_deletion"][0]) == synth.TINY["max_len_deletion"]


# ------------------------------------------------------------------------------------------------------ error paths
def test_error_paths(tiny_archives, tiny_profile_arrays, tmp_path):
    with pytest.raises(api.RsqError, match="does not exists"):
        api.Profile(str(tmp_path / "missing.reseq"), ipf_path="x")
    # statistics without their probability estimates
    lonely = str(tmp_path / "lonely.reseq")
    with open(tiny_archives["plain"], "rb") as f:
        data = f.read()
    with open(lonely, "wb") as f:
        f.write(data)
    with pytest.raises(api.RsqError, match="lonely.reseq.ipf"):
        api.Profile(lonely)
    # truncated archive
    with open(lonely, "wb") as f:
        f.write(data[:len(data) // 2])
    with pytest.raises(api.RsqError, match="lonely.reseq"):
        api.Profile(lonely, ipf_path=tiny_archives["plain"] + ".ipf")
    # estimates fitted to other statistics (ProbabilityEstimates.cpp:1079-1083: the reference would refit)
    st, pe = af.from_rsqp(tiny_profile_arrays, creation_time=1234)
    other = str(tmp_path / "other.ipf")
    ra.write_archive(other, "ProbabilityEstimates", pe)
    with pytest.raises(api.RsqError, match="another statistics file"):
        api.Profile(tiny_archives["plain"], ipf_path=other)
    # a table stored above the precision aim is used as stored and reported
    st, pe = af.from_rsqp(tiny_profile_arrays)
    pe["indels_"][0][0]["precision_"] = 0.2
    loose = str(tmp_path / "loose.ipf")
    ra.write_archive(loose, "ProbabilityEstimates", pe)
    assert "1 fitted tables" in api.Profile(tiny_archives["plain"], ipf_path=loose).warning
    assert "fitted tables" not in api.Profile(tiny_archives["plain"], ipf_path=loose, ipf_precision=25.0).warning
    assert "library version 17" in api.Profile(tiny_archives["plain"], ipf_path=loose, ipf_precision=25.0).warning          # said every time: compatibility is unverified
    # not an archive and not a container
    junk = str(tmp_path / "junk")
    with open(junk, "w") as f:
        f.write("23 something else")
    with pytest.raises(api.RsqError):
        api.Profile(junk)


# ------------------------------------------------------------------------------------------------------ the doubtful token rules
def test_every_alternative_of_the_doubtful_token_rules_loads_to_the_same_tables(tiny_profile_arrays, tmp_path):
    """The reader's token rules are recalled, not checked against a Boost-written file.  The doubtful ones are switches (rsq_archive.h
    Grammar: item_version behind which vector counts, class information for std::array / std::pair / vectors of arithmetic types, the count inside a std::array); a
    file that does not parse under the recalled set is read under the others, and the fixed sizes in the schema tell which fits.  The TINY profile written under
    EVERY combination must load to bit-identical tables, and the warning must name the rules that fitted."""
    import os
    grammars = ra.all_grammars()
    assert len(grammars) == 64 and grammars[0] == ra.DEFAULT_GRAMMAR
    if not os.environ.get("RSQ_ALL_GRAMMARS"):              # every switch on its own, and everything flipped at once (all 64: RSQ_ALL_GRAMMARS=1, five minutes)
        flips = lambda g: sum(g[k] != ra.DEFAULT_GRAMMAR[k] for k in g)
        grammars = [g for g in grammars if flips(g) <= 1 or (flips(g) == 5 and g["item_version"] == 0)]
        assert len(grammars) == 9
    trees = af.from_rsqp(tiny_profile_arrays, 1600000000, np.random.default_rng(7))        # binned tables with missing rows: every container type is in use
    want = None
    for k, g in enumerate(grammars):
        path = str(tmp_path / f"g{k}.reseq")
        af.write_profile_archives(path, tiny_profile_arrays, grammar=g, trees=trees)
        got, warning = _product_arrays(path, None, str(tmp_path / f"g{k}.rsqp"))
        if want is None:
            want = got
            assert "does not follow the recalled token rules" not in warning
        else:
            _assert_same(got, want, f"grammar {g}")
            assert warning.count("does not follow the recalled token rules") in (1, 2), warning           # both files, unless one holds nothing the rule is about (no std::pair in the .ipf)
            assert ("no item_version" in warning) == (g["item_version"] == 0) and ("std::pair without class information" in warning) == (not g["pair_class_info"])
            assert ("std::array without class information, " in warning) == (not g["array_class_info"])
            if k in (1, len(grammars) - 1):
                layout = api.archive_layout(path)
                assert layout.count("token_rules\tNOT as recalled") in (1, 2) and layout.count("parsed\tto the end") == 2
    # two files of one profile need not share the rules (each is tried on its own)
    mixed = str(tmp_path / "mixed.reseq")
    ra.write_archive(mixed, "DataStats", trees[0], grammar=grammars[5])
    ra.write_archive(mixed + ".ipf", "ProbabilityEstimates", trees[1], grammar=grammars[-1])
    _assert_same(_product_arrays(mixed, None, str(tmp_path / "mixed.rsqp"))[0], want, "mixed grammars")


# ------------------------------------------------------------------------------------------------------ diagnosis
def test_archive_layout_and_the_member_path_of_a_parse_error(tiny_archives, tmp_path):
    """The reader cannot be validated against a Boost-written file here, so a file it cannot read must say WHERE: the layout lists the class-info site of
    every serialized type (byte, tracking, version, type, member path of the first object); a parse error names the member path, its type and the
    class-info sites of the types around it."""
    plain = tiny_archives["plain"]
    layout = api.archive_layout(plain)
    files = layout.split("# ")[1:]
    assert len(files) == 2 and files[0].startswith(plain + " (DataStats)") and "(ProbabilityEstimates)" in files[1]
    for text in files:
        assert "library_version\t17" in text and "parsed\tto the end" in text and "\nerror\t" not in text
    rows = [l.split("\t") for l in files[0].splitlines() if l[:1].isdigit()]
    assert len(rows) > 25                                                  # the serialized class types of DataStats
    assert rows[0][3] == "reseq::DataStats" and rows[0][4] == "DataStats" and all(r[1] == "0" and r[2] == "0" for r in rows)
    bytes_ = [int(r[0]) for r in rows]
    assert bytes_ == sorted(bytes_)                                        # sites in stream order
    assert any(r[4].startswith("DataStats.adapters_.") for r in rows) and any("std::array<" in r[3] for r in rows)
    # a profile from a Boost that writes one class-info record fewer (here: the pair of tokens of one site removed): every later token shifts
    data = open(plain, "rb").read()
    site = rows[len(rows) // 2]
    at = int(site[0])
    assert data[at:at + 4] == b"0 0 "
    broken = str(tmp_path / "shifted.reseq")
    with open(broken, "wb") as f:
        f.write(data[:at] + data[at + 4:])
    with pytest.raises(api.RsqError) as e:
        api.Profile(broken, ipf_path=plain + ".ipf")
    msg = str(e.value)
    assert " at DataStats." in msg and "near byte" in msg and "enclosing types:" in msg and "class info read at byte" in msg and "archive library version 17" in msg
    assert "no alternative of the doubtful token rules (64 combinations" in msg                # every grammar was tried; the recalled rules' message is the one shown
    shifted = api.archive_layout(broken, plain + ".ipf")
    assert "\nerror\t" in shifted.split("# ")[1] and "parsed\tto the end" in shifted.split("# ")[2]
    got = [l.split("\t") for l in shifted.split("# ")[1].splitlines() if l[:1].isdigit()]
    assert [r[3] for r in got[:len(rows) // 2]] == [r[3] for r in rows[:len(rows) // 2]]      # the sites in front of the damage are where they were
    # a truncated file: the path of the member the text ends in
    cut = str(tmp_path / "cut.reseq")
    with open(cut, "wb") as f:
        f.write(data[:len(data) // 3])
    with pytest.raises(api.RsqError, match=r" at DataStats\.[a-z_]+"):
        api.Profile(cut, ipf_path=plain + ".ipf")
