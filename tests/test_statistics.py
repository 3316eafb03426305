"""Empirical frequencies against the conditionals the tables imply -- a check that does NOT go through the oracle's reading of
LogArrayResult::Draw / FillReadPart (ProbabilityEstimates.h:359-380,481-508, Simulator.cpp:294-452).

seqToIllumina records are simulated with a profile whose sequence-quality tables have a single outcome and whose indel tables are
reduced to "no indel" (ProbabilityEstimates::RemoveInDelErrors), so that every conditioning value of every quality and base-call draw
can be read off the output: table = (segment, tile, reference base[, dominant error]); rows = (sequence quality, previous quality,
read position, systematic error rate) resp. (quality, read position, errors so far, systematic error rate).  For each draw the
conditional P(outcome) = prod_n T_n[row_n][outcome] / sum is evaluated in numpy from the profile's arrays, and observed counts are
compared with expected ones per (table, outcome), per (table, position, outcome) and per (table, previous quality / quality, outcome):
z = (O - E) / sqrt(sum p (1 - p)) must look standard normal.  A wrong row index, a wrong clamp, a transposed margin or a biased
inverse-CDF scan shows up as |z| of tens to hundreds.

The CPU oracle runs without a GPU, the product on the device (`-m gpu`).
"""
import numpy as np
import pytest

import oracle_lib as O
from reseq_amd import synth
from reseq_amd.container import write_container

N_READS, READ_LEN, SEED = 60_000, 30, 4711


def _profile(workdir):
    arrays = synth.make_profile(synth.TINY, seed=5)
    sq_value = synth.TINY["qual_from"] + 3
    for seg in range(2):
        for tile in range(len(arrays["tiles.tiles"])):
            lim = arrays[f"tab.seq_quality.{seg}.{tile}.limits"].reshape(-1, 2)
            rows = int((lim[:, 1] - lim[:, 0]).sum())
            arrays[f"tab.seq_quality.{seg}.{tile}.par0"] = np.asarray([sq_value], np.uint32)
            arrays[f"tab.seq_quality.{seg}.{tile}.dim2"] = np.ones(rows)
    path = workdir / "stat_profile.rsqp"
    write_container(path, arrays)
    return arrays, str(path), sq_value


def _table(arrays, prefix):
    par0 = arrays[f"tab.{prefix}.par0"].astype(np.int64)
    lim = arrays[f"tab.{prefix}.limits"].reshape(-1, 2).astype(np.int64)
    flat, k, pos, margins = arrays[f"tab.{prefix}.dim2"], len(par0), 0, []
    for lo, hi in lim:
        margins.append(flat[pos:pos + (hi - lo) * k].reshape(hi - lo, k))
        pos += (hi - lo) * k
    return par0, lim, margins


def _conditionals(table, states):
    """P(column | state) for every draw: states is an (N, NM) integer array of conditioning VALUES"""
    par0, lim, margins = table
    p = np.ones((len(states), len(par0)))
    for n, m in enumerate(margins):
        row = np.clip(states[:, n] - lim[n, 0], 0, lim[n, 1] - lim[n, 0] - 1)      # AdjustIndeces: below -> first row, at/above -> last
        p *= m[row]
    s = p.sum(axis=1, keepdims=True)
    ok = s[:, 0] > 0
    return p[ok] / s[ok], ok


def _z_scores(p, outcome_col, groups):
    """z per (group, outcome): groups is an integer label per draw"""
    n_groups, k = int(groups.max()) + 1, p.shape[1]
    expected, var = np.zeros((n_groups, k)), np.zeros((n_groups, k))
    np.add.at(expected, groups, p)
    np.add.at(var, groups, p * (1 - p))
    observed = np.zeros((n_groups, k))
    np.add.at(observed, (groups, outcome_col), 1.0)
    usable = var > 25.0                                  # normal approximation
    return (observed[usable] - expected[usable]) / np.sqrt(var[usable])


def _check(results, rec, arrays, sq_value, enforce=True):
    tiles = {int(t): i for i, t in enumerate(arrays["tiles.tiles"])}
    quality_draws, base_draws = {}, {}
    for i, (seq, qual, cigar, nerr, tile) in enumerate(results):
        n_m = int(cigar.split("M")[0]) if "M" in cigar and cigar.split("M")[0].isdigit() else 0
        if n_m == 0:
            continue
        seg, tile_id = int(rec["seg"][i]), tiles[int(tile)] if int(tile) in tiles else int(tile)
        ref, dom, rate = rec["seqs"][i, :n_m].astype(np.int64), rec["dom"][i, :n_m].astype(np.int64), rec["rate"][i, :n_m].astype(np.int64)
        q = np.frombuffer(qual[:n_m], np.uint8).astype(np.int64) - 33
        call = np.frombuffer(seq[:n_m], np.uint8).astype(np.int64)
        prev = np.concatenate([[1], q[:-1]])                           # ReadFillParameter::qual_ starts at 1 (Simulator.h:232)
        errors_before = np.concatenate([[0], np.cumsum(call != ref)[:-1]])
        pos = np.arange(n_m)
        for b in range(4):
            sel = ref == b
            if sel.any():
                quality_draws.setdefault((seg, tile_id, b), []).append(np.stack([np.full(sel.sum(), sq_value), prev[sel], pos[sel], rate[sel], q[sel]], axis=1))
                for d in range(5):
                    sd = sel & (dom == d)
                    if sd.any():
                        base_draws.setdefault((seg, tile_id, b, d), []).append(np.stack([q[sd], pos[sd], errors_before[sd], rate[sd], call[sd]], axis=1))
        assert int((call != ref).sum()) <= nerr                          # E<n> counts the whole read, the template part is a subset

    worst, n_tests, n_draws = 0.0, 0, 0
    for family, draws, group_cols in (("quality", quality_draws, (2, 1, 3)), ("base_call", base_draws, (0, 1, 2))):
        for key, parts in draws.items():
            a = np.concatenate(parts)
            table = _table(arrays, family + "." + ".".join(map(str, key)))
            if not len(table[0]):
                continue
            p, ok = _conditionals(table, a[:, :4])
            a = a[ok]
            col_of = {int(v): c for c, v in enumerate(table[0])}
            assert all(int(v) in col_of for v in np.unique(a[:, 4])), (family, key, "an outcome the table does not have")
            col = np.asarray([col_of[int(v)] for v in a[:, 4]])
            assert (p[np.arange(len(col)), col] > 0).all(), (family, key, "an outcome of probability 0 was drawn")
            n_draws += len(a)
            for z in [_z_scores(p, col, np.zeros(len(a), np.int64))] + [_z_scores(p, col, a[:, c] - a[:, c].min()) for c in group_cols]:
                if len(z):
                    worst = max(worst, float(np.abs(z).max()))
                    n_tests += len(z)
    if enforce:
        assert n_draws > 1_500_000 and n_tests > 2_000, (n_draws, n_tests)
        assert worst < 5.5, worst                          # the largest of n_tests standard normal values: beyond 5.5 with probability 4e-8 each
    return worst, n_tests, n_draws


def _records():
    arrays = synth.make_profile(synth.TINY, seed=5)
    rec = synth.make_error_model_input(77, N_READS, READ_LEN, arrays, zero_frac=0.7)
    rec["frag_len"][:] = np.maximum(rec["frag_len"], READ_LEN + 5)          # whole reads are template ('M'): no adapter part
    return rec


def test_conditionals_of_the_oracle(workdir):
    arrays, path, sq_value = _profile(workdir)
    rec = _records()
    prof = O.Profile(path)
    O.lib().orc_profile_remove_indel_errors(prof.h)
    results = O.error_model_only(prof, SEED, rec)
    prof.close()
    _check(results, rec, arrays, sq_value)


def test_the_check_notices_a_shifted_margin(workdir):
    """the test of the test: the same data evaluated against tables whose position margin is shifted by one row must fail loudly"""
    arrays, path, sq_value = _profile(workdir)
    rec = _records()
    prof = O.Profile(path)
    O.lib().orc_profile_remove_indel_errors(prof.h)
    results = O.error_model_only(prof, SEED, rec)
    prof.close()
    wrong = dict(arrays)
    for name in list(arrays):
        if name.startswith("tab.quality.") and name.endswith(".dim2"):
            par0, lim, margins = _table(arrays, name[4:-5])
            margins[2] = np.roll(margins[2], 1, axis=0)
            wrong[name] = np.concatenate([m.ravel() for m in margins])
    sub = rec
    worst, n_tests, _ = _check(results, sub, wrong, sq_value, enforce=False)
    assert worst > 8 and n_tests > 500, (worst, n_tests)          # neighbouring position rows are similar: a shift of one row is a subtle error
    assert _check(results, sub, arrays, sq_value, enforce=False)[0] < 5.5


@pytest.mark.gpu
def test_conditionals_of_the_product(workdir):
    from backends import GpuBackend
    arrays, path, sq_value = _profile(workdir)
    rec = _records()
    b = GpuBackend(path, None, 0, {"no_indels": True})
    b.prepare(SEED)
    results = b.error_model(rec)
    b.close()
    _check(results, rec, arrays, sq_value)


def test_gap_draws_pass_every_cell_with_its_own_probability(workdir):
    """The sieve draws the gaps between the cells that pass the zero threshold instead of a uniform per cell (oracle_sim.c orc_gap_hits,
    rsq_kernels.h sieve_gaps; Simulator.cpp:2304-2306).  Per fragment length the passes over many start positions must be binomial with
    p = 1 - thr1[length], passes of neighbouring lengths must be uncorrelated, and probability_chosen of a passing cell must be uniform
    on [thr1, 1).  Also with thresholds small enough that the running product is restarted (segments), and with a threshold of zero."""
    import parity_cases as P
    ppath, fpath, seqs = P.make_inputs(workdir, "gaps", synth.TINY, [6000])
    oprof, oref = O.Profile(ppath), O.Reference(seqs)
    sim = O.Sim(oprof, oref, 77, num_pairs=9000)
    try:
        n_starts = 40_000
        base = sim.thresholds()
        to = base.shape[1]
        lo = int(sim.gap_passes(0, [])[2][0])                   # the first fragment length, max(1, InsertLengths().from())
        for variant in ("profile", "small", "with_zero"):
            thr = base.copy()
            if variant == "small":                              # cells pass with probability 0.7 .. 1 - 1e-9: the running product would underflow without segments
                thr[0, :, 1] = np.geomspace(3e-1, 1e-9, to)
                thr[0, :, 0] = np.sqrt(thr[0, :, 1])
            if variant == "with_zero":
                thr[0, lo + 7, 1] = 0.0
                thr[0, lo + 7, 0] = 0.0
            sim.set_normalization(sim.bias_normalization(), thr)
            n = 4000 if variant == "small" else n_starts
            passes, q, seg_end = sim.gap_passes(0, np.arange(n))
            if variant == "small":
                assert len(set(seg_end[lo:].tolist())) > 1      # several segments
            start = np.array([p[0] for p in passes])
            length = np.array([p[1] for p in passes])
            pc = np.array([p[2] for p in passes])
            assert length.min() >= lo and length.max() < to
            assert len(set(zip(start.tolist(), length.tolist()))) == len(passes)          # a cell passes once
            p_pass = 1 - thr[0, :, 1]
            counts = np.bincount(length, minlength=to).astype(float)
            sel = np.arange(lo, to)
            var = n * p_pass[sel] * (1 - p_pass[sel])
            ok = var > 25                                       # the normal approximation holds
            z = (counts[sel][ok] - n * p_pass[sel][ok]) / np.sqrt(var[ok])
            assert np.abs(z).max() < 4.8 and abs(z.mean()) < 4.0 / np.sqrt(len(z)) and 0.75 < z.std() < 1.25, (variant, np.abs(z).max(), z.mean(), z.std())
            rare = (var > 0) & ~ok                              # cells that (nearly) always or (nearly) never pass
            assert (np.abs(counts[sel][rare] - n * p_pass[sel][rare]) <= 5 * np.sqrt(var[rare]) + 3).all()
            assert np.array_equal(counts[sel][var == 0], n * p_pass[sel][var == 0])       # certain cells pass always, impossible ones never
            # independence of neighbouring lengths: joint passes of (len, len + 1) against the product of their probabilities
            grid = np.zeros((n, to), bool)
            grid[start, length] = True
            both = (grid[:, lo:-1] & grid[:, lo + 1:]).sum(0).astype(float)
            pj = p_pass[lo:-1] * p_pass[lo + 1:]
            vj = n * pj * (1 - pj)
            zj = (both[vj > 0] - n * pj[vj > 0]) / np.sqrt(vj[vj > 0])
            assert np.abs(zj).max() < 4.8 and abs(zj.mean()) < 4.0 / np.sqrt(len(zj)), (variant, np.abs(zj).max(), zj.mean())
            # probability_chosen | pass ~ U[thr1, 1)
            t1 = thr[0, length, 1]
            u = (pc - t1) / (1 - t1)
            assert u.min() >= 0 and u.max() < 1
            hist = np.bincount((u * 10).astype(int), minlength=10).astype(float)
            chi2 = ((hist - len(u) / 10) ** 2 / (len(u) / 10)).sum()
            assert chi2 < 35, (variant, chi2)                                             # 9 degrees of freedom
    finally:
        sim.close()
        oref.close()
        oprof.close()


# ------------------------------------------------------------------------------------------------------ indel draws
# The same idea for the third draw of FillReadPart (Simulator.cpp:322-326): indel ~ InDels(previous_indel_type, base_call) given
# {indel_pos, read_pos, gc_seq}.  Every conditioning value follows from the CIGAR and the bases of the read by the reference's own
# bookkeeping (:357-372 after a template base, :394-408 after a deletion, :421-439 after an insertion; initial values Simulator.h:232-234;
# gc_seq :479-500), restated below in numpy straight from those lines -- not from the oracle or the product.
def _indel_draws(results, rec):
    """per draw: table index, indel_pos, read_pos, gc_seq, outcome (0 none, 1 deletion, 2 + base insertion); template part only"""
    import re
    n = len(results)
    ops, seqs, read_len = [], [], np.zeros(n, np.int64)
    for i, (seq, _qual, cigar, _nerr, _tile) in enumerate(results):
        template_part = re.split("[SH]", cigar)[0]                    # the adapter part ('S') has its own bookkeeping; its count's digits are left over and ignored
        expanded = "".join(op * int(cnt) for cnt, op in re.findall(r"(\d+)([MID])", template_part))
        ops.append(expanded)
        seqs.append(np.frombuffer(seq, np.uint8))
        read_len[i] = len(seq)
    width = max(len(o) for o in ops)
    op = np.full((n, width), ord(" "), np.uint8)
    for i, o in enumerate(ops):
        op[i, :len(o)] = np.frombuffer(o.encode(), np.uint8)
    base = np.full((n, int(read_len.max()) + 1), 4, np.int64)
    for i, sq in enumerate(seqs):
        base[i, :len(sq)] = sq
    tmpl_len = rec["seqs"].shape[1]
    seq_length = np.minimum(read_len, tmpl_len)
    gc_count = np.array([int(np.isin(rec["seqs"][i, :seq_length[i]], (1, 2)).sum()) for i in range(n)])
    gc_seq = np.where(seq_length > 0, (gc_count * 100 + seq_length // 2) // np.maximum(seq_length, 1), 0)      # utilities::Percent
    read_pos, indel_pos, prev_type = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)
    base_call = np.full(n, 5, np.int64)
    element = np.full(n, ord("M"), np.uint8)
    rows = []
    for t in range(width):
        cur = op[:, t]
        act = cur != ord(" ")
        if not act.any():
            break
        is_m, is_d, is_i = act & (cur == ord("M")), act & (cur == ord("D")), act & (cur == ord("I"))
        outcome = np.where(is_d, 1, np.where(is_i, 2 + base[np.arange(n), np.minimum(read_pos, base.shape[1] - 1)], 0))
        rows.append(np.stack([prev_type * 6 + base_call, indel_pos, read_pos, gc_seq, outcome], axis=1)[act])
        # :357-372
        changed = is_m & (element != ord("M"))
        base_call = np.where(is_m, base[np.arange(n), np.minimum(read_pos, base.shape[1] - 1)], base_call)
        indel_pos = np.where(changed, 0, indel_pos)
        prev_type = np.where(changed, 0, prev_type)
        # :394-408
        cont_d, new_d = is_d & (element == ord("D")), is_d & (element != ord("D"))
        indel_pos = np.where(cont_d, indel_pos + 1, np.where(new_d, 1, indel_pos))
        prev_type = np.where(new_d, 1, prev_type)
        # :421-439
        cont_i, new_i = is_i & (element == ord("I")), is_i & (element != ord("I"))
        indel_pos = np.where(cont_i, indel_pos + 1, np.where(new_i, 1, indel_pos))
        prev_type = np.where(new_i, 0, prev_type)
        element = np.where(act, cur, element)
        read_pos = read_pos + (is_m | is_i)
    return np.concatenate(rows)


def _check_indels(results, rec, arrays, enforce=True):
    draws = _indel_draws(results, rec)
    worst, n_tests, n_rare = 0.0, 0, 0
    for table_id in np.unique(draws[:, 0]):
        a = draws[draws[:, 0] == table_id]
        table = _table(arrays, f"indels.{table_id // 6}.{table_id % 6}")
        if not len(table[0]):
            continue
        p, ok = _conditionals(table, a[:, 1:4])
        a = a[ok]
        col_of = {int(v): c for c, v in enumerate(table[0])}
        assert all(int(v) in col_of for v in np.unique(a[:, 4])), (table_id, "an outcome the table does not have")
        col = np.asarray([col_of[int(v)] for v in a[:, 4]])
        assert (p[np.arange(len(col)), col] > 0).all(), (table_id, "an outcome of probability 0 was drawn")
        n_rare += int((col > 0).sum())
        groups = [np.zeros(len(a), np.int64), np.minimum(a[:, 1], 3), a[:, 2] // 6, a[:, 3] // 10]      # all, indel position, read position, G/C
        for g in groups:
            z = _z_scores(p, col, g - g.min())
            if len(z):
                worst = max(worst, float(np.abs(z).max()))
                n_tests += len(z)
    if enforce:
        assert len(draws) > 1_500_000 and n_rare > 30_000 and n_tests > 300, (len(draws), n_rare, n_tests)
        assert worst < 5.5, worst
    return worst, n_tests, len(draws)


def _indel_profile(workdir):
    arrays = synth.make_profile(synth.TINY, seed=5)
    path = workdir / "stat_indel_profile.rsqp"
    write_container(path, arrays)
    return arrays, str(path)


def test_indel_conditionals_of_the_oracle(workdir):
    arrays, path = _indel_profile(workdir)
    rec = _records()
    prof = O.Profile(path)
    results = O.error_model_only(prof, SEED, rec)
    prof.close()
    assert _check_indels(results, rec, arrays)[0] < 5.5
    # the test of the test: indel-position rows shifted by one (the row of a running indel taken for the row of none) must fail loudly
    wrong = dict(arrays)
    for name in list(arrays):
        if name.startswith("tab.indels.") and name.endswith(".dim2"):
            par0, lim, margins = _table(arrays, name[4:-5])
            margins[0] = np.roll(margins[0], 1, axis=0)
            wrong[name] = np.concatenate([m.ravel() for m in margins])
    assert _check_indels(results, rec, wrong, enforce=False)[0] > 8


@pytest.mark.gpu
def test_indel_conditionals_of_the_product(workdir):
    from backends import GpuBackend
    arrays, path = _indel_profile(workdir)
    rec = _records()
    b = GpuBackend(path, None, 0)
    b.prepare(SEED)
    results = b.error_model(rec)
    b.close()
    assert _check_indels(results, rec, arrays)[0] < 5.5


# --------------------------------------------------------------------------------- sequence quality and read length draws
# FillRead's first two draws (Simulator.cpp:468-531): read_len ~ ReadLength(segment, fragment length) (Simulator.h:185-198: the counts of
# ReadLengthsByFragmentLength scanned from the longest read down) and seq_qual ~ SequenceQuality(segment, tile) given {G/C percent, mean
# systematic error rate, fragment length / 10} of the first min(read_len, template) template bases (:480-504, rounding utilities.hpp:450,552).
# The read length is an output.  The sequence quality is not, so the profile's quality tables are replaced by ones whose margin over the
# sequence quality is the identity and whose other margins are flat: every base of a read then carries seq_qual as its quality.
def _revealing_profile(workdir):
    arrays = synth.make_profile(synth.TINY, seed=5)
    values = np.arange(synth.TINY["qual_from"], synth.TINY["qual_to"], dtype=np.uint32)
    k = len(values)
    for seg in range(2):
        for tile in range(len(arrays["tiles.tiles"])):
            assert np.array_equal(np.sort(arrays[f"tab.seq_quality.{seg}.{tile}.par0"]), values)
            for base in range(4):
                prefix = f"tab.quality.{seg}.{tile}.{base}"
                arrays[prefix + ".par0"] = values
                arrays[prefix + ".limits"] = np.asarray([values[0], values[-1] + 1, 0, 1, 0, 1, 0, 1], arrays[prefix + ".limits"].dtype)
                arrays[prefix + ".dim2"] = np.concatenate([np.eye(k).ravel(), np.ones(3 * k)])
    path = workdir / "stat_revealing_profile.rsqp"
    write_container(path, arrays)
    return arrays, str(path)


def _percent(nom, den):
    return (nom * 100 + den // 2) // den                      # utilities::Percent / Divide: round half up


def _check_first_draws(results, rec, arrays, enforce=True):
    tiles = {int(t): i for i, t in enumerate(arrays["tiles.tiles"])}
    n = len(results)
    read_len = np.array([len(r[0]) for r in results])
    seq_qual = np.array([r[1][0] - 33 for r in results])
    assert all(len(set(r[1])) == 1 for r in results[:2000])                # every base carries the sequence quality
    seg, frag = rec["seg"].astype(np.int64), rec["frag_len"].astype(np.int64)
    tile = np.array([tiles[int(r[4])] if int(r[4]) in tiles else int(r[4]) for r in results])
    tmpl_len = rec["seqs"].shape[1]
    seq_length = np.minimum(read_len, tmpl_len)
    ar = np.arange(tmpl_len)[None, :] < seq_length[:, None]
    gc = _percent((np.isin(rec["seqs"], (1, 2)) & ar).sum(1), seq_length)
    mean_err = _percent_free_divide((rec["rate"].astype(np.int64) * ar).sum(1), seq_length)
    worst, n_tests = 0.0, 0
    # sequence quality: one table per (segment, tile)
    for s in range(2):
        for t in range(len(tiles)):
            sel = (seg == s) & (tile == t)
            table = _table(arrays, f"seq_quality.{s}.{t}")
            states = np.stack([gc[sel], mean_err[sel], frag[sel] // 10], axis=1)
            p, ok = _conditionals(table, states)
            assert ok.all()
            col_of = {int(v): c for c, v in enumerate(table[0])}
            col = np.asarray([col_of[int(v)] for v in seq_qual[sel]])
            assert (p[np.arange(len(col)), col] > 0).all()
            for g in (np.zeros(len(col), np.int64), states[:, 0] // 10, states[:, 1], states[:, 2] // 2):
                z = _z_scores(p, col, g - g.min())
                if len(z):
                    worst = max(worst, float(np.abs(z).max()))
                    n_tests += len(z)
    # read length: P(read_len | segment, fragment length) = count / sum of the fragment length's row
    for s in range(2):
        ptr, rfrom, vals = arrays[f"rl_by_fl.{s}.row_ptr"].astype(np.int64), arrays[f"rl_by_fl.{s}.row_from"].astype(np.int64), arrays[f"rl_by_fl.{s}.values"].astype(np.float64)
        first = int(arrays[f"rl_by_fl.{s}.from"][0])
        sel = np.flatnonzero(seg == s)
        rows = frag[sel] - first
        width = int((ptr[1:] - ptr[:-1]).max())
        lo = int(rfrom.min())
        p = np.zeros((len(sel), width + int(rfrom.max()) - lo))
        for i, r in enumerate(rows):
            v = vals[ptr[r]:ptr[r + 1]]
            p[i, rfrom[r] - lo:rfrom[r] - lo + len(v)] = v / v.sum()
        col = read_len[sel] - lo
        assert (col >= 0).all() and (p[np.arange(len(col)), col] > 0).all()
        for g in (np.zeros(len(col), np.int64), rows // 8):
            z = _z_scores(p, col, g - g.min())
            if len(z):
                worst = max(worst, float(np.abs(z).max()))
                n_tests += len(z)
    if enforce:
        assert n > 50_000 and n_tests > 150, (n, n_tests)
        assert worst < 5.5, worst
    return worst, n_tests


def _percent_free_divide(nom, den):
    return (nom + den // 2) // den                           # utilities::Divide


def _first_draw_records():
    arrays = synth.make_profile(synth.TINY, seed=5)
    rec = synth.make_error_model_input(78, N_READS, READ_LEN, arrays, zero_frac=0.5)
    return rec                                               # fragment lengths as drawn: short fragments end in adapter, the template totals cover min(read_len, fragment) bases


def test_sequence_quality_and_read_length_of_the_oracle(workdir):
    arrays, path = _revealing_profile(workdir)
    rec = _first_draw_records()
    rec["frag_len"][:] = np.maximum(rec["frag_len"], READ_LEN + 5)
    prof = O.Profile(path)
    O.lib().orc_profile_remove_indel_errors(prof.h)
    results = O.error_model_only(prof, SEED, rec)
    prof.close()
    assert _check_first_draws(results, rec, arrays)[0] < 5.5
    # the test of the test: G/C rows shifted by one
    wrong = dict(arrays)
    for name in list(arrays):
        if name.startswith("tab.seq_quality.") and name.endswith(".dim2"):
            par0, lim, margins = _table(arrays, name[4:-5])
            margins[1] = np.roll(margins[1], 2, axis=0)
            wrong[name] = np.concatenate([m.ravel() for m in margins])
    assert _check_first_draws(results, rec, wrong, enforce=False)[0] > 8


@pytest.mark.gpu
def test_sequence_quality_and_read_length_of_the_product(workdir):
    from backends import GpuBackend
    arrays, path = _revealing_profile(workdir)
    rec = _first_draw_records()
    rec["frag_len"][:] = np.maximum(rec["frag_len"], READ_LEN + 5)
    b = GpuBackend(path, None, 0, {"no_indels": True})
    b.prepare(SEED)
    results = b.error_model(rec)
    b.close()
    assert _check_first_draws(results, rec, arrays)[0] < 5.5


# -------------------------------------------------------------------------------------------------- fragment counts
# GetFragmentCounts (FragmentDistributionStats.cpp:3615-3627): the number of fragments of a chosen (start, length, strand) site is the negative-binomial
# quantile of a uniform on [thr0, 1), thr0 = the probability of no fragment at the LARGEST bias -- so for any site P(count = j | count >= 1) is the
# zero-truncated negative binomial with mean = bias * normalisation and r = Dispersion(mean) (:900-907, :3602-3613), bias = reference bias x length
# bias x G/C bias x surrounding(start) x surrounding(end) (Reference.h:167,283; Surrounding.h:114-120; SurroundingBase.hpp:64-81).  Evaluated here
# in numpy from the profile's arrays and the reference's bases for every site the sieve reports, and compared with the duplicates it reports.
def _site_counts(frags):
    key = np.stack([frags["seq"].astype(np.int64), frags["start"].astype(np.int64), frags["len"].astype(np.int64), frags["strand"].astype(np.int64)], axis=1)
    sites, inverse = np.unique(key, axis=0, return_inverse=True)
    counts = np.zeros(len(sites), np.int64)
    np.maximum.at(counts, inverse.ravel(), frags["dup"].astype(np.int64) + 1)
    assert np.array_equal(np.bincount(inverse.ravel(), minlength=len(sites)), counts)          # duplicates 0 .. count - 1, each once
    return sites, counts


def _kmer_bias(sur_bias, bases):
    """bases: (N, 30) codes, first base most significant within each block of ten (SurroundingBase.hpp:64-73)"""
    weights = 4 ** np.arange(9, -1, -1)
    total = np.zeros(len(bases))
    for b in (2, 1, 0):                                       # Surrounding.h:114-120 sums the blocks last to first
        total = total + sur_bias[b][(bases[:, 10 * b:10 * b + 10] * weights).sum(1)]
    return 2.0 / (1.0 + np.exp(-total))


def _check_fragment_counts(frags, arrays, seqs, bias_normalization, enforce=True, shift_gc=0, dispersion_scale=1.0):
    sites, counts = _site_counts(frags)
    sur_bias = arrays["frag.sur_bias"].reshape(3, -1)
    gc_bias, disp = arrays["frag.gc_bias"], arrays["frag.dispersion_parameters"] * dispersion_scale
    ilb, ilb_from = arrays["frag.insert_lengths_bias"], int(arrays["frag.insert_lengths_bias.from"][0])
    means, conds, cols = [], [], []
    for s, (_, codes) in enumerate(seqs):
        sel = sites[:, 0] == s
        if not sel.any():
            continue
        codes = codes.astype(np.int64)
        L = len(codes)
        start, length = sites[sel, 1], sites[sel, 2]
        end = start + length
        fwd = codes[(start[:, None] - 10 + np.arange(30)[None, :]) % L]
        rev = 3 - codes[(end[:, None] - 1 + 10 - np.arange(30)[None, :]) % L]          # reverse complement, walking down from end - 1 + 10
        prefix = np.concatenate([[0], np.cumsum(np.isin(codes, (1, 2)))])
        gc = np.minimum(_percent(prefix[end] - prefix[start], length) + shift_gc, 100)
        bias = arrays["frag.ref_seq_bias"][s] * ilb[length - ilb_from] * gc_bias[gc] * _kmer_bias(sur_bias, fwd) * _kmer_bias(sur_bias, rev)
        mean = bias * bias_normalization
        r = np.minimum(mean / (disp[0] + disp[1] * mean), mean * 1e10)
        p = mean / (mean + r)
        pmf0 = (1 - p) ** r
        pmf1 = pmf0 * p * r                                  # pc *= p * ((r - 1) / k + 1), k = 1
        pmf2 = pmf1 * p * ((r - 1) / 2 + 1)
        conds.append(np.stack([pmf1, pmf2, 1 - pmf0 - pmf1 - pmf2], axis=1) / (1 - pmf0)[:, None])      # counts 1, 2, 3 and more
        means.append(mean)
        cols.append(np.minimum(counts[sel], 3) - 1)
    mean, cond, col = np.concatenate(means), np.concatenate(conds), np.concatenate(cols)
    group = np.searchsorted(np.quantile(mean, np.linspace(0, 1, 9)[1:-1]), mean)         # eight groups of sites by their mean
    expected, observed, var = np.zeros((8, 3)), np.zeros((8, 3)), np.zeros((8, 3))
    np.add.at(expected, group, cond)
    np.add.at(var, group, cond * (1 - cond))
    np.add.at(observed, (group, col), 1.0)
    usable = var > 25
    z = (observed[usable] - expected[usable]) / np.sqrt(var[usable])
    if enforce:
        assert len(sites) > 40_000 and usable.sum() >= 16 and (counts > 1).sum() > 2_000, (len(sites), int(usable.sum()), int((counts > 1).sum()))
        assert np.abs(z).max() < 5.0, z
    return float(np.abs(z).max())


def _count_case(workdir):
    import parity_cases as P
    cfg = dict(synth.TINY, name="TINYd", dispersion=(0.5, 0.8))                   # sizeable over-dispersion: duplicates are common
    return cfg, P.make_inputs(workdir, "stat_counts", cfg, [40000, 25000], prof_seed=6, ref_seed=4)


def test_fragment_counts_of_the_oracle(workdir):
    cfg, (ppath, fpath, seqs) = _count_case(workdir)
    arrays = synth.make_profile(cfg, seed=6, n_ref_seqs=2)
    oprof, oref = O.Profile(ppath), O.Reference(seqs)
    sim = O.Sim(oprof, oref, 91, num_pairs=160_000)
    try:
        frags = sim.sieve(1, sim.total_blocks() + 1)
        assert _check_fragment_counts(frags, arrays, seqs, sim.bias_normalization()) < 5.0
        # the test of the test: with r proportional to the mean (small means) the truncated law hardly depends on the mean itself, but it does on the
        # dispersion parameters -- a fifth more must fail loudly; and a normalisation off by a factor of three shows as well
        assert _check_fragment_counts(frags, arrays, seqs, sim.bias_normalization(), enforce=False, dispersion_scale=1.2) > 8
        assert _check_fragment_counts(frags, arrays, seqs, sim.bias_normalization() * 3.0, enforce=False) > 8
    finally:
        sim.close()
        oref.close()
        oprof.close()


@pytest.mark.gpu
def test_fragment_counts_of_the_product(workdir):
    from backends import GpuBackend
    cfg, (ppath, fpath, seqs) = _count_case(workdir)
    arrays = synth.make_profile(cfg, seed=6, n_ref_seqs=2)
    b = GpuBackend(ppath, fpath)
    info = b.prepare(91, num_pairs=160_000)
    frags, _, _ = b.pairs(1, info["total_blocks"] + 1)
    b.close()
    assert _check_fragment_counts(frags, arrays, seqs, info["bias_normalization"]) < 5.0


# ------------------------------------------------------------------------------------------------------ the sieve: literal loop against gap draws
def _chi2_equal_exposure(a, b, min_total=10.0):
    """two count vectors from the SAME number of trials per bin (the same cells, another random stream): under one process a_i - b_i has mean 0 and variance about
    a_i + b_i (binomial counts of rare events), so sum (a - b)^2 / (a + b) is chi-square with one degree of freedom per bin -- totals included, unlike a test of
    homogeneity, which only compares shapes.  Bins with few events pooled into one.  (statistic, degrees of freedom)"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    small = (a + b) < min_total
    if small.any():
        a, b = np.append(a[~small], a[small].sum()), np.append(b[~small], b[small].sum())
    keep = (a + b) > 0
    return float(((a - b)[keep] ** 2 / (a + b)[keep]).sum()), int(keep.sum())


def _chi2_two_samples(a, b, min_expected=5.0):
    """homogeneity of two count vectors over the same bins (bins with small expectations pooled into one): (statistic, degrees of freedom)"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    A, B = a.sum(), b.sum()
    e_a, e_b = (a + b) * A / (A + B), (a + b) * B / (A + B)
    small = np.minimum(e_a, e_b) < min_expected
    if small.any():
        a, b = np.append(a[~small], a[small].sum()), np.append(b[~small], b[small].sum())
        e_a, e_b = (a + b) * A / (A + B), (a + b) * B / (A + B)
    keep = (a + b) > 0
    stat = ((a - e_a)[keep] ** 2 / e_a[keep] + (b - e_b)[keep] ** 2 / e_b[keep]).sum()
    return float(stat), int(keep.sum() - 1)


def test_gap_sieve_and_the_references_loop_are_two_samples_of_one_process(workdir):
    """Product and oracle draw the sieve's passing cells by their gaps (oracle_sim.c orc_gap_hits, rsq_kernels.h sieve_gaps); the reference draws one uniform per
    (start, fragment length) cell (Simulator.cpp:2302-2306).  orc_sieve_blocks_literal is that loop as written, on a random stream of its own, so for one seed the
    two routes are independent samples and bit parity between product and oracle says nothing about them being the same process -- this test does: over 1.4 * 10^7
    cells (8 seeds x 20 000 start positions x 89 lengths) the two routes must agree in (i) passing cells per fragment length, (ii) simulated sites per fragment length,
    (iii) sites with one and with two strands, (iv) (site, strand)s with 1, 2, ... pairs -- counts over the same cells, compared bin by bin with their totals
    (_chi2_equal_exposure) -- and (v) where probability_chosen lies within its range (homogeneity).  Negative controls: a gap run whose pass probabilities are 5 % off
    must fail (i), one whose strand thresholds are 5 % off must fail (iii) or (iv)."""
    from scipy.stats import chi2
    import parity_cases as P
    ppath, fpath, seqs = P.make_inputs(workdir, "two_samples", synth.TINY, [20000])
    oprof, oref = O.Profile(ppath), O.Reference(seqs)
    L = len(seqs[0][1])

    def sample(route, seeds, scale_pass=1.0, scale_strand=1.0):
        out = dict(passes=None, frags=None, strands=np.zeros(3), pairs=np.zeros(8), pc=np.zeros(10), cells=0)
        for seed in seeds:
            sim = O.Sim(oprof, oref, seed, num_pairs=30000)
            thr = sim.thresholds()
            to = thr.shape[1]
            if scale_pass != 1.0 or scale_strand != 1.0:
                thr = thr.copy()
                thr[0, :, 1] = 1 - np.minimum(1.0, scale_pass * (1 - thr[0, :, 1]))           # P(the cell passes) scaled
                thr[0, :, 0] = 1 - np.minimum(1.0, scale_strand * (1 - thr[0, :, 0]))         # P(a strand is non-zero) scaled
                sim.set_normalization(sim.bias_normalization(), thr)
            lo = int(sim.gap_passes(0, [])[2][0])
            if out["passes"] is None:
                out["passes"], out["frags"] = np.zeros(to), np.zeros(to)
            passes = sim.literal_passes(0, np.arange(L)) if route == "literal" else sim.gap_passes(0, np.arange(L))[0]
            length = np.array([p[1] for p in passes])
            pc = np.array([p[2] for p in passes])
            out["passes"] += np.bincount(length, minlength=to)
            t1 = thr[0, length, 1]
            u = (pc - t1) / np.where(t1 < 1, 1 - t1, 1.0)
            assert u.min() >= 0 and u.max() < 1
            out["pc"] += np.bincount((u * 10).astype(int), minlength=10)
            out["cells"] += L * (to - lo)
            fr = (sim.sieve_literal if route == "literal" else sim.sieve)(1, sim.total_blocks() + 1)
            site = fr["start"].astype(np.int64) * to + fr["len"]
            out["frags"] += np.bincount(np.unique(site) % to, minlength=to)       # simulated sites per length (pairs per site are compound: their own test below)
            per_strand = {}
            for s_, st_, d_ in zip(site.tolist(), fr["strand"].tolist(), fr["dup"].tolist()):
                per_strand[(s_, st_)] = max(per_strand.get((s_, st_), 0), d_ + 1)
            strands_of_site = {}
            for (s_, st_), n_ in per_strand.items():
                strands_of_site[s_] = strands_of_site.get(s_, 0) + 1
                out["pairs"][min(n_, 7)] += 1
            for n_ in strands_of_site.values():
                out["strands"][n_] += 1
            sim.close()
        return out

    lit, gap = sample("literal", range(100, 108)), sample("gap", range(100, 108))
    assert lit["cells"] == gap["cells"] > 1.4e7 and lit["passes"].sum() > 300_000 and lit["frags"].sum() > 100_000
    verdicts = {}
    for what in ("passes", "frags", "strands", "pairs", "pc"):
        stat, df = (_chi2_two_samples if what == "pc" else _chi2_equal_exposure)(lit[what], gap[what])
        verdicts[what] = (stat, df, chi2.sf(stat, df))
        assert stat < chi2.ppf(1 - 1e-5, df), (what, stat, df)                  # fixed seeds: the test is deterministic; 1e-5 is the margin against an unlucky choice of them
    print("two-sample verdicts (statistic, df, p):", {k: (round(v[0], 1), v[1], round(v[2], 4)) for k, v in verdicts.items()})
    assert all(df >= 1 for _, df, _ in verdicts.values()) and verdicts["passes"][1] > 60
    # negative controls: the same tests see a 5 % error
    off_pass = sample("gap", range(100, 108), scale_pass=1.05)
    stat, df = _chi2_equal_exposure(lit["passes"], off_pass["passes"])
    print("negative control, passes 5 % off:", round(stat, 1), df)
    assert stat > chi2.ppf(1 - 1e-9, df), ("passes, 5 % off", stat, df)
    off_strand = sample("gap", range(100, 108), scale_strand=1.05)
    stat_s, df_s = _chi2_equal_exposure(lit["strands"], off_strand["strands"])
    stat_p, df_p = _chi2_equal_exposure(lit["pairs"], off_strand["pairs"])
    print("negative control, strand thresholds 5 % off:", round(stat_s, 1), df_s, round(stat_p, 1), df_p)
    assert stat_s > chi2.ppf(1 - 1e-9, df_s) or stat_p > chi2.ppf(1 - 1e-9, df_p), ("strands / pairs, 5 % off", stat_s, df_s, stat_p, df_p)
    oref.close()
    oprof.close()
