"""Builds ReSeq profile *archives* (`.reseq` + `.reseq.ipf`) for the tests.

`from_rsqp` inverts what DataStats::PrepareProcessing / ProbabilityEstimates::PrepareResult do: from the named arrays of a synthetic
profile (reseq_amd/synth.py) it constructs a DataStats / ProbabilityEstimates value tree whose prepared form is exactly those arrays
again (result tables are already laid out as GetResults leaves them, so presenting their columns as the fit's dimension 0 in that
order reproduces them).  `rebin` then stores some tables bin-reduced and knocks out rows, which exercises FullExpansion and
ImputeMissingValues.  The trees go through oracle/reseq_archive.py's writer.
"""
import numpy as np

from oracle_archive import ra


def _vect(offset, values):
    return ra.vect(offset, [int(x) for x in values])


def _vect2(offset, rows):
    return {"vec_": (int(offset), rows)}


def _table_from_arrays(arrays, prefix):
    par0 = arrays[f"tab.{prefix}.par0"]
    lim = arrays[f"tab.{prefix}.limits"].reshape(-1, 2)
    flat = arrays[f"tab.{prefix}.dim2"]
    k, pos, margins = len(par0), 0, []
    for lo, hi in lim:
        n = (int(hi) - int(lo)) * k
        margins.append(flat[pos:pos + n].reshape(int(hi) - int(lo), k) if k else np.zeros((0, 0)))
        pos += n
    return par0, lim, margins


def ipf_from_table(par0, limits, margins, steps=37, precision=0.01):
    """A converged, un-binned LogIPF<N> (N = len(margins) + 1) whose PrepareResult is the given result table."""
    n_dims = len(margins) + 1
    n_margins = n_dims * (n_dims - 1) // 2
    k = len(par0)
    if k == 0:
        dims = [[] for _ in range(n_dims)]
        return dict(steps_=0, precision_=1.7976931348623157e308, estimates_=dict(dim2_=[[] for _ in range(n_margins)], dim_size_=[0] * n_dims),
                    dim_indices_=dims, initial_dim_indices_reduced_=[[] for _ in range(n_dims)], dim_indices_reduced_=[[] for _ in range(n_dims)],
                    margin_precision_=[1.7976931348623157e308] * n_margins, update_dist_=[2] * n_margins)
    dims = [[int(v) for v in par0]] + [list(range(int(lo), int(hi))) for lo, hi in limits]
    size = [len(d) for d in dims]
    dim2 = [np.ascontiguousarray(m).ravel() for m in margins]
    # the margins between two conditions: never read by the simulation
    a, b = n_dims, n_dims - 1
    others = {}
    for n in range(n_margins - 1, -1, -1):               # the (dim_a, dim_b) walk of LogArrayCalc::SetUp
        a -= 1
        if a == b:
            b -= 1
            a = n_dims - 1
        others[n] = (a, b)
    for n in range(n_dims - 1, n_margins):
        a, b = others[n]
        dim2.append(np.full(size[a] * size[b], 1.0))
    ident = [list(range(s)) for s in size]
    return dict(steps_=steps, needed_updates_=steps, precision_=precision, margin_precision_=[precision] * n_margins, last_margin_=1,
                last_update_=[steps] * n_margins, update_dist_=[2] * n_margins, estimates_=dict(dim2_=dim2, dim_size_=size),
                dim_indices_=dims, initial_dim_indices_reduced_=ident, dim_indices_reduced_=[list(x) for x in ident])


def rebin(ipf, rng, drop_rows=True):
    """Store the fit bin-reduced: neighbouring indices of every dimension share a bin (twice: initial and final reduction),
    the stored margins shrink to the bins.  Also removes some condition values from `dim_indices_`, which leaves all-zero rows
    that ImputeMissingValues has to fill."""
    n_dims = len(ipf["dim_indices_"])
    if not len(ipf["dim_indices_"][0]):
        return ipf
    out = dict(ipf)
    size = [len(d) for d in ipf["dim_indices_"]]
    initial, final = [], []
    for n in range(n_dims):
        # initial reduction: runs of 1-3 indices; final reduction: runs of 1-2 of those
        runs = []
        while sum(runs) < size[n]:
            runs.append(int(rng.integers(1, 4)))
        runs[-1] -= sum(runs) - size[n]
        init = [b for b, r in enumerate(runs) for _ in range(r)]
        runs2 = []
        while sum(runs2) < len(runs):
            runs2.append(int(rng.integers(1, 3)))
        runs2[-1] -= sum(runs2) - len(runs)
        fin = [b for b, r in enumerate(runs2) for _ in range(r)]
        initial.append(init)
        final.append(fin)
    bins = [max(f) + 1 for f in final]
    n_margins = n_dims * (n_dims - 1) // 2
    dim2 = []
    a, b = n_dims, n_dims - 1
    pairs = {}
    for n in range(n_margins - 1, -1, -1):
        a -= 1
        if a == b:
            b -= 1
            a = n_dims - 1
        pairs[n] = (a, b)
    for n in range(n_margins):
        a, b = pairs[n]
        dim2.append(np.exp(rng.normal(0.0, 0.5, size=bins[a] * bins[b])))
    out["estimates_"] = dict(dim2_=dim2, dim_size_=bins)
    out["initial_dim_indices_reduced_"] = initial
    out["dim_indices_reduced_"] = final
    if drop_rows:
        dims = [list(d) for d in ipf["dim_indices_"]]
        for n in range(1, n_dims):
            # spread the values of this condition out, so that values in between have no data
            step = rng.integers(1, 4, size=len(dims[n]))
            dims[n] = [int(dims[n][0] + s) for s in np.cumsum(step) - step[0]]
        out["dim_indices_"] = dims
    return out


def from_rsqp(arrays, creation_time=1600000000, rng=None):
    """(DataStats tree, ProbabilityEstimates tree) of a synthetic profile; with `rng` every second table is stored re-binned."""
    n_tiles = len(arrays["tiles.tiles"])
    st = {"creation_time_": creation_time}
    st["phred_quality_offset_"] = int(arrays["phred_quality_offset"][0])
    st["corrected_coverage_"] = float(arrays["corrected_coverage"][0])
    st["coverage_"] = {"reset_distance_": int(arrays["coverage.reset_distance"][0]), "coverage_threshold_": 10}
    max_del = int(arrays["errors.max_len_deletion"][0])
    indel_pos = [[ra.vect(0, []) for _ in range(6)] for _ in range(2)]
    # deletions of up to max_del bases were seen after an 'A' call; shorter ones elsewhere
    indel_pos[1][0] = _vect2(0, [_vect(0, [5, 1]) for _ in range(max_del)])
    indel_pos[1][2] = _vect2(0, [_vect(0, [9]) for _ in range(max(max_del - 1, 0))])
    indel_pos[0][1] = _vect2(0, [_vect(0, [4, 0, 1]) for _ in range(max_del + 3)])      # insertions do not count
    st["errors_"] = {"indel_by_indel_pos_": indel_pos}

    def vect_of(name, conv=int):
        return ra.vect(int(arrays[name + ".from"][0]), [conv(x) for x in arrays[name]])

    st["fragment_distribution_"] = {
        "insert_lengths_": vect_of("frag.insert_lengths"),
        "insert_lengths_bias_": vect_of("frag.insert_lengths_bias", float),
        "gc_fragment_content_bias_": vect_of("frag.gc_bias", float),
        "fragment_surroundings_bias_": {"bias_": list(np.asarray(arrays["frag.sur_bias"]).reshape(3, -1))},
        "fragment_surroundings_": {"counts_": [np.zeros(1 << 20, np.uint64) for _ in range(3)]},
        "dispersion_parameters_": [float(x) for x in arrays["frag.dispersion_parameters"]],
        "ref_seq_bias_": [float(x) for x in arrays["frag.ref_seq_bias"]],
        "abundance_": [7] * len(arrays["frag.ref_seq_bias"]),
    }
    rl, by_fl, non_mapped = [], [], []
    for seg in range(2):
        rl.append(vect_of(f"read_lengths.{seg}"))
        ptr, frm, vals = arrays[f"rl_by_fl.{seg}.row_ptr"], arrays[f"rl_by_fl.{seg}.row_from"], arrays[f"rl_by_fl.{seg}.values"]
        nm = arrays[f"rl_by_fl_nonmapped.{seg}.values"]
        rows = [_vect(frm[i], vals[ptr[i]:ptr[i + 1]]) for i in range(len(frm))]
        by_fl.append(_vect2(arrays[f"rl_by_fl.{seg}.from"][0], rows))
        # stored with another shape than the mapped ones: only rows that hold something, trimmed
        first = int(arrays[f"rl_by_fl.{seg}.from"][0])
        nm_rows = {}
        for i in range(len(frm)):
            row = nm[ptr[i]:ptr[i + 1]]
            nz = np.nonzero(row)[0]
            if len(nz):
                nm_rows[first + i] = _vect(int(frm[i]) + int(nz[0]), row[nz[0]:nz[-1] + 1])
        if nm_rows:
            lo, hi = min(nm_rows), max(nm_rows) + 1
            non_mapped.append(_vect2(lo, [nm_rows.get(i, ra.vect(0, [])) for i in range(lo, hi)]))
        else:
            non_mapped.append(_vect2(0, []))
    st["read_lengths_"], st["read_lengths_by_fragment_length_"], st["non_mapped_read_lengths_by_fragment_length_"] = rl, by_fl, non_mapped
    st["tiles_"] = {"tiles_": [int(x) for x in arrays["tiles.tiles"]], "abundance_": [int(x) for x in arrays["tiles.abundance"]]}

    seqs, cuts, n_ad = [], [], []
    for seg in range(2):
        codes, ptr = arrays[f"adapters.{seg}.seqs"], arrays[f"adapters.{seg}.seq_ptr"]
        seqs.append(["".join("ACGT"[c] for c in codes[ptr[i]:ptr[i + 1]]) for i in range(len(ptr) - 1)])
        cp, cf, cv = arrays[f"adapters.{seg}.start_cut_ptr"], arrays[f"adapters.{seg}.start_cut_from"], arrays[f"adapters.{seg}.start_cut"]
        cuts.append([_vect(cf[i], cv[cp[i]:cp[i + 1]]) for i in range(len(cf))])
        n_ad.append(len(ptr) - 1)
    assert n_ad[0] == n_ad[1] and np.array_equal(arrays["adapters.0.counts"], arrays["adapters.1.counts"])
    counts = []
    for a1 in range(n_ad[0]):
        row = []
        for a2 in range(n_ad[1]):
            if a1 == a2:
                # the detections of the pair, at the adapters' full lengths (beyond every shared prefix), plus detections at
                # length 0/0 that SumCounts must leave out whenever the adapter shares its first base with a neighbour
                full = _vect2(len(seqs[0][a1]), [_vect(len(seqs[1][a2]), [int(arrays["adapters.0.counts"][a1])])])
                row.append(full)
            else:
                row.append(_vect2(0, []))
        counts.append(row)
    st["adapters_"] = {
        "names_": [["adapter %d/%d" % (seg, i) for i in range(n_ad[seg])] for seg in range(2)],
        "combinations_": [[a1 == a2 for a2 in range(n_ad[1])] for a1 in range(n_ad[0])],
        "counts_": counts, "start_cut_": cuts, "seqs_archive": seqs,
        "polya_tail_length_": vect_of("adapters.polya_tail_length"),
        "overrun_bases_": [int(x) for x in arrays["adapters.overrun_bases"]],
    }

    flip = [0]

    def ipf(prefix):
        t = ipf_from_table(*_table_from_arrays(arrays, prefix))
        if rng is not None:
            flip[0] += 1
            if flip[0] % 2:
                t = rebin(t, rng)
        return t

    pe = {"stats_creation_time_": creation_time}
    pe["quality_"] = [[[ipf(f"quality.{s}.{t}.{b}") for b in range(4)] for t in range(n_tiles)] for s in range(2)]
    pe["sequence_quality_"] = [[ipf(f"seq_quality.{s}.{t}") for t in range(n_tiles)] for s in range(2)]
    pe["base_call_"] = [[[[ipf(f"base_call.{s}.{t}.{b}.{d}") for d in range(5)] for b in range(4)] for t in range(n_tiles)] for s in range(2)]
    pe["dom_error_"] = [[[ipf(f"dom_error.{b}.{p}.{d}") for d in range(5)] for p in range(5)] for b in range(4)]
    pe["error_rate_"] = [[ipf(f"error_rate.{b}.{d}") for d in range(5)] for b in range(4)]
    pe["indels_"] = [[ipf(f"indels.{t}.{c}") for c in range(6)] for t in range(2)]
    return st, pe


def write_profile_archives(stats_path, arrays, creation_time=1600000000, rng=None, ipf_path=None, grammar=None, trees=None):
    """`grammar`: an alternative of the doubtful token rules (oracle/reseq_archive.py all_grammars); `trees`: (DataStats, ProbabilityEstimates) made earlier"""
    st, pe = trees or from_rsqp(arrays, creation_time, rng)
    ra.write_archive(stats_path, "DataStats", st, grammar=grammar)
    ra.write_archive(ipf_path or str(stats_path) + ".ipf", "ProbabilityEstimates", pe, grammar=grammar)
    return st, pe
