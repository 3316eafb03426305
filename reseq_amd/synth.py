"""Synthetic ReSeq profiles, references and seqToIllumina inputs (numpy, seeded).

No fitted `.reseq`/`.reseq.ipf` profile and neither of the two test genomes
exist in the reference tree, so every workload in
BASELINE.json is restated on synthetic data of the same shape (SURVEY.md
§8(d)).  This module produces those inputs as RSQP containers
(container.py): the arrays carry the names of the reference members they stand
for, and the result tables are laid out exactly as
`LogArrayResult<N>::GetResults` leaves them (ProbabilityEstimates.h:386-453):
outcome columns sorted by ascending mean likelihood, conditioning axes
re-based to `value - from`.

It is host-side tooling for tests and bench.py; nothing here runs in the
simulation path.
"""
import numpy as np

from .container import write_container

SUR_RANGE = 10          # Surrounding::kRange      (Surrounding.h:17)
SUR_BLOCKS = 3          # Surrounding::kNumBlocks  (Surrounding.h:17)
SUR_SIZE = 1 << (2 * SUR_RANGE)


# --------------------------------------------------------------------------
# LogArrayResult layout
# --------------------------------------------------------------------------
def result_table(margins, values, limits):
    """Pack raw likelihood margins the way GetResults does.

    margins: list (one per conditioning axis) of arrays [rows_n, n_values]
             in VALUE order (column j <-> values[j]).
    values:  outcome values (dim_indices.at(0)).
    limits:  list of (from, to) per conditioning axis; rows_n == to - from.
    Returns dict(par0 u32[K], limits u32[NM,2], dim2 f64[sum rows_n*K]).
    """
    values = np.asarray(values, dtype=np.uint32)
    k = len(values)
    if k == 0:
        return dict(par0=np.zeros(0, np.uint32),
                    limits=np.zeros((len(limits), 2), np.uint32),
                    dim2=np.zeros(0, np.float64))
    # ProbabilityEstimates.h:397-408: order key = sum over margins of the
    # column mean; ascending sort, ties broken by the original index.
    key = np.zeros(k)
    for m in margins[::-1]:
        assert m.shape[1] == k
        key += m[::-1].sum(axis=0) / m.shape[0]
    order = np.lexsort((np.arange(k), key))
    par0 = values[order]
    lim = np.asarray(limits, dtype=np.uint32).reshape(len(limits), 2)
    parts = []
    for m, (lo, hi) in zip(margins, limits):
        assert m.shape[0] == hi - lo
        parts.append(np.ascontiguousarray(m[:, order], dtype=np.float64).ravel())
    return dict(par0=par0, limits=lim, dim2=np.concatenate(parts))


def ragged_table(tab, j):
    """The table with rows cut off the ends of its conditioning ranges (by how many depends on the table's number j), as the tables of a real profile
    differ in the value ranges they were fitted on (ProbabilityEstimates.h:386-396: limits_ are per table); draws outside a table's own range take its edge rows."""
    k = len(tab["par0"])
    if k == 0:
        return tab
    lim, parts, at = tab["limits"].copy(), [], 0
    cut = np.random.default_rng(4000 + j).integers(0, 4, size=(len(lim), 2))
    for n, (lo, hi) in enumerate(tab["limits"].tolist()):
        rows = hi - lo
        m = tab["dim2"][at:at + rows * k].reshape(rows, k)
        at += rows * k
        front = int(cut[n, 0]) if rows > 6 else 0
        back = int(cut[n, 1]) if rows - front > 6 else 0
        parts.append(m[front:rows - back].ravel())
        lim[n] = (lo + front, hi - back)
    return dict(par0=tab["par0"], limits=lim, dim2=np.concatenate(parts))


def _noise(rng, shape, sigma=0.05):
    return np.exp(rng.normal(0.0, sigma, size=shape))


# --------------------------------------------------------------------------
# Table families (ProbabilityEstimates.h:1312-1325)
# --------------------------------------------------------------------------
def _quality_table(rng, cfg, seg, base):
    q = np.arange(cfg["qual_from"], cfg["qual_to"], dtype=np.float64)     # outcome values
    nq = len(q)
    rl = cfg["read_len_max"]
    top = q[-1]
    sq = np.arange(cfg["qual_from"], cfg["qual_to"], dtype=np.float64)[:, None]
    pq = np.arange(cfg["qual_from"], cfg["qual_to"], dtype=np.float64)[:, None]
    pos = np.arange(rl, dtype=np.float64)[:, None]
    er = np.arange(0, 101, dtype=np.float64)[:, None]
    spread = max(2.0, nq / 7.0)
    prior = np.exp(4.5 * (q[None, :] - top) / nq)          # Illumina base qualities pile up at the top
    t0 = 100.0 * (np.exp(-0.5 * ((q[None, :] - (sq + 1.0 - 0.5 * seg)) / spread) ** 2) + 1e-3) * prior   # margin scale is arbitrary in IPF
    t1 = np.exp(-0.5 * ((q[None, :] - pq) / (0.5 * spread)) ** 2) + 2e-2
    t2 = np.exp(-(pos / rl) * (0.03 + 0.01 * seg) * (q[None, :] - top) * 40.0 / nq)
    t3 = np.exp(-(er / 100.0) * 0.1 * (q[None, :] - top) * 40.0 / nq)
    t0 *= _noise(rng, t0.shape) * (1.0 + 0.02 * base)
    t1 *= _noise(rng, t1.shape)
    t2 *= _noise(rng, t2.shape, 0.02)
    t3 *= _noise(rng, t3.shape, 0.02)
    lim = [(cfg["qual_from"], cfg["qual_to"]), (cfg["qual_from"], cfg["qual_to"]), (0, rl), (0, 101)]
    return result_table([t0, t1, t2, t3], q.astype(np.uint32), lim)


def _seq_quality_table(rng, cfg, seg):
    q = np.arange(cfg["qual_from"], cfg["qual_to"], dtype=np.float64)
    nq = len(q)
    centre = q[0] + 0.8 * (nq - 1) - seg
    gc = np.arange(cfg["sq_gc_from"], cfg["sq_gc_to"], dtype=np.float64)[:, None]
    me = np.arange(0, cfg["sq_err_to"], dtype=np.float64)[:, None]
    fl = np.arange(cfg["ins_from"] // 10, (cfg["ins_to"] - 1) // 10 + 1, dtype=np.float64)[:, None]
    spread = max(1.5, nq / 10.0)
    t0 = np.exp(-0.5 * ((q[None, :] - (centre - 0.0006 * (gc - 50.0) ** 2 * nq / 40.0)) / spread) ** 2) + 1e-4
    t1 = np.exp(-0.5 * ((q[None, :] - (centre - 0.5 * me)) / (1.5 * spread)) ** 2) + 1e-4
    t2 = np.exp(-0.5 * ((q[None, :] - (centre - 0.01 * fl)) / (2.0 * spread)) ** 2) + 1e-4
    for t in (t0, t1, t2):
        t *= _noise(rng, t.shape)
    lim = [(cfg["sq_gc_from"], cfg["sq_gc_to"]), (0, cfg["sq_err_to"]),
           (cfg["ins_from"] // 10, (cfg["ins_to"] - 1) // 10 + 1)]
    return result_table([t0, t1, t2], q.astype(np.uint32), lim)


def _base_call_table(rng, cfg, seg, ref, dom):
    calls = np.arange(5)
    rl = cfg["read_len_max"]
    qv = np.arange(cfg["qual_from"], cfg["qual_to"], dtype=np.float64)
    e = np.minimum(0.75, cfg["err_scale"] * 10.0 ** (-qv / 10.0))[:, None]
    w = np.ones(5)
    w[ref] = 0.0
    w[4] = 0.02
    if dom < 4 and dom != ref:
        w[dom] = 3.0
    w = w / w.sum()
    t0 = e * w[None, :]
    t0[:, ref] = 1.0 - e[:, 0]
    t0[0, 4] += 0.5 if cfg["qual_from"] <= 2 else 0.0         # lowest quality: many N calls
    is_err = (calls != ref).astype(np.float64)[None, :]
    pos = np.arange(rl, dtype=np.float64)[:, None]
    ne = np.arange(cfg["bc_nerr_to"], dtype=np.float64)[:, None]
    er = np.arange(0, 101, dtype=np.float64)[:, None]
    t1 = 1.0 + is_err * (0.5 + 0.3 * seg) * pos / rl
    t2 = 1.0 + is_err * 0.15 * ne
    t3 = np.ones((101, 5))
    if dom < 4 and dom != ref:
        t3[:, dom] = 1.0 + 3.0 * er[:, 0]
    t3[:, ref] = 1.0 - 0.009 * er[:, 0]
    for t in (t0, t1, t2, t3):
        t *= _noise(rng, t.shape, 0.02)
    lim = [(cfg["qual_from"], cfg["qual_to"]), (0, rl), (0, cfg["bc_nerr_to"]), (0, 101)]
    return result_table([t0, t1, t2, t3], calls, lim)


def _indel_table(rng, cfg, prev_type, last_call):
    vals = np.arange(6)          # kNoInDel, kDeletion, kInsertionA..T (ErrorStats.h:17-22)
    rl = cfg["read_len_max"]
    ip = np.arange(cfg["indel_pos_to"], dtype=np.float64)[:, None]
    t0 = np.ones((cfg["indel_pos_to"], 6))
    t0[:, 1] = cfg["del_rate"]
    t0[:, 2:] = cfg["ins_rate"] / 4.0
    if last_call < 4:
        t0[:, 2 + last_call] *= 3.0
    if prev_type == 1:
        t0[1:, 1] = 0.25 / ip[1:, 0]             # deletion extension
    else:
        t0[1:, 2:] = 0.05 / ip[1:, :]            # insertion extension
    pos = np.arange(rl, dtype=np.float64)[:, None]
    gc = np.arange(0, 101, dtype=np.float64)[:, None]
    t1 = np.ones((rl, 6))
    t1[:, 1:] = 1.0 + 0.5 * pos / rl
    t2 = np.ones((101, 6))
    t2[:, 1:] = 1.0 + 0.004 * np.abs(gc - 50.0)
    for t in (t0, t1, t2):
        t *= _noise(rng, t.shape, 0.02)
    # the scale of a margin's column is arbitrary (only the products matter) but decides the column order of result_table: with
    # cfg["indel_columns_shuffled"] two insertion columns sort above "no indel", one with a sizeable rate
    if cfg.get("indel_columns_shuffled"):
        t0[:, 2] *= 300.0
        t0[:, 2:4] /= 4.0
        t1[:, 2:4] *= 4.0
    lim = [(0, cfg["indel_pos_to"]), (0, rl), (0, 101)]
    return result_table([t0, t1, t2], vals, lim)


def _sys_limits(cfg):
    dist_to = (cfg["reset_distance"] - 1 + 9) // 10 + 1     # CoverageStats.cpp:683 max_error_dist
    return [(0, dist_to), (0, 101), (0, 101)]


def _dom_error_table(rng, cfg, ref, prev, dom5):
    vals = np.arange(5)
    lim = _sys_limits(cfg)
    w = np.full(5, cfg["sys_rate"] / 3.0)
    w[ref] = 0.0
    w[4] = 1.0 - cfg["sys_rate"]
    if prev < 4 and prev != ref:
        w[prev] *= 2.0
    if dom5 != ref:
        w[dom5] *= 1.5
    d = np.arange(lim[0][1], dtype=np.float64)[:, None]
    sr = np.arange(0, 101, dtype=np.float64)[:, None]
    t0 = np.tile(w, (lim[0][1], 1))
    t0[1:, :4] *= 1.0 + 4.0 / d[1:]                   # inside an error region errors cluster
    t1 = np.ones((101, 5)) * _noise(rng, (101, 5), 0.02)
    t2 = np.ones((101, 5))
    t2[:, :4] = 1.0 + 0.05 * sr
    t0 = t0 * _noise(rng, t0.shape, 0.02)
    t0[:, ref] = 0.0
    return result_table([t0, t1, t2], vals, lim)


def _error_rate_table(rng, cfg, ref, dom_err):
    k = cfg["err_rate_to"]
    vals = np.arange(k)
    lim = _sys_limits(cfg)
    r = np.arange(k, dtype=np.float64)
    if dom_err == 4 or dom_err == ref:
        w = np.full(k, 1e-6)
        w[0] = 1.0
    else:
        w = np.exp(-r / 8.0)
        w[0] = 0.02
    sr = np.arange(0, 101, dtype=np.float64)[:, None]
    t0 = np.tile(w, (lim[0][1], 1)) * _noise(rng, (lim[0][1], k), 0.02)
    t1 = np.ones((101, k)) * _noise(rng, (101, k), 0.02)
    t2 = np.exp(-np.abs(r[None, :] - 0.6 * sr) / 12.0) + 0.05
    t2[0, :] = 1.0
    return result_table([t0, t1, t2], vals, lim)


# --------------------------------------------------------------------------
# Surrounding bias (Surrounding.cpp:194-219 CombinePositions)
# --------------------------------------------------------------------------
def combine_positions(separated):
    """separated[4*30] -> bias[3][4^10]; k-mer code is big-endian base-4."""
    separated = np.asarray(separated, dtype=np.float64).reshape(SUR_BLOCKS, SUR_RANGE, 4)
    out = np.zeros((SUR_BLOCKS, SUR_SIZE))
    codes = np.arange(SUR_SIZE)
    for b in range(SUR_BLOCKS):
        acc = np.zeros(SUR_SIZE)
        for pos in range(SUR_RANGE):           # same summation order as the reference's pos loop
            base = (codes >> (2 * (SUR_RANGE - 1 - pos))) & 3
            acc = acc + separated[b, pos, base]
        out[b] = acc
    return out


# --------------------------------------------------------------------------
# Profile
# --------------------------------------------------------------------------
P0 = dict(
    name="P0", read_len_max=150, read_len_var=False,
    qual_from=2, qual_to=42, phred_offset=33,
    ins_from=50, ins_to=1000, ins_mu=350.0, ins_sigma=0.25, ins_total=2_000_000,
    adapter_only=0,
    tiles=[1101], tile_abundance=[1],
    n_adapters=2, adapter_len=58,
    sq_gc_from=15, sq_gc_to=86, sq_err_to=12, bc_nerr_to=24, indel_pos_to=8,
    err_scale=1.0, del_rate=2e-5, ins_rate=1.5e-5, sys_rate=0.03, err_rate_to=101,
    reset_distance=150, max_len_deletion=4,
    dispersion=(0.04, 0.3), sur_sigma=0.12, corrected_coverage=30.0,
)

def p0_with_tiles(n_tiles):
    """P0 with `n_tiles` tiles of unequal abundance (reseq illuminaPE --tiles, main.cpp:277-283: per-tile tables,
    ProbabilityEstimates.h:1497-1514); HiSeq numbering: surface, swath, tile"""
    tiles = [1101 + (i % 16) + 100 * ((i // 16) % 3) + 1000 * (i // 48) for i in range(n_tiles)]
    return dict(P0, name=f"P0t{n_tiles}", tiles=tiles, tile_abundance=[1 + (i * 5) % 7 for i in range(n_tiles)])


TINY = dict(
    name="TINY", read_len_max=30, read_len_var=True,
    qual_from=2, qual_to=12, phred_offset=33,
    ins_from=12, ins_to=90, ins_mu=45.0, ins_sigma=0.3, ins_total=200_000,
    adapter_only=150,
    tiles=[1101, 1102, 2308], tile_abundance=[5, 3, 2],
    n_adapters=3, adapter_len=9,
    sq_gc_from=10, sq_gc_to=91, sq_err_to=20, bc_nerr_to=8, indel_pos_to=4,
    err_scale=1.5, del_rate=1.5e-2, ins_rate=1.2e-2, sys_rate=0.15, err_rate_to=101,
    reset_distance=30, max_len_deletion=3,
    dispersion=(0.2, 0.4), sur_sigma=0.3, corrected_coverage=8.0,
)


def _vect(out, name, values, offset, dtype):
    out[name] = np.asarray(values, dtype=dtype)
    out[name + ".from"] = np.asarray([offset], dtype=np.uint64)


def make_profile(cfg=None, seed=103741084, n_ref_seqs=1):
    """Return the dict of named arrays of a synthetic profile."""
    cfg = dict(P0 if cfg is None else cfg)
    rng = np.random.default_rng(seed)
    out = {}
    rl = cfg["read_len_max"]
    nt = len(cfg["tiles"])

    out["phred_quality_offset"] = np.asarray([cfg["phred_offset"]], np.uint8)
    out["corrected_coverage"] = np.asarray([cfg["corrected_coverage"]], np.float64)
    out["errors.max_len_deletion"] = np.asarray([cfg["max_len_deletion"]], np.uint16)
    out["coverage.reset_distance"] = np.asarray([cfg["reset_distance"]], np.uint32)

    # insert lengths (FragmentDistributionStats.h:441-442)
    lens = np.arange(cfg["ins_from"], cfg["ins_to"])
    pdf = np.exp(-0.5 * ((np.log(lens) - np.log(cfg["ins_mu"])) / cfg["ins_sigma"]) ** 2) / lens
    counts = np.floor(pdf / pdf.sum() * cfg["ins_total"]).astype(np.uint64)
    if cfg["adapter_only"]:
        full = np.zeros(cfg["ins_to"], np.uint64)
        full[cfg["ins_from"]:] = counts
        full[0] = cfg["adapter_only"]
        _vect(out, "frag.insert_lengths", full, 0, np.uint64)
        ilb = np.zeros(cfg["ins_to"])
        ilb[cfg["ins_from"]:] = counts / counts.max()
        _vect(out, "frag.insert_lengths_bias", ilb, 0, np.float64)
    else:
        _vect(out, "frag.insert_lengths", counts, cfg["ins_from"], np.uint64)
        _vect(out, "frag.insert_lengths_bias", counts / counts.max(), cfg["ins_from"], np.float64)
    gc = np.arange(101)
    gcb = 0.15 + 0.85 * np.exp(-0.5 * ((gc - 48.0) / 18.0) ** 2)
    _vect(out, "frag.gc_bias", gcb * _noise(rng, 101, 0.01), 0, np.float64)
    sep = rng.normal(0.0, cfg["sur_sigma"], size=4 * SUR_BLOCKS * SUR_RANGE)
    out["frag.sur_bias_separated"] = sep
    out["frag.sur_bias"] = combine_positions(sep)
    out["frag.dispersion_parameters"] = np.asarray(cfg["dispersion"], np.float64)
    out["frag.ref_seq_bias"] = np.ones(n_ref_seqs)

    # read lengths (DataStats.h:196-199) and read lengths by fragment length (rows = fragment lengths, CSR)
    n_reads = int(counts.sum())
    ins = out["frag.insert_lengths"]
    ioff = int(out["frag.insert_lengths.from"][0])
    for seg in range(2):
        if cfg["read_len_var"]:
            rls = np.arange(rl - 6, rl + 1)
            w = np.asarray([1, 1, 2, 3, 5, 8, 80], dtype=np.float64)
        else:
            rls = np.arange(rl, rl + 1)
            w = np.ones(1)
        rc = np.floor(w / w.sum() * n_reads).astype(np.uint64)
        _vect(out, f"read_lengths.{seg}", rc, rls[0], np.uint64)
        row_ptr, row_from, vals = [0], [], []
        for fl in range(ioff, ioff + len(ins)):
            c = int(ins[fl - ioff])
            row = np.floor(w / w.sum() * c).astype(np.uint64)
            row[-1] += np.uint64(c - int(row.sum()))
            row_from.append(rls[0])
            vals.append(row)
            row_ptr.append(row_ptr[-1] + len(row))
        out[f"rl_by_fl.{seg}.from"] = np.asarray([ioff], np.uint64)
        out[f"rl_by_fl.{seg}.row_ptr"] = np.asarray(row_ptr, np.uint32)
        out[f"rl_by_fl.{seg}.row_from"] = np.asarray(row_from, np.uint32)
        out[f"rl_by_fl.{seg}.values"] = np.concatenate(vals).astype(np.uint64)
        out[f"rl_by_fl_nonmapped.{seg}.values"] = np.zeros(row_ptr[-1], np.uint64)

    out["tiles.tiles"] = np.asarray(cfg["tiles"], np.uint16)
    out["tiles.abundance"] = np.asarray(cfg["tile_abundance"], np.uint64)

    # adapters (AdapterStats.h:101-107)
    for seg in range(2):
        seqs = [rng.integers(0, 4, size=cfg["adapter_len"] + 3 * i, dtype=np.uint8)
                for i in range(cfg["n_adapters"])]
        out[f"adapters.{seg}.seqs"] = np.concatenate(seqs)
        out[f"adapters.{seg}.seq_ptr"] = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint32)
        cnt = np.asarray([1000 // (i + 1) for i in range(cfg["n_adapters"])], np.uint64)
        out[f"adapters.{seg}.counts"] = cnt
        sig = cnt.copy()
        sig[cnt < np.ceil(cnt.max() * 0.1)] = 0             # AdapterStats::PrepareSimulation
        out[f"adapters.{seg}.significant_counts"] = sig
        cut_ptr, cut_from, cut_vals = [0], [], []
        for i in range(cfg["n_adapters"]):
            v = np.asarray([50, 20, 10, 5][: 2 + i % 3], np.uint64)
            cut_from.append(i % 2)
            cut_vals.append(v)
            cut_ptr.append(cut_ptr[-1] + len(v))
        out[f"adapters.{seg}.start_cut_ptr"] = np.asarray(cut_ptr, np.uint32)
        out[f"adapters.{seg}.start_cut_from"] = np.asarray(cut_from, np.uint32)
        out[f"adapters.{seg}.start_cut"] = np.concatenate(cut_vals)
    _vect(out, "adapters.polya_tail_length", [40, 20, 10, 6, 3, 1], 0, np.uint64)
    out["adapters.overrun_bases"] = np.asarray([300, 220, 240, 260, 7], np.uint64)

    # result tables
    def put(prefix, tab):
        for k, v in tab.items():
            out[f"tab.{prefix}.{k}"] = v

    # ragged_tables: every table of the read kernel's three families over its own value ranges, and an empty table in each of two families
    ragged = (lambda tab, j: ragged_table(tab, j)) if cfg.get("ragged_tables") else (lambda tab, j: tab)
    empty = (lambda tab: result_table([], [], tab["limits"].tolist())) if cfg.get("ragged_tables") else (lambda tab: tab)
    j = 0
    for seg in range(2):
        for tile in range(nt):
            put(f"seq_quality.{seg}.{tile}", _seq_quality_table(rng, cfg, seg))
            for base in range(4):
                put(f"quality.{seg}.{tile}.{base}", ragged(_quality_table(rng, cfg, seg, base), j := j + 1))
                for dom in range(5):
                    tab = ragged(_base_call_table(rng, cfg, seg, base, dom), j := j + 1)
                    put(f"base_call.{seg}.{tile}.{base}.{dom}", empty(tab) if (seg, tile, base, dom) == (1, nt - 1, 2, 4) else tab)
    for base in range(4):
        for prev in range(5):
            for dom5 in range(4):
                put(f"dom_error.{base}.{prev}.{dom5}", _dom_error_table(rng, cfg, base, prev, dom5))
            # dom_last5 == 4 never occurs for an N-free sequence: keep the table empty,
            # which exercises the prob_sum == 0 fall-backs (Simulator.h:341-348)
            put(f"dom_error.{base}.{prev}.4", result_table([], [], _sys_limits(cfg)))
        for dom_err in range(5):
            put(f"error_rate.{base}.{dom_err}", _error_rate_table(rng, cfg, base, dom_err))
    for prev_type in range(2):
        for last_call in range(6):
            tab = ragged(_indel_table(rng, cfg, prev_type, last_call), j := j + 1)
            put(f"indels.{prev_type}.{last_call}", empty(tab) if (prev_type, last_call) == (1, 4) else tab)
    return out


def write_profile(path, arrays):
    write_container(path, arrays)


# --------------------------------------------------------------------------
# References and seqToIllumina inputs
# --------------------------------------------------------------------------
def make_reference(seed, lengths, gc=0.5, names=None):
    """List of (name, uint8 codes A=0,C=1,G=2,T=3) with i.i.d. bases."""
    rng = np.random.default_rng(seed)
    p = np.asarray([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    out = []
    for i, n in enumerate(lengths):
        name = names[i] if names else f"synth{i + 1} len={n}"
        out.append((name, rng.choice(4, size=n, p=p).astype(np.uint8)))
    return out


def write_fasta(path, seqs, width=80):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for name, codes in seqs:
            f.write(b">" + name.encode() + b"\n")
            txt = lut[codes]
            for i in range(0, len(txt), width):
                f.write(txt[i:i + width].tobytes() + b"\n")


def encode_sys_rate(rate):
    """Error percent -> FASTQ-safe value (Simulator.cpp:2569-2575)."""
    rate = np.asarray(rate, dtype=np.int64)
    q = np.where(rate > 86, rate - (rate - 85) // 2, rate)
    return (q + 33).astype(np.uint8)


def fixed_width_fasta(rec):
    """seqToIllumina's input for the records as a byte matrix, one row per record: ">r" 9 digits " " segment ";" 3 digits ";" dominant errors ";" rates,
    a line end, the bases, a line end (fragment lengths 100..999; number_rows writes the ids).  The measuring tools write files of any size from it."""
    n, L = rec["seqs"].shape
    fl = np.asarray(rec["frag_len"], np.int64)
    assert fl.min() >= 100 and fl.max() <= 999
    block = np.zeros((n, 21 + 3 * L), np.uint8)
    block[:, 0:2] = np.frombuffer(b">r", np.uint8)
    block[:, 11] = ord(" ")
    block[:, 12] = np.asarray(rec["seg"], np.uint8) + ord("1")
    block[:, 13] = ord(";")
    for k in range(3):
        block[:, 14 + k] = (fl // 10 ** (2 - k)) % 10 + ord("0")
    block[:, 17] = ord(";")
    block[:, 18:18 + L] = np.frombuffer(b"ACGTN", np.uint8)[rec["dom"]]
    block[:, 18 + L] = ord(";")
    block[:, 19 + L:19 + 2 * L] = encode_sys_rate(rec["rate"])
    block[:, 19 + 2 * L] = ord("\n")
    block[:, 20 + 2 * L:20 + 3 * L] = np.frombuffer(b"ACGT", np.uint8)[rec["seqs"]]
    block[:, 20 + 3 * L] = ord("\n")
    return block


def number_rows(block, first):
    """ids "r{first + row:09d}" into the rows of fixed_width_fasta"""
    idx = np.arange(first, first + len(block))
    for k in range(9):
        block[:, 2 + k] = (idx // 10 ** (8 - k)) % 10 + ord("0")


def make_error_model_input(seed, n, read_len, profile, zero_frac=0.97):
    """Records for seqToIllumina (Simulator.cpp:2403-2512): per record a
    template, the segment, fragment length, dominant-error bases and rates."""
    rng = np.random.default_rng(seed)
    seqs = rng.integers(0, 4, size=(n, read_len), dtype=np.uint8)
    seg = (np.arange(n) % 2).astype(np.uint8)
    ins = profile["frag.insert_lengths"].astype(np.float64)
    off = int(profile["frag.insert_lengths.from"][0])
    ins = ins.copy()
    if off == 0:
        ins[0] = 0
    frag = (rng.choice(len(ins), size=n, p=ins / ins.sum()) + off).astype(np.uint32)
    dom = rng.choice(5, size=(n, read_len), p=[0.01, 0.01, 0.01, 0.01, 0.96]).astype(np.uint8)
    rate = np.where(rng.random((n, read_len)) < zero_frac, 0,
                    rng.integers(1, 61, size=(n, read_len))).astype(np.uint8)
    return dict(seqs=seqs, seg=seg, frag_len=frag, dom=dom, rate=rate)
