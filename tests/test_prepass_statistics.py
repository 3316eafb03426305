"""The pre-passes against numpy restatements of the reference's statements -- checks that do NOT go through the oracle's reading (missing #5).

a13  SetSystematicErrors (Simulator.h:337-382): from a strand's finished track and its bases every conditioning value of every position's two draws is replayed --
     previous base, dominant base of the last five (utilities.hpp:229-298), G/C percent of the last sys_gc_range bases (Simulator.h:360-371), and the error region's
     (distance, start rate) by CoverageStats::UpdateDistances (CoverageStats.cpp:379-396) -- and the observed dominant errors and error rates are compared with the
     conditionals of the tables DominantError(ref, last, dom) and ErrorRate(ref, dominant error) (z scores per table and outcome; a shifted margin must be noticed).
a14  CalculateBiasNormalization (FragmentDistributionStats.cpp:3504-3582, Reference::SumBias Reference.cpp:622-659): for sampled fragment lengths the sum and the
     maximum of the bias over all start positions, recomputed from bases and tables, give normalization_by_frag_len, the thresholds
     (CalculateNonZeroThreshold :2969-2976: thr0 = (r / (r + m))^r, thr1 = thr0^(2 alleles)) and bias_normalization = total pairs / (2 sum of the norms).
a16  CTConversion (Simulator.cpp:1925-2002): with errors switched off a read is its template, so the share of template C that reads T inside a region of the BED file
     is 1 - methylation of the region (binomial z score per region, both strands; outside the regions nothing converts).

For the oracle and the host emulation on the CPU, for the device under `-m gpu`."""
import numpy as np
import pytest

import parity_cases as P
from backends import EmuBackend, GpuBackend
from reseq_amd import synth
from test_statistics import _conditionals, _kmer_bias, _percent, _table, _z_scores


# ---------------------------------------------------------------------------------------------------------------- a13
def _dominant_of_last_five(codes):
    """DominantBase::FindDominant for every position >= 5: the most frequent base among the five before it, ties to the one nearest to the position"""
    n = len(codes)
    last = np.stack([codes[5 - k - 1:n - k - 1] for k in range(5)], axis=1)          # column k: the base k + 1 positions back, for positions 5 .. n - 1
    counts = np.stack([(last == b).sum(1) for b in range(4)], axis=1)
    best = counts.max(1)
    dom = np.full(n, -1)
    chosen = np.full(n - 5, -1)
    for k in range(5):                                                              # the nearest base whose count is the maximum
        hit = (chosen < 0) & (counts[np.arange(n - 5), last[:, k]] == best)
        chosen[hit] = last[hit, k]
    dom[5:] = chosen
    return dom


def _chain_states(codes, rate, gc_range, reset_distance):
    """(distance feature, G/C percent, start rate) in front of every position's draws, and which positions are far enough from the chain's start to be replayed"""
    n = len(codes)
    is_gc = np.isin(codes, (1, 2)).astype(np.int64)
    prefix = np.concatenate([[0], np.cumsum(is_gc)])
    pos = np.arange(n)
    lo = np.maximum(pos - gc_range, 0)
    bases = pos - lo
    gc = np.where(bases > 0, (100 * (prefix[pos] - prefix[lo]) + bases // 2) // np.maximum(bases, 1), 50)      # SafePercent, Divide with rounding
    dist, start = np.zeros(n, np.int64), np.zeros(n, np.int64)
    d = s = 0
    for i in range(n):                                                              # UpdateDistances, CoverageStats.cpp:379-396
        dist[i], start[i] = d, s
        r = int(rate[i])
        if d:
            if s < r:
                d, s = 0, r
            else:
                d += 1
                if d >= reset_distance:
                    d = s = 0
        elif r:
            d, s = 1, r
    return (dist + 9) // 10, gc, start                                             # TransformDistanceToStartOfErrorRegion


def _check_chain(arrays, codes, dom, rate, enforce=True, shift_distance=0):
    cfg_reset = int(arrays["coverage.reset_distance"][0]) if "coverage.reset_distance" in arrays else None
    reads = sum(int(arrays[f"read_lengths.{seg}"].sum()) for seg in range(2))
    total = sum(int((arrays[f"read_lengths.{seg}"] * (np.arange(len(arrays[f"read_lengths.{seg}"])) + int(arrays[f"read_lengths.{seg}.from"][0]))).sum()) for seg in range(2))
    gc_range = ((total + reads // 2) // reads) // 2                                # Simulator.cpp:2713-2721
    codes = codes.astype(np.int64)
    feature, gc, start = _chain_states(codes, rate, gc_range, cfg_reset)
    feature = feature + shift_distance
    last = np.concatenate([[0], codes[:-1]])
    dom5 = _dominant_of_last_five(codes)
    use = np.arange(len(codes)) >= 4 * max(cfg_reset, gc_range)                   # whatever state the chain began with has been forgotten
    states = np.stack([feature, gc, start], axis=1)
    zs = []
    # the dominant-error draw: one table per (base, previous base, dominant base)
    p_all, col_all, group_all, g = [], [], [], 0
    for b in range(4):
        for l in range(4):
            for d5 in range(4):
                sel = use & (codes == b) & (last == l) & (dom5 == d5)
                if sel.sum() < 200:
                    continue
                tab = _table(arrays, f"dom_error.{b}.{l}.{d5}")
                p, ok = _conditionals(tab, states[sel])
                col = np.searchsorted(np.sort(tab[0]), dom[sel][ok])
                col = np.argsort(tab[0])[col]                                      # column of the observed value
                assert np.array_equal(tab[0][col], dom[sel][ok])
                p_all.append(p), col_all.append(col), group_all.append(np.full(len(col), g))
                g += 1
    k = max(p.shape[1] for p in p_all)
    p_dom = np.concatenate([np.pad(p, ((0, 0), (0, k - p.shape[1]))) for p in p_all])
    zs.append(_z_scores(p_dom, np.concatenate(col_all), np.concatenate(group_all)))
    # the error-rate draw: one table per (base, dominant error just drawn)
    p_all, col_all, group_all, g = [], [], [], 0
    for b in range(4):
        for e in range(5):
            sel = use & (codes == b) & (dom == e)
            if sel.sum() < 200:
                continue
            tab = _table(arrays, f"error_rate.{b}.{e}")
            p, ok = _conditionals(tab, states[sel])
            lookup = {int(v): c for c, v in enumerate(tab[0])}
            col = np.asarray([lookup[int(v)] for v in rate[sel][ok]])
            # coarse outcome classes so that every class has counts: rate 0, 1-9, 10-19, ...
            cls = np.minimum((tab[0] + 9) // 10, 6)
            pc = np.zeros((len(p), 7))
            for c in range(p.shape[1]):
                pc[:, cls[c]] += p[:, c]
            p_all.append(pc), col_all.append(cls[col]), group_all.append(np.full(len(col), g))
            g += 1
    zs.append(_z_scores(np.concatenate(p_all), np.concatenate(col_all), np.concatenate(group_all)))
    z = np.concatenate(zs)
    if enforce:
        assert len(z) >= 30, len(z)
        assert np.abs(z).max() < 5.0, np.sort(np.abs(z))[-5:]
    return float(np.abs(z).max())


def _tracks_of(sim_sys_errors, codes):
    """both strands as (bases in the strand's direction, dominant errors, rates)"""
    L = len(codes)
    out = []
    for strand in (0, 1):
        dom, rate = sim_sys_errors(strand, 0, L)
        bases = codes if strand == 0 else 3 - codes[::-1]                           # the reverse track is indexed by the position on the reverse strand
        out.append((bases.astype(np.int64), np.asarray(dom).astype(np.int64), np.asarray(rate).astype(np.int64)))
    return out


def _profile_arrays_with_reset(cfg):
    arrays = synth.make_profile(cfg, seed=5, n_ref_seqs=1)
    arrays["coverage.reset_distance"] = np.asarray([cfg["reset_distance"]])
    return arrays


def _run_chain_check(backend_cls, workdir, use_oracle):
    p = P.Pair(backend_cls, workdir, "chain_stat", synth.TINY, [60000], seed=31, num_pairs=1000)
    try:
        arrays = _profile_arrays_with_reset(synth.TINY)
        codes = p.seqs[0][1]
        get = (lambda strand, seq, L: p.osim.sys_errors(strand, seq)) if use_oracle else p.b.sys_errors
        worst, control = 0.0, 0.0
        for bases, dom, rate in _tracks_of(get, codes):
            assert (rate > 0).mean() > 0.02                                         # the profile does draw systematic errors
            worst = max(worst, _check_chain(arrays, bases, dom, rate))
            control = max(control, _check_chain(arrays, bases, dom, rate, enforce=False, shift_distance=1))
        assert control > 8.0, control                                               # the check notices a distance row that is one off
    finally:
        p.close()


def test_systematic_error_chains_of_the_oracle(workdir):
    _run_chain_check(EmuBackend, workdir, use_oracle=True)


def test_systematic_error_chains_of_the_host_emulation(workdir):
    _run_chain_check(EmuBackend, workdir, use_oracle=False)


@pytest.mark.gpu
def test_systematic_error_chains_of_the_product(workdir):
    _run_chain_check(GpuBackend, workdir, use_oracle=False)


# ---------------------------------------------------------------------------------------------------------------- a14
def _bias_of_all_starts(arrays, codes, length, seq):
    codes = codes.astype(np.int64)
    L = len(codes)
    start = np.arange(L - length + 1)
    end = start + length
    fwd = codes[(start[:, None] - 10 + np.arange(30)[None, :]) % L]
    rev = 3 - codes[(end[:, None] - 1 + 10 - np.arange(30)[None, :]) % L]
    prefix = np.concatenate([[0], np.cumsum(np.isin(codes, (1, 2)))])
    gc = _percent(prefix[end] - prefix[start], length)
    sur = arrays["frag.sur_bias"].reshape(3, -1)
    ilb_from = int(arrays["frag.insert_lengths_bias.from"][0])
    general = arrays["frag.ref_seq_bias"][seq] * arrays["frag.insert_lengths_bias"][length - ilb_from]
    return general * arrays["frag.gc_bias"][gc] * _kmer_bias(sur, fwd) * _kmer_bias(sur, rev)


def _run_normalization_check(backend_cls, workdir, use_oracle):
    lengths = [6000, 4100]
    p = P.Pair(backend_cls, workdir, "norm_stat", synth.TINY, lengths, seed=37, num_pairs=5000)
    try:
        arrays = synth.make_profile(synth.TINY, seed=5, n_ref_seqs=len(lengths))
        src = p.osim if use_oracle else p.b
        norm, thr = np.asarray(src.norm_by_len()), np.asarray(src.thresholds())
        info = p.info
        il, il_from = arrays["frag.insert_lengths"], int(arrays["frag.insert_lengths.from"][0])
        first = next(k for k in range(len(il)) if il[k] and k + il_from >= 1) + il_from
        disp = arrays["frag.dispersion_parameters"]
        total_pairs = info["total_pairs"]
        bias_normalization = total_pairs / (2.0 * norm.sum())
        got = p.osim.bias_normalization() if use_oracle else info["bias_normalization"]
        assert abs(got / bias_normalization - 1.0) < 1e-11                          # :3562-3566
        for length in (first, first + 20, first + 40):                             # sample positions: the first length with counts, then every 20 (FragmentDistributionStats.h:235)
            biases = [_bias_of_all_starts(arrays, codes, length, s) for s, (_, codes) in enumerate(p.seqs) if len(codes) >= length]
            total, biggest = sum(b.sum() for b in biases), max(b.max() for b in biases)
            assert abs(norm[length] / total - 1.0) < 1e-10, (length, norm[length], total)
            # one coverage group here (the sequences' biases lie within a factor of two): its maximum at a sampled length is the maximum over all starts
            assert thr.shape[0] == 1
            mean = bias_normalization * biggest                                    # CalculateNonZeroThreshold, one allele
            r = min(mean / (disp[0] + disp[1] * mean), mean * 1e10)
            thr0 = (r / (r + mean)) ** r
            assert abs(thr[0, length, 0] / thr0 - 1.0) < 1e-9, (length, thr[0, length, 0], thr0)
            assert abs(thr[0, length, 1] / thr0 ** 2 - 1.0) < 1e-9
    finally:
        p.close()


def test_bias_normalization_of_the_oracle(workdir):
    _run_normalization_check(EmuBackend, workdir, use_oracle=True)


def test_bias_normalization_of_the_host_emulation(workdir):
    _run_normalization_check(EmuBackend, workdir, use_oracle=False)


@pytest.mark.gpu
def test_bias_normalization_of_the_product(workdir):
    _run_normalization_check(GpuBackend, workdir, use_oracle=False)


# ---------------------------------------------------------------------------------------------------------------- a16
REGIONS = [(0, 1, 1.0), (2000, 8000, 0.0), (9000, 15000, 0.3), (16000, 22000, 0.75), (23000, 29000, 1.0)]      # the first one takes the reverse walk's blind spot


def _run_conversion_check(backend_cls, workdir, use_oracle):
    p = P.Pair(backend_cls, workdir, "meth_stat", synth.TINY, [30000], seed=41, num_pairs=40000, edits={"no_substitutions": True, "no_indels": True})
    try:
        name = p.seqs[0][0].split(" ")[0]
        bed = workdir / "meth_stat.bed"
        bed.write_text("".join(f"{name}\t{a}\t{b}\t{m}\n" for a, b, m in REGIONS))
        (p.osim if use_oracle else p.b).read_methylation(bed)
        p.align_normalization()
        tb = p.info["total_blocks"]
        if use_oracle:
            frags = p.osim.sieve(1, tb + 1)
            texts = p.osim.create_reads(frags)
        else:
            frags, r1, r2 = p.b.pairs(1, tb + 1)
            texts = (r1, r2)
        codes = p.seqs[0][1].astype(np.int64)
        region_of = np.full(len(codes), -1)
        for k, (a, b, _) in enumerate(REGIONS):
            region_of[a:b] = k
        # the reference's reverse walk begins one past the fragment's last base (Simulator.cpp:2237,2243; DESIGN.md row a16, kept): the base at a region's border
        # belongs to the neighbouring stretch there -- borders are left out of the count
        border = np.zeros(len(codes), bool)
        for a, b, _ in REGIONS:
            border[max(a - 1, 0):a + 1] = border[b - 1:b + 1] = True
        letters = np.frombuffer(b"ACGT", np.uint8)
        lut = np.full(256, 9, np.int64)
        lut[letters] = np.arange(4)
        kept = np.zeros(len(REGIONS) + 1), np.zeros(len(REGIONS) + 1)                # [converted, not converted] per region; the last entry: outside every region
        seen = set()
        for seg, text in enumerate(texts):
            lines = text.split(b"\n")
            for i, f in enumerate(frags):
                site = (int(f["start"]), int(f["len"]), int(f["strand"]), seg)
                if f["dup"] or site in seen:                                       # duplicates of a site share its converted template
                    continue
                seen.add(site)
                read = lut[np.frombuffer(lines[4 * i + 1], np.uint8)]
                reverse = seg != int(f["strand"])
                n = min(len(read), int(f["len"]))                                   # beyond the fragment the read runs into the adapter
                at = int(f["start"]) + int(f["len"]) - 1 - np.arange(n) if reverse else int(f["start"]) + np.arange(n)
                template = 3 - codes[at] if reverse else codes[at]
                is_c = (template == 1) & ~border[at]
                assert np.array_equal(read[:n][template != 1], template[template != 1])               # errors are off: everything but a converted C is the template
                assert np.isin(read[:n][is_c], (1, 3)).all()
                reg = np.where(region_of[at[is_c]] >= 0, region_of[at[is_c]], len(REGIONS))
                np.add.at(kept[0], reg, read[:n][is_c] == 3)
                np.add.at(kept[1], reg, read[:n][is_c] == 1)
        converted, stayed = kept
        assert converted[len(REGIONS)] == 0 and stayed[len(REGIONS)] > 10000           # outside the regions nothing converts
        for k, (_, _, m) in enumerate(REGIONS[1:], start=1):
            n = converted[k] + stayed[k]
            assert n > 5000, (k, n)
            want = 1.0 - m
            if want in (0.0, 1.0):
                assert converted[k] == want * n, (k, converted[k], n)
            else:
                z = (converted[k] - want * n) / np.sqrt(n * want * (1 - want))
                assert abs(z) < 5.0, (k, z, converted[k] / n)
                assert abs((converted[k] - (want + 0.05) * n) / np.sqrt(n * want * (1 - want))) > 5.0      # a rate five points off would be noticed
    finally:
        p.close()


def test_conversion_rate_of_the_oracle(workdir):
    _run_conversion_check(EmuBackend, workdir, use_oracle=True)


def test_conversion_rate_of_the_host_emulation(workdir):
    _run_conversion_check(EmuBackend, workdir, use_oracle=False)


@pytest.mark.gpu
def test_conversion_rate_of_the_product(workdir):
    _run_conversion_check(GpuBackend, workdir, use_oracle=False)
