#!/usr/bin/env python3
"""bench.py -- simulated 2x150 bp read pairs per second of the illuminaPE hot path on N MI355X.

Workload (BASELINE.json configs[1], restated on synthetic data as SURVEY.md section 8(d) prescribes): per GPU a
4 641 652 bp E. coli-sized reference sequence (i.i.d. bases, GC 50.8 %) and 10 M read pairs, the pre-fitted synthetic
profile P0 (2x150, qualities 2..41, insert lengths ~ LogNormal(350, 0.25) in [50,1000), one tile).
One step = one pass of the hot path over the job: coverage sieve -> fragments -> reads -> FASTQ text of both mates, all
resident in HBM (rsq_sim_pairs over every block of the rank's share, batched by block range).  Pre-passes (table packing,
bias normalisation, systematic-error tracks) happen once before the timed region.

--gpus N is ONE job: a reference of N such sequences, N x 10 M pairs, one seed; the job's blocks of 1000 start positions
are split into contiguous ranges by reseq_amd.sharding.partition_blocks and every rank simulates its range ("fragments
sharded by reference block").  Per-GPU work is fixed as N grows (weak scaling); blocks are independent, so there is no
data-path collective -- torch.distributed (RCCL) carries the timing barrier and the job totals.  `value` is the whole-job
aggregate.  Prints ONE JSON line on rank 0.

--scaling strong: the fixed-size job instead -- ONE such sequence and 10 M pairs whatever N is, its blocks split over the N ranks
by expected pairs (sharding.block_weights); the line says "scaling": "strong".  Under --gpus N > 1 the default (weak) line also
carries a `strong_scaling` leg measured after the headline's timed region, so that one driver run records both kinds.
"""
import argparse
import glob
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from reseq_amd import api, sharding, synth  # noqa: E402

GENOME = 4_641_652
PAIRS = 10_000_000
A_PAIR = 1436                 # algorithmic HBM bytes per 2x150 pair (DESIGN.md "Roofline")
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec
LDS_BYTES_PER_PAIR = 2 * 150 * 808   # algorithmic LDS bytes per 2x150 pair (808 B of single-precision table rows per base)
LDS_PEAK_TBS = 256 * 256 * 2.4e9 / 1e12   # MI355X_MICROARCH.md "LDS": 256 B/clk/CU for ds_read_b64/b128, 256 CUs, 2.4 GHz = 157 TB/s
TA_CYCLES_PER_LOAD = 23.0     # a wave-level 16-byte load occupies the CU's vector-memory path that long (exp/ta_bench.hip)
N_SIMD, N_CU, N_XCD = 1024, 256, 8


# ------------------------------------------------------------------------------------------------ CPU baseline (the oracle)
_ORACLE = {}                 # the prepared oracle simulation of the current sample, inherited by the forked workers


def _oracle_worker(part, parts, barrier, queue, literal):
    """one process: the oracle on a block range of the sample"""
    sim = _ORACLE["sim"]
    lo, hi = sharding.partition_blocks(sim.total_blocks(), parts)[part]
    barrier.wait()
    t0 = time.perf_counter()
    fr = sim.sieve_literal(lo, hi) if literal else sim.sieve(lo, hi)
    sieve_s = time.perf_counter() - t0
    r1, r2 = sim.create_reads(fr)
    dt = time.perf_counter() - t0
    keep = part == 0 and parts == 1 and not literal        # the one-process run also hands its text and pre-pass results to the parity check
    queue.put((len(fr), len(r1) + len(r2), dt, (r1, r2, sim.bias_normalization(), sim.thresholds()) if keep else None, sieve_s))


def _oracle_run(profile_path, seqs, seed, sample_bp, procs, literal=False):
    """pre-passes once in this process, then `procs` forked workers on a block range each (they share the prepared state)"""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    ctx = mp.get_context("fork")
    name, codes = seqs[0]
    n_pairs = int(round(PAIRS * sample_bp / GENOME))
    prof = O.Profile(profile_path)
    ref = O.Reference([(name, codes[:sample_bp])])
    _ORACLE["sim"] = O.Sim(prof, ref, seed, n_pairs)
    barrier, queue = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=_oracle_worker, args=(i, procs, barrier, queue, literal)) for i in range(procs)]
    for p in ps:
        p.start()
    res = [queue.get() for _ in ps]
    for p in ps:
        p.join()
    _ORACLE.pop("sim").close()
    ref.close()
    prof.close()
    pairs, nbytes, wall = sum(r[0] for r in res), sum(r[1] for r in res), max(r[2] for r in res)
    text = next((r[3] for r in res if r[3] is not None), None)
    if literal:
        return pairs, nbytes, wall, n_pairs, max(r[4] for r in res)
    return pairs, nbytes, wall, n_pairs, text


def usable_cpus():
    """hardware threads this process may run on at once: its affinity mask, capped by the container's CPU quota (cgroup cpu.max / cfs_quota_us) --
    a box of this pool shows 256 threads and grants 16 CPUs; processes beyond the quota only take turns"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota, period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(profile_path, seqs, seed):
    """The CPU oracle (a port of the reference's algorithm, oracle/liboracle.so) on bounded samples of the same workload -- the first
    bases of the reference at the workload's pair density; sieve + CreateReads timed, pre-passes excluded like the GPU figure: one
    process on 300 kb, then one process per CPU the container may use (usable_cpus) on a block range each of 1.5 Mb."""
    cores, host_threads = usable_cpus(), len(os.sched_getaffinity(0))
    p1, b1, t1, _, text = _oracle_run(profile_path, seqs, seed, 300_000, 1)
    many_bp = 1_500_000
    pn, bn, tn, _, _ = _oracle_run(profile_path, seqs, seed, many_bp, cores)
    # the reference's own loop shape: one uniform per (start position, fragment length) (Simulator.cpp:2302-2306, orc_sieve_blocks_literal) -- about a thousand draws per
    # start position where the gap sieve above makes one or two.  A smaller sample: the cells, not the pairs, are what costs.
    literal_bp = 100_000
    pl, _, tl, _, sieve_l = _oracle_run(profile_path, seqs, seed, literal_bp, 1, literal=True)
    out = {"value": pn / tn, "unit": "read-pairs/s", "cores": cores, "kind": "port",
           "single_thread": {"value": p1 / t1, "unit": "read-pairs/s", "cores": 1, "sieve": "gaps"},
           "single_thread_reference_shaped": {"value": pl / tl, "unit": "read-pairs/s", "cores": 1, "sieve": "literal: one uniform per (start, fragment length), Simulator.cpp:2302-2306",
                                              "sample_bp": literal_bp, "pairs": pl, "seconds": round(tl, 2), "sieve_seconds": round(sieve_l, 2),
                                              "sieve_ns_per_start_position": round(sieve_l / literal_bp * 1e9, 1)},
           "sample": f"oracle/liboracle.so with the GAP sieve (the product's substitution for the reference's per-cell loop: the passing cells of a start position drawn by their gaps, "
                     f"1-2 draws per start position instead of ~1000 -- `value` and `single_thread` are therefore FASTER than the reference's algorithm would be; `single_thread_reference_shaped` "
                     f"times the reference's loop as written, orc_sieve_blocks_literal, on the first {literal_bp} bp: {pl} pairs in {tl:.1f} s of which {sieve_l:.1f} s sieve).  "
                     f"1 thread: first 300000 bp, {p1} pairs, {b1} FASTQ bytes in {t1:.1f} s; {cores} processes (one per CPU of the container's quota; the host shows {host_threads} hardware threads; a block range "
                     f"each): first {many_bp} bp, {pn} pairs in {tn:.1f} s; sieve + CreateReads, pre-passes excluded.  The bridge to the reference itself (SURVEY.md 8(d) item 4, cannot be "
                     f"re-measured here: the reference does not build in this image): BASELINE.md's survey ran Simulator::Simulate under a header shim on an 8-vCPU 2.1 GHz Xeon with a degenerate "
                     f"(hand-filled) profile -- 17.7 k pairs/s on one thread, 95 k on eight; from the three per-base draws alone with realistic K, 9.3 k pairs/s per core",
           "reference_survey": {"pairs_per_s_one_thread": 17700, "pairs_per_s_eight_threads": 95000, "source": "BASELINE.md (survey-time measurement, degenerate profile, another host)"}}
    return out, text


# ------------------------------------------------------------------------------ statistical parity on the baseline's sample
def _kmer_counts(seq_lines, k=8):
    lut = np.full(256, 4, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    counts = np.zeros(4 ** k, np.int64)
    for whole in seq_lines:                              # reads of one length at a time, 50 000 reads per pass
        for at in range(0, len(whole), 50_000):
            group = whole[at:at + 50_000]
            a = lut[np.frombuffer(b"".join(group), np.uint8).reshape(len(group), -1)]
            ok = np.ones((a.shape[0], a.shape[1] - k + 1), bool)
            code = np.zeros(ok.shape, np.int32)
            for j in range(k):
                col = a[:, j:a.shape[1] - k + 1 + j]
                ok &= col < 4
                code = code * 4 + np.minimum(col, 3)
            counts += np.bincount(code[ok], minlength=4 ** k)
    return counts


def _by_length(lines):
    groups = {}
    for line in lines:
        groups.setdefault(len(line), []).append(line)
    return [g for n, g in groups.items() if n >= 8]


def error_rate_by_context(lines, codes, read_len=150, pos_bin=10):
    """Substitution rate by (read position bin, quality) -- BASELINE.json's "error-rate-by-context" -- of the reads whose CIGAR is one run of
    matches: the read id carries the fragment's start and end on the reference (Simulator.cpp:609-631), a mate reads the forward strand from the
    start or the reverse strand from the end, whichever fits better.  Returns (mismatches, bases) as [position bin][quality] arrays."""
    lut = np.full(256, 4, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    plain = b" %dM " % read_len
    ids = [i for i in range(0, len(lines) - 3, 4) if plain in lines[i] and len(lines[i + 1]) == read_len]
    mism, total = np.zeros((read_len // pos_bin, 64), np.int64), np.zeros((read_len // pos_bin, 64), np.int64)
    if not ids:
        return mism, total
    codes = np.concatenate([codes.astype(np.uint8), np.full(read_len, 4, np.uint8)])       # fragments shorter than a read run into the adapter: not compared
    L = len(codes) - read_len
    for at in range(0, len(ids), 100_000):
        part = ids[at:at + 100_000]
        fields = [lines[i].split(b":") for i in part]
        a, b = np.array([int(f[1]) for f in fields]), np.array([int(f[3]) for f in fields])
        start, end = np.minimum(a, b) - 1, np.maximum(a, b)              # the id gives first and last base (1-based), swapped for fragments of the reverse strand
        bases = lut[np.frombuffer(b"".join(lines[i + 1] for i in part), np.uint8).reshape(len(part), read_len)]
        qual = np.frombuffer(b"".join(lines[i + 3] for i in part), np.uint8).reshape(len(part), read_len) - 33
        k = np.arange(read_len)[None, :]
        fwd = codes[np.minimum(start[:, None] + k, L + read_len - 1)]
        rev = 3 - codes[np.clip(end[:, None] - 1 - k, 0, L - 1)].astype(np.int16)
        whole = (end - start >= read_len)[:, None]                                          # the template covers the read
        m_f, m_r = (bases != fwd) & whole, (bases != rev) & whole
        use_f = (m_f.sum(1) <= m_r.sum(1))[:, None]
        m = np.where(use_f, m_f, m_r)
        cell = (k // pos_bin) * 64 + np.minimum(qual, 63)
        total += np.bincount(cell[np.broadcast_to(whole, cell.shape)], minlength=total.size).reshape(total.shape)
        mism += np.bincount(cell[m], minlength=total.size).reshape(total.shape)
    return mism, total


def fastq_statistics(text, codes=None):
    """8-mer spectrum, per-position quality histogram and the sum of the ids' error counts (E<n>) of FASTQ text; with the reference's bases
    also the substitution rate by (read position bin, quality)"""
    lines = text.split(b"\n")
    seqs, quals = lines[1::4], lines[3::4]
    kmers = _kmer_counts(_by_length([s for s in seqs if s]))
    n = max((len(q) for q in quals), default=0)
    qhist = np.zeros((n, 64), np.int64)
    for group in _by_length([q for q in quals if q]):
        a = np.frombuffer(b"".join(group), np.uint8).reshape(len(group), -1) - 33
        for p in range(a.shape[1]):
            qhist[p] += np.bincount(np.minimum(a[:, p], 63), minlength=64)
    errors = sum(int(l.rsplit(b" E", 1)[1]) for l in lines[0::4] if l)
    context = error_rate_by_context(lines, codes) if codes is not None else None
    return kmers, qhist, errors, context


def kl_divergence(p_counts, q_counts):
    p, q = p_counts / max(p_counts.sum(), 1), q_counts / max(q_counts.sum(), 1)
    m = (p > 0) & (q > 0)
    missing = float(p[(p > 0) & (q == 0)].sum())         # mass the other spectrum does not have at all
    return float((p[m] * np.log(p[m] / q[m])).sum()) + (np.inf if missing > 0 else 0.0)


def parity_on_sample(profile_path, seqs, seed, device, oracle_text):
    """The metrics BASELINE.json names beside the throughput, GPU output against the CPU oracle's on the baseline's one-thread sample
    (same seed: the two are bit-identical by construction, so every distance must be exactly 0)."""
    sample_bp = 300_000
    name, codes = seqs[0]
    tmp = tempfile.mkdtemp(prefix="rsq_parity_")
    fpath = os.path.join(tmp, "sample.fa")
    synth.write_fasta(fpath, [(name, codes[:sample_bp])])
    prof, ref = api.Profile(profile_path), api.Reference(fpath, seed)
    sim = api.Simulator(prof, ref, device)
    info = sim.prepare(seed, int(round(PAIRS * sample_bp / GENOME)))
    # the device sums the bias normalisation as a tree, the oracle sequentially (relative 1e-11): both continue from the oracle's thresholds
    bn, thr = oracle_text[2], oracle_text[3]
    out_norm = abs(info.bias_normalization / bn - 1.0)
    sim.set_normalization(bn, thr)
    _, g1, g2 = sim.pairs(1, info.total_blocks + 1)
    sim.close()
    out = {"sample": f"first {sample_bp} bp, seed {seed}", "bias_normalization_rel_diff": out_norm, "fastq_identical": bool(g1 == oracle_text[0] and g2 == oracle_text[1])}
    kg, qg, eg, cg = fastq_statistics(g1 + g2, codes[:sample_bp])
    ko, qo, eo, co = fastq_statistics(oracle_text[0] + oracle_text[1], codes[:sample_bp])
    out["kmer_kl"] = kl_divergence(kg, ko)
    out["quality_histogram_max_abs_diff"] = int(np.abs(qg - qo).max())
    out["error_count_diff"] = int(eg - eo)
    # error rate by context: substitutions per base by (read position bin of 10, quality), device against oracle, over the cells with at least 1000 bases
    (mg, tg), (mo, to) = cg, co
    busy = (tg >= 1000) & (to >= 1000)
    rate_g, rate_o = mg[busy] / np.maximum(tg[busy], 1), mo[busy] / np.maximum(to[busy], 1)
    out["error_rate_by_context"] = {"cells": int(busy.sum()), "bases": int(tg.sum()), "substitutions": int(mg.sum()), "max_abs_rate_diff": float(np.abs(rate_g - rate_o).max()) if busy.any() else None,
                                    "counts_identical": bool(np.array_equal(mg, mo) and np.array_equal(tg, to)),
                                    "rate_by_quality_decile": [round(float(mg[:, q:q + 10].sum() / max(tg[:, q:q + 10].sum(), 1)), 6) for q in range(0, 50, 10)]}
    out["pairs"] = int(g1.count(b"\n") // 4)
    return out


# ------------------------------------------------------------------------------------------------------- roofline evidence
def committed_counters(tiles=1):
    """Counters cannot be read from inside an un-profiled run: the newest committed collection of profiles/collect.sh for this workload is reported
    with its file name and the kernel it was taken from (bench.py's own launch time of that kernel is next to it, and `counters_stale` says whether
    the kernel sources have changed since)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_pmc.json")))      # by name: round, then letter (mtimes do not survive a checkout)
    files = [f for f in files if (f"_tiles{tiles}_" in os.path.basename(f)) == (tiles > 1) and ("_tiles" in os.path.basename(f)) == (tiles > 1)]
    if not files:
        return None
    tag = os.path.basename(files[-1])[:-len("_pmc.json")]
    try:
        pmc = json.load(open(files[-1]))
        recorded = pmc.pop("_kernel_source_hash", None)
        kernel = sorted((k for k in pmc if "fill_reads" in k), key=lambda k: not k.startswith("rsq_spec_"))[0]      # the profile's own kernel (rsq_spec_fill_reads) when it ran
        c = {n: v["mean"] for n, v in pmc[kernel].items()}
        cycles = c["GRBM_GUI_ACTIVE"] / N_XCD
        from reseq_amd.provenance import kernel_source_hash
        out = {"source": f"profiles/{tag}_pmc.json", "kernel_source_hash_when_collected": recorded, "counters_stale": recorded != kernel_source_hash(), "kernel": kernel, "kernel_cycles": cycles, "kernel_ms_at_2.4GHz": cycles / 2.4e6,
               "vmem_loads_per_launch": c["SQ_INSTS_VMEM_RD"], "valu_instructions_per_launch": c["SQ_INSTS_VALU"],
               "vmem_issue_frac": c["SQ_INSTS_VMEM_RD"] / N_CU * TA_CYCLES_PER_LOAD / cycles, "ta_busy_frac": c["TA_BUSY_avr"] / cycles,
               "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / (N_SIMD * cycles), "lds_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / (N_CU * cycles),
               "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]}
        tfile = os.path.join(ROOT, "profiles", f"{tag}_traffic.json")
        out["hbm_bytes_per_launch"] = float(json.load(open(tfile))["hbm_bytes_per_launch"]) if os.path.exists(tfile) else None
        return out
    except (OSError, ValueError, KeyError, IndexError):
        return None


class TorchBuffer:
    """device memory from torch (uint8) with the two attributes api.Simulator.pairs_device reads"""

    def __init__(self, torch, nbytes, device):
        self.t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        self.ptr, self.nbytes = self.t.data_ptr(), int(nbytes)


# ------------------------------------------------------------------------------------------- BASELINE.json configs[2]-[4]
def _parity_case(name, *args):
    """one of the suite's parity cases (tests/parity_cases.py: the C ABI on this device against the CPU oracle, byte for byte) as a yes/no in the bench line"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pathlib import Path
    import parity_cases as P
    from backends import GpuBackend
    t0 = time.perf_counter()
    try:
        with tempfile.TemporaryDirectory(prefix="rsq_parity_") as d:
            getattr(P, name)(GpuBackend, Path(d), *args)
        return {"case": f"tests/parity_cases.py::{name}{args if args else ''}", "equal_to_oracle": True, "seconds": round(time.perf_counter() - t0, 1)}
    except AssertionError as e:
        return {"case": f"tests/parity_cases.py::{name}{args if args else ''}", "equal_to_oracle": False, "what": str(e)[:300]}


def _timed_pairs_job(sim, nb, batch, device, bufs, hash_blocks=48000):
    """every block of the job once, in calls of `batch` blocks: pairs, bytes, seconds on the device, summed kernel times, SHA-256 of the two files' text of the first
    `hash_blocks` blocks (downloaded after the call's time is taken)"""
    import hashlib
    h1, h2 = hashlib.sha256(), hashlib.sha256()
    n = nbytes = 0
    t_gpu = 0.0
    kernel_ms = {}
    for lo in range(1, nb + 1, batch):
        hi = min(nb + 1, lo + batch)
        if not bufs:                                                       # sized for the largest call of the leg: `bufs` arrives with the block count to size for
            raise ValueError("size the buffers first (_pairs_buffers)")
        t1 = time.perf_counter()
        k, l1, l2, rc = sim.pairs_device(lo, hi, bufs[0], bufs[1])
        t_gpu += time.perf_counter() - t1
        if rc != api.RSQ_OK:
            raise api.RsqError(rc, api.lib().rsq_last_error().decode())
        for key in ("slot_table", "variant_templates", "sieve", "sieve_screen", "sieve_emit", "fill_reads", "format_write", "scan"):
            if sim.last_kernel_launches(key):
                kernel_ms[key] = kernel_ms.get(key, 0.0) + sim.last_kernel_ms(key)
        n += k
        nbytes += l1 + l2
        if hash_blocks and hi <= hash_blocks + 1:
            h1.update(bufs[0].to_numpy(np.uint8, l1).tobytes())
            h2.update(bufs[1].to_numpy(np.uint8, l2).tobytes())
    out = {"pairs": n, "fastq_bytes": nbytes, "gpu_s": round(t_gpu, 4), "pairs_per_s": n / t_gpu, "kernel_ms": {k: round(v, 1) for k, v in kernel_ms.items()}}
    if hash_blocks:
        out[f"sha256_first_{hash_blocks}_blocks"] = h1.hexdigest() + ":" + h2.hexdigest()
    return out


def _pairs_buffers(sim, nb, batch, device):
    """two FASTQ buffers for calls of up to `batch` blocks: the need of every such call is asked first (a call without buffers returns the sizes)"""
    need1 = need2 = 0
    for lo in range(1, nb + 1, batch):
        _, l1, l2, _ = sim.pairs_device(lo, min(nb + 1, lo + batch), None, None)
        need1, need2 = max(need1, l1), max(need2, l2)
    return [api.DeviceArray(device, need1 + 4096), api.DeviceArray(device, need2 + 4096)]


def other_configs(device, seed, which=("2", "3", "4")):
    """BASELINE.json's configs[2], [3] and [4] on ONE GPU, each measured once after the headline's timed region: rates with the inputs resident in
    HBM, per-kernel times, checksums, invariance under another batching, and a small oracle parity case of the same kind of input.  Not the headline metric."""
    import ctypes as C
    import hashlib
    from reseq_amd import workloads
    out = {}
    tmp = tempfile.mkdtemp(prefix="rsq_other_")
    ppath = os.path.join(tmp, "p0.rsqp")
    arrays = workloads.p0_profile(ppath)
    if "2" in which:
        # configs[2] seqToIllumina (replaceQuals): 8 M records of 150 bases as FASTA text resident in HBM, found, parsed, simulated and formatted on the device
        t0 = time.perf_counter()
        n = 8_000_000
        rows, _ = workloads.seq_to_illumina_rows(n, arrays)
        width = rows.shape[1]
        d_text = api.DeviceArray.from_numpy(device, np.concatenate([rows.reshape(-1), np.zeros(8, np.uint8)]))
        del rows
        make_s = time.perf_counter() - t0
        prof = api.Profile(ppath)
        sim = api.Simulator(prof, None, device)
        sim.prepare(seed)
        d_out = api.DeviceArray(device, n * (2 * 150 + 64 + 16))
        need, k, used = C.c_size_t(0), C.c_uint64(0), C.c_size_t(0)

        def call():
            t = time.perf_counter()
            api._check(api.lib().rsq_sim_error_model_fasta(sim.h, 0, d_text.ptr, n * width, 1, d_out.ptr, d_out.nbytes, C.byref(need), C.byref(k), C.byref(used), None))
            return time.perf_counter() - t
        call()
        seconds = [call() for _ in range(3)]
        assert k.value == n and used.value == n * width
        ms = {name: round(sim.last_kernel_ms(name), 2) for name in ("parse_records", "fill_reads", "format_write")}
        text = d_out.to_numpy(np.uint8, need.value)
        out["configs[2]"] = {"workload": f"seqToIllumina: {n} records of 150 bases as FASTA text ({n * width} bytes) resident in HBM, one rsq_sim_error_model_fasta call", "reads_per_s": n / min(seconds),
                             "seconds": [round(t, 4) for t in seconds], "kernel_ms": ms, "fastq_bytes": int(need.value), "sha256": hashlib.sha256(text.tobytes()).hexdigest(),
                             "make_inputs_s": round(make_s, 1), "parity_sample": _parity_case("case_error_model_p0")}
        del text
        for d in (d_text, d_out):
            d.free()
        sim.close()
        prof.close()
    if "3" in which:
        # configs[3] Drosophila-sized at full size: 143.7 Mb in 7 sequences + 1000 scaffolds, coverage 30
        t0 = time.perf_counter()
        fpath, lengths = workloads.drosophila_sized(tmp)
        make_s = time.perf_counter() - t0
        prof, ref = api.Profile(ppath), api.Reference(fpath, 7)
        sim = api.Simulator(prof, ref, device)
        t0 = time.perf_counter()
        info = sim.prepare(7, 0, 30.0)
        prep_s = time.perf_counter() - t0
        bufs = _pairs_buffers(sim, info.total_blocks, info.total_blocks, device)         # also the warm-up (kernels compiled for the profile)
        # the whole job in one call (as the headline runs its job), and in calls of 24000 and 8000 blocks (the command line takes about 4 M pairs per call);
        # the checksum covers the first 48000 blocks: the batched runs have a call that ends there, the whole job is cut there for it
        runs = {f"batch_{b}": _timed_pairs_job(sim, info.total_blocks, b, device, bufs) for b in (24000, 8000)}
        runs["one_call"] = _timed_pairs_job(sim, info.total_blocks, info.total_blocks, device, bufs, hash_blocks=0)
        same = len({(r["pairs"], r["fastq_bytes"]) for r in runs.values()}) == 1 and runs["batch_24000"]["sha256_first_48000_blocks"] == runs["batch_8000"]["sha256_first_48000_blocks"]
        out["configs[3]"] = {"workload": f"illuminaPE: Drosophila-sized reference ({sum(lengths)} bp in {len(lengths)} sequences, 1000 of them scaffolds shorter than the longest insert), P0, coverage 30, ONE GPU",
                             "pairs_per_s": max(r["pairs_per_s"] for r in runs.values()), "pairs": runs["batch_24000"]["pairs"], "runs": runs, "batching_invariant": same, "prepare_s": round(prep_s, 2),
                             "make_inputs_s": round(make_s, 1), "parity_sample": _parity_case("case_coverage_driven")}
        for d in bufs:
            d.free()
        sim.close()
        ref.close()
        prof.close()
        os.remove(fpath)
    if "4" in which:
        # configs[4] human-sized at 1/10 scale with variants of every kind on two alleles and methylation
        t0 = time.perf_counter()
        job = workloads.human_sized(tmp, 0.1)
        make_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        prof, ref = api.Profile(ppath), api.Reference(job["fasta"], 7)
        alleles = ref.read_variants(job["vcf"])
        sim = api.Simulator(prof, ref, device)
        sim.read_methylation(job["bed"])
        load_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        info = sim.prepare(7, 0, 30.0)
        prep_s = time.perf_counter() - t0
        bufs = _pairs_buffers(sim, info.total_blocks, info.total_blocks, device)
        runs = {f"batch_{b}": _timed_pairs_job(sim, info.total_blocks, b, device, bufs) for b in (24000, 12000)}
        runs["one_call"] = _timed_pairs_job(sim, info.total_blocks, info.total_blocks, device, bufs, hash_blocks=0)
        same = len({(r["pairs"], r["fastq_bytes"]) for r in runs.values()}) == 1 and runs["batch_24000"]["sha256_first_48000_blocks"] == runs["batch_12000"]["sha256_first_48000_blocks"]
        out["configs[4]"] = {"workload": f"illuminaPE at 1/10 of the human-sized job: {sum(job['lengths'])} bp in 24 sequences, {alleles} alleles, {job['substitutions']} substitutions + {job['indels']} "
                                         f"insertions / deletions, {job['regions']} methylation regions, P0, coverage 30, ONE GPU",
                             "pairs_per_s": max(r["pairs_per_s"] for r in runs.values()), "pairs": runs["batch_24000"]["pairs"], "runs": runs, "batching_invariant": same,
                             "load_s": round(load_s, 2), "prepare_s": round(prep_s, 2), "make_inputs_s": round(make_s, 1), "parity_sample": _parity_case("case_p0_variants", "meth")}
        for d in bufs:
            d.free()
        sim.close()
        ref.close()
        prof.close()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


class PairsJob:
    """One illuminaPE job on this rank's device: profile + reference loaded, pre-passes done, the rank's block range (balanced by expected pairs) cut into
    batches, FASTQ buffers sized from the largest batch.  measure() = `warmup` untimed steps, then `steps` timed ones bracketed by synchronize + barrier."""

    def __init__(self, torch, dist, dev, rank, world, profile_path, fasta_path, seed, total_pairs, seq_lens, batch_blocks):
        self.torch, self.dist = torch, dist
        self.prof = api.Profile(profile_path)
        self.ref = api.Reference(fasta_path, seed)
        self.sim = api.Simulator(self.prof, self.ref, dev.index)
        t0 = time.perf_counter()
        self.info = self.sim.prepare(seed, total_pairs)
        self.prep_s = time.perf_counter() - t0
        weights = sharding.block_weights(seq_lens, self.info.insert_to, self.sim.ref_seq_bias(len(seq_lens)))
        self.my_lo, self.my_hi = sharding.partition_blocks(self.info.total_blocks, world, weights)[rank]
        self.batches = sharding.batches(self.my_lo, self.my_hi, batch_blocks)
        # size the FASTQ buffers once from the largest batch (first pass measures), then reuse them
        self.need1 = self.need2 = 0
        for lo, hi in self.batches:
            n, l1, l2, rc = self.sim.pairs_device(lo, hi, None, None)
            if rc not in (api.RSQ_OK, api.RSQ_ENOSPC):
                raise api.RsqError(rc, api.lib().rsq_last_error().decode())
            self.need1, self.need2 = max(self.need1, l1), max(self.need2, l2)
        self.bufs = [(TorchBuffer(torch, self.need1 + 4096, dev), TorchBuffer(torch, self.need2 + 4096, dev)) for _ in range(2)]

    def step(self):
        pairs = nbytes = launches = 0
        fill_ms = 0.0
        for lo, hi in self.batches:
            n, l1, l2, rc = self.sim.pairs_device(lo, hi, self.bufs[0][0], self.bufs[0][1])
            if rc != api.RSQ_OK:
                raise api.RsqError(rc, api.lib().rsq_last_error().decode())
            pairs += n
            nbytes += l1 + l2
            if n:
                fill_ms += self.sim.last_kernel_ms("fill_reads")              # summed over the call's launches (pipelined sub-ranges)
                launches += self.sim.last_kernel_launches("fill_reads")
        return pairs, nbytes, fill_ms, launches

    def sync(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def measure(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        self.sync()
        t0 = time.perf_counter()
        pairs = nbytes = launches = 0
        fill_ms = 0.0
        for _ in range(steps):
            p, b, f, n_launch = self.step()
            pairs += p
            nbytes += b
            fill_ms += f
            launches += n_launch
        self.sync()
        return {"pairs": pairs, "nbytes": nbytes, "fill_ms": fill_ms, "launches": launches, "elapsed": time.perf_counter() - t0}

    def close(self):
        self.bufs = None
        self.sim.close()
        self.ref.close()
        self.prof.close()


def ranks_or_relaunch(args):
    """(rank, local rank, world size) of this process.  Started bare with --gpus N > 1 -- no launcher's environment -- the script starts its N ranks itself, one process
    per GPU through torch.distributed.run, as Simulator::Simulate starts its own workers (Simulator.cpp:2830-2836); a launcher's world size that is not --gpus is refused."""
    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and args.gpus > 1:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): start it as `python bench.py --gpus N` or with --nproc-per-node equal to --gpus")
    if args.backend == "gloo" and not (args.emulate or args.shareDevice):
        raise SystemExit("bench.py: --backend gloo needs --shareDevice (ranks as processes on shared devices) or --emulate; ranks that own a GPU each talk through RCCL (nccl)")
    if args.shareDevice and args.backend != "gloo":
        raise SystemExit("bench.py: --shareDevice needs --backend gloo: RCCL cannot put two ranks on one device")
    return rank, local_rank, world


def single_rank_rendezvous():
    """--dist-single started bare: the rendezvous a launcher would have put into the environment (no effect under a launcher)"""
    import socket
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)


def rank_times(dist, device, elapsed, steps, world):
    """ms per step of every rank, in rank order (all-gather of one double)"""
    if dist is None:
        return [elapsed / steps * 1e3]
    import torch
    t = torch.tensor([elapsed / steps * 1e3], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def main_emulated(args, rank, world):
    """--emulate: this script's launch, sharding, timing brackets and totals with tests/hostemu's CPU loop over the kernels' per-lane functions where the device would be.
    The line it prints is marked as such and is no measurement of anything; the CPU suite runs it with two ranks over gloo (tests/test_bench_launch.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from backends import EmuBackend
    dist = None
    if world > 1 or args.dist_single:
        import torch.distributed as dist
        single_rank_rendezvous()
        dist.init_process_group("gloo")
    genome, pairs_per_rank = min(args.genome, 6000), min(args.pairs, 1500)
    tmp = tempfile.mkdtemp(prefix=f"rsq_bench_emu_{rank}_")
    ppath, fpath = os.path.join(tmp, "tiny.rsqp"), os.path.join(tmp, "ref.fa")
    n_seqs = world if args.scaling == "weak" else 1
    synth.write_profile(ppath, synth.make_profile(synth.TINY, seed=103741084, n_ref_seqs=n_seqs))
    seqs = []
    for i in range(n_seqs):
        seqs += synth.make_reference(2 + i, [genome], gc=args.gc, names=[f"synthTiny{i} len={genome}"])
    synth.write_fasta(fpath, seqs)
    sim = EmuBackend(ppath, fpath, args.seed)
    info = sim.prepare(args.seed, pairs_per_rank * n_seqs)
    weights = sharding.block_weights([genome] * n_seqs, info["insert_to"], sim.ref_seq_bias(n_seqs))
    my_lo, my_hi = sharding.partition_blocks(info["total_blocks"], world, weights)[rank]
    batches = sharding.batches(my_lo, my_hi, args.batch_blocks)

    def step():
        pairs = nbytes = 0
        for lo, hi in batches:
            fr, r1, r2 = sim.pairs(lo, hi)
            pairs += len(fr)
            nbytes += len(r1) + len(r2)
        return pairs, nbytes

    def sync():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    pairs = nbytes = 0
    for _ in range(args.steps):
        p, b = step()
        pairs += p
        nbytes += b
    sync()
    elapsed = time.perf_counter() - t0
    per_rank = rank_times(dist, "cpu", elapsed, args.steps, world)
    total_pairs, total_bytes, elapsed = sharding.job_totals(dist, "cpu", pairs, nbytes, elapsed)
    sim.close()
    if rank == 0:
        print(json.dumps({"metric": "EMULATED on the CPU (tests/hostemu), not a measurement: simulated read-pairs/sec", "emulated": True, "value": total_pairs / elapsed, "unit": "read-pairs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_per_rank": per_rank,
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": "TINY profile, host emulation -- the launcher's test", "backend": args.backend, "reference_bp": genome * n_seqs, "pairs_per_step": total_pairs / args.steps,
                                     "collectives": None if dist is None else "torch.distributed gloo (the emulation's stand-in for RCCL): barrier, all_reduce of the totals, all_gather of the ranks' times",
                                     "fastq_bytes_per_step": total_bytes / args.steps, "blocks_of_rank_0": [my_lo, my_hi], "total_blocks": info["total_blocks"]}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=PAIRS, help="read pairs per GPU")
    ap.add_argument("--genome", type=int, default=GENOME, help="reference bases per GPU")
    ap.add_argument("--batch-blocks", type=int, default=4800, help="blocks of 1000 start positions per rsq_sim_pairs call (4800: the whole E. coli-sized job in one call)")
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--gc", type=float, default=0.508, help="G+C fraction of the synthetic reference (E. coli: 0.508)")
    ap.add_argument("--tiles", type=int, default=1, help="NOT the headline: P0 with so many tiles (per-tile tables; above one tile the read kernel serves one tile per workgroup)")
    ap.add_argument("--lib", default=None, help="another build of libreseq_amd.so (experiment builds, exp/)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="rsq_set_option before the simulator is created (measurements)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend of a run with more than one rank (and of --dist-single); gloo only with "
                    "--shareDevice or --emulate")
    ap.add_argument("--shareDevice", action="store_true", help="NOT a scaling measurement: rank r runs on device r %% devices and the small exchanges go over gloo on the CPU -- the N-rank "
                    "path of this script with the real kernels on a host with fewer devices than ranks (the line says so)")
    ap.add_argument("--emulate", action="store_true", help="TEST SWITCH, not a measurement: the launch / sharding / totals path of this script with the host emulation of the kernels "
                    "(tests/hostemu, the TINY profile, a few thousand pairs) in the device's place -- what the CPU suite runs with --gpus 2 --backend gloo")
    ap.add_argument("--dist-single", action="store_true", help="initialise torch.distributed although there is one rank (the RCCL calls of the N-rank path on one GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak (default): N sequences and N x --pairs on N GPUs; strong: ONE sequence and --pairs pairs split over the N GPUs")
    ap.add_argument("--no-strong-leg", action="store_true", help="with --gpus N > 1 and weak scaling: skip the extra fixed-size measurement (`strong_scaling` in the line)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the `other_configs` leg (BASELINE.json's configs[2]-[4] on one GPU after the headline; about two minutes)")
    ap.add_argument("--other-configs", default="2,3,4", help="which of configs[2], [3], [4] the leg runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-delivery", action="store_true", help="skip the value_to_host leg (it is skipped anyway with more than one GPU: every rank would pin 15 GB of host memory)")
    args = ap.parse_args()

    if args.lib:
        api.use_library(args.lib)
    rank, local_rank, world = ranks_or_relaunch(args)
    if args.emulate:
        return main_emulated(args, rank, world)

    # the job: `world` sequences, world x pairs; every rank holds the (small) reference and the tables, and simulates its block range
    tmp = tempfile.mkdtemp(prefix=f"rsq_bench_{rank}_")
    ppath = os.path.join(tmp, "p0.rsqp")
    fpath = os.path.join(tmp, "ref.fa")
    cfg = synth.P0 if args.tiles <= 1 else synth.p0_with_tiles(args.tiles)
    n_seqs = world if args.scaling == "weak" else 1
    synth.write_profile(ppath, synth.make_profile(cfg, seed=103741084, n_ref_seqs=n_seqs))
    seqs = []
    for i in range(n_seqs):
        seqs += synth.make_reference(2 + i, [args.genome], gc=args.gc, names=[f"synthEcoli{i} len={args.genome}"])
    synth.write_fasta(fpath, seqs)
    ppath1, fpath1 = ppath, fpath
    if n_seqs > 1 and not args.no_strong_leg:        # the fixed-size job of the strong-scaling leg: the first sequence alone
        ppath1, fpath1 = os.path.join(tmp, "p0_one.rsqp"), os.path.join(tmp, "ref_one.fa")
        synth.write_profile(ppath1, synth.make_profile(cfg, seed=103741084, n_ref_seqs=1))
        synth.write_fasta(fpath1, seqs[:1])

    # the CPU oracle first (rank 0 of a single-GPU run only): its worker processes are forked before any device context exists
    baseline = oracle_text = None
    if not args.no_cpu_baseline and world == 1 and not args.dist_single and "TORCHELASTIC_RUN_ID" not in os.environ:
        baseline, oracle_text = cpu_baseline(ppath, seqs, args.seed)

    import torch
    dist = None
    hip_device = local_rank % torch.cuda.device_count() if args.shareDevice else local_rank
    torch.cuda.set_device(hip_device)
    dev = torch.device("cuda", hip_device)
    xdev = "cpu" if args.shareDevice else f"cuda:{hip_device}"                   # where the ranks' small exchanges live
    if world > 1 or args.dist_single or "TORCHELASTIC_RUN_ID" in os.environ:      # more than one rank, or one rank under a launcher: the same calls over RCCL
        import torch.distributed as dist
        single_rank_rendezvous()
        if args.shareDevice:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    for item in args.option:                      # after torch: the library binds to the HIP runtime torch has loaded
        name, value = item.split("=", 1)
        api.set_option(name, int(value))
    job = PairsJob(torch, dist, dev, rank, world, ppath, fpath, args.seed, args.pairs * n_seqs, [args.genome] * n_seqs, args.batch_blocks)
    sim, info, prep_s, my_lo, my_hi, bufs, need1, need2 = job.sim, job.info, job.prep_s, job.my_lo, job.my_hi, job.bufs, job.need1, job.need2
    sync = job.sync
    m = job.measure(args.steps, args.warmup)
    pairs, nbytes, fill_ms, fill_launches, elapsed = m["pairs"], m["nbytes"], m["fill_ms"], m["launches"], m["elapsed"]
    kernel_ms = {k: sim.last_kernel_ms(k) for k in ("sieve", "sieve_screen", "sieve_emit", "fill_reads", "format_write", "scan")}
    plan = sim.fill_plan()
    specialized, spec_note = sim.specialize(0)                         # already done by prepare(): this asks what runs
    plan["read_kernel"] = "compiled for the profile at run time (hiprtc)" if specialized else "the library's own instantiation"
    plan["read_kernel_note"] = spec_note
    try:
        kernel_ms["bin_tiles"] = sim.last_kernel_ms("bin_tiles")          # only when the read kernel runs binned by tile
    except api.RsqError:
        pass
    per_rank = rank_times(dist, xdev, elapsed, args.steps, world)
    total_pairs, total_bytes, elapsed = sharding.job_totals(dist, xdev, pairs, nbytes, elapsed)      # sum, sum, max over ranks

    # the fixed-size job beside the weak one (N > 1): ONE sequence and --pairs pairs, its blocks split over the ranks by expected pairs -- BASELINE's configs[3] / [4]
    # are of this kind (one genome sharded over 8).  After the headline's timed region, with its own barriers and the same max-over-ranks timing.
    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_strong_leg:
        fixed = PairsJob(torch, dist, dev, rank, world, ppath1, fpath1, args.seed, args.pairs, [args.genome], args.batch_blocks)
        ms = fixed.measure(args.steps, 1)
        s_rank = rank_times(dist, xdev, ms["elapsed"], args.steps, world)
        s_pairs, _, s_elapsed = sharding.job_totals(dist, xdev, ms["pairs"], ms["nbytes"], ms["elapsed"])
        strong = {"scaling": "strong", "value": s_pairs / s_elapsed, "unit": "read-pairs/s", "ms_per_step": s_elapsed / args.steps * 1e3, "ms_per_step_per_rank": s_rank,
                  "pairs_per_step": s_pairs / args.steps, "reference_bp": args.genome, "blocks_of_rank_0": [fixed.my_lo, fixed.my_hi], "total_blocks": fixed.info.total_blocks,
                  "note": "ONE E. coli-sized sequence and --pairs pairs split over the ranks by sharding.block_weights; compare with the N = 1 headline of the same workload"}
        fixed.close()

    # the same steps delivered to the host (what Simulator::Flush hands to the writer, Simulator.cpp:150-182): generation of batch k+1
    # overlaps the copy of batch k into page-locked host memory on a second stream
    to_host = None
    if not args.no_host_delivery and world == 1:
        host = [(torch.empty(need1 + 4096, dtype=torch.uint8, pin_memory=True), torch.empty(need2 + 4096, dtype=torch.uint8, pin_memory=True)) for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=dev)
        done = [torch.cuda.Event(), torch.cuda.Event()]

        turn = [0]
        # smaller batches than the device-resident leg: on this stack the copy to the host is a blit kernel (__amd_rocclr_copyBuffer in the
        # kernel trace), which gets compute units only between the launches of the persistent read kernel
        host_batches = sharding.batches(my_lo, my_hi, min(args.batch_blocks, 1200))

        def host_step():
            moved = 0
            for lo, hi in host_batches:
                k = turn[0] & 1                                         # the two buffer pairs alternate across batches and steps
                turn[0] += 1
                done[k].synchronize()                                   # the copy that last used this pair of buffers
                n, l1, l2, rc = sim.pairs_device(lo, hi, bufs[k][0], bufs[k][1])      # returns when the text is complete
                if rc != api.RSQ_OK:
                    raise api.RsqError(rc, api.lib().rsq_last_error().decode())
                with torch.cuda.stream(copy_stream):
                    host[k][0][:l1].copy_(bufs[k][0].t[:l1], non_blocking=True)
                    host[k][1][:l2].copy_(bufs[k][1].t[:l2], non_blocking=True)
                    done[k].record(copy_stream)
                moved += l1 + l2
            copy_stream.synchronize()
            return moved

        host_step()
        sync()
        t0 = time.perf_counter()
        moved = sum(host_step() for _ in range(args.steps))
        sync()
        host_elapsed = time.perf_counter() - t0
        _, moved_all, host_elapsed = sharding.job_totals(dist, xdev, 0, moved, host_elapsed)
        # the same delivered as .gz: every batch's text becomes gzip members on the device (rsq_sim_gzip_device), and the members -- a third of the bytes -- cross the link
        gz_bufs = [(TorchBuffer(torch, need1 // 2 + (1 << 20), dev), TorchBuffer(torch, need2 // 2 + (1 << 20), dev)) for _ in range(2)]
        gz_ms = [0.0]

        sim.gzip_keep_code(True)                                      # one Huffman code for the run, as the command line keeps it
        gz_batches = sharding.batches(my_lo, my_hi, args.batch_blocks)      # whole calls: a third of the bytes cross the link, the copies no longer crowd the read kernel out

        def host_step_gz():
            moved = text = 0
            for lo, hi in gz_batches:
                k = turn[0] & 1
                turn[0] += 1
                n, l1, l2, rc = sim.pairs_device(lo, hi, bufs[0][0], bufs[0][1])
                if rc != api.RSQ_OK:
                    raise api.RsqError(rc, api.lib().rsq_last_error().decode())
                done[k].synchronize()                                   # the copy that last used this pair of member buffers
                sizes = []
                for f, length in ((0, l1), (1, l2)):
                    size, rc = sim.gzip_device(bufs[0][f].ptr, length, gz_bufs[k][f].ptr, gz_bufs[k][f].nbytes)
                    if rc != api.RSQ_OK:
                        raise api.RsqError(rc, api.lib().rsq_last_error().decode())
                    gz_ms[0] += sim.last_kernel_ms("gzip")
                    sizes.append(size)
                with torch.cuda.stream(copy_stream):
                    host[k][0][:sizes[0]].copy_(gz_bufs[k][0].t[:sizes[0]], non_blocking=True)
                    host[k][1][:sizes[1]].copy_(gz_bufs[k][1].t[:sizes[1]], non_blocking=True)
                    done[k].record(copy_stream)
                moved += sum(sizes)
                text += l1 + l2
            return moved, text                                          # the copies run on: the next batch -- of this step or the next -- is generated under them

        host_step_gz()
        copy_stream.synchronize()
        sync()
        gz_ms[0] = 0.0
        t0 = time.perf_counter()
        gz_moved = gz_text = 0
        gz_steps = max(args.steps, 6)                                   # the last step's copy has nothing to hide under: a few more steps than the headline's three
        for _ in range(gz_steps):
            m_, t_ = host_step_gz()
            gz_moved += m_
            gz_text += t_
        copy_stream.synchronize()                                       # every member of every step is in host memory
        sync()
        gz_elapsed = time.perf_counter() - t0
        compressed = {"value": total_pairs / args.steps * gz_steps / gz_elapsed, "unit": "read-pairs/s", "steps": gz_steps, "ms_per_step": gz_elapsed / gz_steps * 1e3, "host_gbytes_per_s": gz_moved / gz_elapsed / 1e9,
                      "text_over_members": gz_text / max(gz_moved, 1), "gzip_kernels_ms_per_step": gz_ms[0] / gz_steps, "gzip_gbytes_of_text_per_s": gz_text / max(gz_ms[0], 1e-9) / 1e6, "batch_blocks": args.batch_blocks,
                      "note": "FASTQ text of both mates as gzip members made on the device (rsq_deflate.h: BGZF-framed, one dynamic Huffman code per call) copied to page-locked host "
                              "buffers; the copy of batch k overlaps generation and compression of batch k+1"}
        to_host = {"value": total_pairs / host_elapsed, "unit": "read-pairs/s", "ms_per_step": host_elapsed / args.steps * 1e3, "compressed": compressed,
                   "host_gbytes_per_s": moved_all / host_elapsed / 1e9, "batch_blocks": min(args.batch_blocks, 1200),
                   "note": "FASTQ text of both mates copied to page-locked host buffers, copy of batch k "
                   "overlapping the generation of batch k+1 (the link binds: 7.5 GB per step and GPU)"}

    others = None
    if world == 1 and not args.no_other_configs and args.tiles <= 1:
        job.close()                                                        # the headline's buffers go back first
        sim = None
        others = other_configs(local_rank, args.seed, tuple(args.other_configs.split(",")))
    if rank == 0:
        launches = max(fill_launches, 1)                                   # k_fill_reads launches in the timed region
        avg_fill_s = fill_ms / 1e3 / launches
        achieved = A_PAIR * (pairs / launches) / avg_fill_s / 1e9          # GB/s of algorithmic traffic in the dominant kernel
        counters = committed_counters(args.tiles)
        out = {
            "metric": "simulated read-pairs/sec (2x150 bp)", "value": total_pairs / elapsed, "unit": "read-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_per_rank": per_rank, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: E. coli-sized 4.64 Mb synthetic reference sequence and 10 M pairs " + ("per GPU" if args.scaling == "weak" else "in all, split over the GPUs") + ", pre-fitted synthetic profile P0 (2x150), "
                                   "illuminaPE hot path (sieve + CreateReads + FASTQ text) resident in HBM" +
                                   (f" -- NOT the headline: P0 with {args.tiles} tiles (per-tile tables)" if args.tiles > 1 else ""),
                       "tiles": args.tiles, "fill_plan": plan, "options": args.option, "reference_bp": args.genome * n_seqs,
                       "pairs_requested": args.pairs * n_seqs, "pairs_per_step_per_gpu": pairs // args.steps, "fastq_bytes_per_step_per_gpu": nbytes // args.steps,
                       "batch_blocks": args.batch_blocks, "read_kernel_launches_per_step": launches / args.steps, "blocks_of_rank_0": [my_lo, my_hi], "total_blocks": info.total_blocks,
                       "sharding": "one job; contiguous block ranges per GPU balanced by expected pairs (partition_blocks with block_weights); no data-path collective",
                       "collectives": None if dist is None else ("torch.distributed gloo on the CPU (the ranks share devices)" if args.shareDevice else "torch.distributed nccl (RCCL)") +
                                      ": barrier, all_reduce of the totals, all_gather of the ranks' times",
                       **({"ranks_share_devices": True, "note": f"--shareDevice: {world} processes on {torch.cuda.device_count()} device(s) -- the N-rank path with the real kernels, "
                                                                "not a scaling measurement"} if args.shareDevice else {})},
            "roofline": {"bound": "valu", "bound_of_achieved": "hbm", "kernel": "k_fill_reads", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": counters["hbm_bytes_per_launch"] if counters else None, "traffic_source": counters["source"].replace("_pmc", "_traffic") if counters else None,
                         "algorithmic_bytes_per_launch": A_PAIR * (pairs / launches), "bytes_per_pair": A_PAIR, "pairs_per_launch": pairs / launches,
                         "avg_launch_ms": avg_fill_s * 1e3,
                         "note": "table-lookup + RNG bound, not HBM bound: 1.4 KB of algorithmic HBM traffic per pair (DESIGN.md); what binds is in `secondary`",
                         "secondary": None if not counters else {
                             "resource": "VALU issue", "frac": counters["valu_busy_frac"], "vmem_issue_frac": counters["vmem_issue_frac"], "ta_busy_frac": counters["ta_busy_frac"],
                             "lds_busy_frac": counters["lds_busy_frac"], "lds_bank_conflict_frac": counters["lds_bank_conflict_frac"],
                             "lds_roofline": {"achieved": LDS_BYTES_PER_PAIR * (pairs / launches) / avg_fill_s / 1e12, "peak": LDS_PEAK_TBS, "unit": "TB/s",
                                              "frac": LDS_BYTES_PER_PAIR * (pairs / launches) / avg_fill_s / 1e12 / LDS_PEAK_TBS, "bytes_per_pair": LDS_BYTES_PER_PAIR,
                                              "note": "algorithmic LDS bytes (SURVEY.md 8(d)) over this run's launch time; the array's measured busy share is lds_busy_frac, "
                                                      "of which lds_bank_conflict_frac are conflict cycles"},
                             "kernel": counters["kernel"], "kernel_ms_when_profiled": counters["kernel_ms_at_2.4GHz"], "source": counters["source"],
                             "counters_stale": counters["counters_stale"],
                             "note": "fractions of the kernel's cycles from the committed PMC collection: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles); "
                                     "SQ_INSTS_VMEM_RD / 256 CUs x 23 cycles / cycles; TA_BUSY_avr / cycles; SQ_LDS_IDX_ACTIVE (LDS-array cycles, all CUs) / (256 CUs x cycles)"}},
            "kernel_ms_last_batch": kernel_ms,
            "prepare_s": prep_s, "sys_chain_passes": info.sys_chain_passes,
        }
        if others:
            out["other_configs"] = others
        if strong:
            out["strong_scaling"] = strong
        if to_host:
            out["value_to_host"] = to_host
        if baseline:
            out["cpu_baseline"] = baseline
            out["parity_sample"] = parity_on_sample(ppath, seqs, args.seed, local_rank, oracle_text)
            out["kmer_kl"] = out["parity_sample"]["kmer_kl"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
