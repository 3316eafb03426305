#!/usr/bin/env python3
"""bench.py -- simulated 2x150 bp read pairs per second of the illuminaPE hot path on N MI355X.

Workload (BASELINE.json configs[1], restated on synthetic data as SURVEY.md section 8(d) prescribes): a
4 641 652 bp E. coli-sized reference (i.i.d. bases, GC 50.8 %), the pre-fitted synthetic profile P0
(2x150, qualities 2..41, insert lengths ~ LogNormal(350, 0.25) in [50,1000), one tile), 10 M read pairs.
One step = one pass of the hot path over the whole reference: coverage sieve -> fragments -> reads -> FASTQ
text of both mates, all resident in HBM (rsq_sim_pairs over every block, batched by block range).
Pre-passes (table packing, bias normalisation, systematic-error tracks) happen once before the timed region.

With --gpus N each rank simulates its own reference shard of the same size (weak scaling; blocks are independent,
so there is no data-path collective); `value` is the whole-job aggregate.
Prints ONE JSON line on rank 0.
"""
import argparse
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


def cpu_baseline(profile_path, seqs, seed, sample_bp=150_000):
    """The CPU oracle (a port, 1 thread) on a bounded sample of the same workload: the first `sample_bp` bases of the
    reference at the same pair density.  Times sieve + CreateReads only (pre-passes excluded, like the GPU figure)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    name, codes = seqs[0]
    sub = [(name, codes[:sample_bp])]
    n_pairs = int(round(PAIRS * sample_bp / GENOME))
    prof = O.Profile(profile_path)
    ref = O.Reference(sub)
    sim = O.Sim(prof, ref, seed, n_pairs)
    t0 = time.perf_counter()
    fr = sim.sieve(1, sim.total_blocks() + 1)
    r1, r2 = sim.create_reads(fr)
    dt = time.perf_counter() - t0
    out = {"value": len(fr) / dt, "unit": "read-pairs/s", "cores": 1, "kind": "port",
           "sample": f"oracle/liboracle.so, first {sample_bp} bp of the reference at the workload's pair density: {len(fr)} pairs, "
                     f"{len(r1) + len(r2)} FASTQ bytes in {dt:.1f} s (sieve + CreateReads, pre-passes excluded)"}
    sim.close()
    ref.close()
    prof.close()
    return out


def measured_traffic():
    """HBM bytes per k_fill_reads launch from the PMC passes of profiles/collect.sh (FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md, plus WRITE_SIZE): counters cannot be read from inside an un-profiled run, so the
    figure of the newest committed collection is reported together with its file name."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return float(json.load(f)["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError, KeyError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=PAIRS)
    ap.add_argument("--genome", type=int, default=GENOME)
    ap.add_argument("--batch-blocks", type=int, default=1200)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--gc", type=float, default=0.508, help="G+C fraction of the synthetic reference (E. coli: 0.508)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    tmp = tempfile.mkdtemp(prefix=f"rsq_bench_{rank}_")
    ppath = os.path.join(tmp, "p0.rsqp")
    fpath = os.path.join(tmp, "ref.fa")
    synth.write_profile(ppath, synth.make_profile(synth.P0, seed=103741084))
    seqs = synth.make_reference(2 + rank, [args.genome], gc=args.gc, names=[f"synthEcoli{rank} len={args.genome}"])
    synth.write_fasta(fpath, seqs)

    prof = api.Profile(ppath)
    ref = api.Reference(fpath, args.seed)
    sim = api.Simulator(prof, ref, local_rank)
    t0 = time.perf_counter()
    info = sim.prepare(args.seed, args.pairs)
    prep_s = time.perf_counter() - t0
    nb = info.total_blocks
    batches = [(lo, min(nb + 1, lo + args.batch_blocks)) for lo in range(1, nb + 1, args.batch_blocks)]

    # size the FASTQ buffers once from the largest batch (first pass measures), then reuse them
    need1 = need2 = 0
    for lo, hi in batches:
        n, l1, l2, rc = sim.pairs_device(lo, hi, None, None)
        if rc not in (api.RSQ_OK, api.RSQ_ENOSPC):
            raise api.RsqError(rc, api.lib().rsq_last_error().decode())
        need1, need2 = max(need1, l1), max(need2, l2)
    r1 = api.DeviceArray(local_rank, need1 + 4096)
    r2 = api.DeviceArray(local_rank, need2 + 4096)

    def step():
        pairs = nbytes = 0
        fill_ms = 0.0
        for lo, hi in batches:
            n, l1, l2, rc = sim.pairs_device(lo, hi, r1, r2)
            if rc != api.RSQ_OK:
                raise api.RsqError(rc, api.lib().rsq_last_error().decode())
            pairs += n
            nbytes += l1 + l2
            if n:
                fill_ms += sim.last_kernel_ms("fill_reads")
        return pairs, nbytes, fill_ms

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    pairs = nbytes = 0
    fill_ms = 0.0
    for _ in range(args.steps):
        p, b, f = step()
        pairs += p
        nbytes += b
        fill_ms += f
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = {k: sim.last_kernel_ms(k) for k in ("sieve", "sieve_screen", "sieve_emit", "fill_reads", "format_write", "scan")}

    total_pairs, total_bytes, elapsed = sharding.job_totals(dist, f"cuda:{local_rank}", pairs, nbytes, elapsed)      # sum, sum, max over ranks

    if rank == 0:
        launches = args.steps * len(batches)
        avg_fill_s = fill_ms / 1e3 / launches
        achieved = A_PAIR * (pairs / launches) / avg_fill_s / 1e9          # GB/s of algorithmic traffic in the dominant kernel
        traffic, traffic_source = measured_traffic()
        out = {
            "metric": "simulated read-pairs/sec (2x150 bp)", "value": total_pairs / elapsed, "unit": "read-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: E. coli-sized 4.64 Mb synthetic reference, pre-fitted synthetic profile P0 (2x150), 10 M pairs, illuminaPE hot path "
                                   "(sieve + CreateReads + FASTQ text) resident in HBM", "reference_bp": args.genome, "pairs_requested": args.pairs,
                       "pairs_per_step_per_gpu": pairs // args.steps, "fastq_bytes_per_step_per_gpu": nbytes // args.steps, "batch_blocks": args.batch_blocks,
                       "sharding": "one reference shard per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": "k_fill_reads", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": A_PAIR * (pairs / launches), "bytes_per_pair": A_PAIR, "pairs_per_launch": pairs / launches, "avg_launch_ms": avg_fill_s * 1e3,
                         "note": "table-lookup + RNG bound, not HBM bound: 1.4 KB of algorithmic HBM traffic per pair (DESIGN.md)"},
            "kernel_ms_last_batch": kernel_ms,
            "prepare_s": prep_s, "sys_chain_passes": info.sys_chain_passes,
        }
        if not args.no_cpu_baseline and world == 1:                    # the oracle, on rank 0 of a single-GPU run only
            out["cpu_baseline"] = cpu_baseline(ppath, seqs, args.seed)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
