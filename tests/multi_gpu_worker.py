"""One rank of the in-process multi-GPU checks, started by torch.distributed.run (tests/test_multi_gpu.py):

    python -m torch.distributed.run --nproc-per-node N ... tests/multi_gpu_worker.py prepass <workdir> [--backend gloo --emulate | --backend gloo --shareDevice]

prepass: sharding.sharded_prepare over the ranks against rsq_sim_prepare of the whole reference on the same rank -- thresholds and normalisation equal exactly,
the systematic-error tracks equal over every position the rank's reads can touch, and chain states did cross shard borders.  On GPUs every rank owns device
LOCAL_RANK and the exchanges run over RCCL; with --shareDevice the ranks are processes on device LOCAL_RANK % devices with the real kernels and the exchanges run over gloo
on the CPU; with --emulate the host emulation stands where the device would be and the exchanges run over gloo."""
import argparse
import os
import pathlib
import sys

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["prepass"])
    ap.add_argument("workdir")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--emulate", action="store_true")
    ap.add_argument("--shareDevice", action="store_true")
    a = ap.parse_args()
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    import parity_cases as P
    from backends import EmuBackend, GpuBackend
    from reseq_amd import sharding, synth
    if a.emulate:
        dist.init_process_group("gloo")
        device, make = "cpu", lambda ppath, fpath: EmuBackend(ppath, fpath, 0)
    elif a.shareDevice:
        from reseq_amd import api
        dist.init_process_group("gloo")
        hip_device = local_rank % api.device_count()
        device, make = "cpu", lambda ppath, fpath: GpuBackend(ppath, fpath, 0, device=hip_device)
    else:
        assert a.backend == "nccl", "ranks on GPUs talk through RCCL"
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        device, make = f"cuda:{local_rank}", lambda ppath, fpath: GpuBackend(ppath, fpath, 0, device=local_rank)
    work = pathlib.Path(a.workdir) / f"pre{rank}"
    work.mkdir(parents=True, exist_ok=True)
    # one sequence spans several ranks, one is too short for blocks, one is a single block; with more ranks than the first case has work for, longer ones
    lengths = [9400, 80, 3210, 1000] if world <= 4 else [9400 * world // 4, 80, 3210, 1000, 5100]
    ppath, fpath, _ = P.make_inputs(work, "prepass", synth.TINY, lengths)

    class B:                                          # the interface sharding.sharded_prepare drives
        def __init__(self):
            self.b = make(ppath, fpath)
            self.seq_len = lengths
            self.states = []

        def ref_seq_bias(self):
            return self.b.ref_seq_bias(len(lengths))

        def prepare_sys_errors(self, lo, hi, in_state):
            out = self.b.prepare_sys_errors(lo, hi, in_state)
            self.states.append((list(in_state), out))
            return out

        def __getattr__(self, name):
            return getattr(self.b, name)

    b = B()
    info, (lo, hi), rounds = sharding.sharded_prepare(b, dist, device, rank, world, 23, 20000, 0.0, 1, "Pre")
    whole = make(ppath, fpath)
    winfo = whole.prepare(23, 20000, 0.0, 1, "Pre")
    assert info["total_pairs"] == winfo["total_pairs"] and info["bias_normalization"] == winfo["bias_normalization"], (info, winfo)
    assert np.array_equal(b.b.thresholds(), whole.thresholds()) and np.array_equal(b.b.norm_by_len(), whole.norm_by_len())
    # what the rank's blocks simulate is the whole pre-pass's, fragments and text (every base of a read takes its systematic error from the tracks)
    got, want = b.b.pairs(lo, hi), whole.pairs(lo, hi)
    assert len(got[0]) == len(want[0]) and (hi == lo or len(got[0]) > 0) and np.array_equal(got[0], want[0]) and got[1] == want[1] and got[2] == want[2], (rank, lo, hi)
    first_block, covered = 1, 0
    for seq, L in enumerate(lengths):                 # the tracks over the positions the rank's reads can touch
        if L < info["insert_to"]:
            continue
        nb = (L + 999) // 1000
        blo, bhi = max(first_block, lo), min(first_block + nb, hi)
        if blo < bhi:
            p_lo, t_hi = (blo - first_block) * 1000, min(L, min(L, (bhi - first_block) * 1000) + info["insert_to"])
            for strand in (0, 1):
                if not a.emulate:                     # the product hands out whole tracks only (rsq_sim_get_sys_errors refuses a sequence whose chains this rank did not
                    covered += t_hi - p_lo            # finish): there the reads above are the check; the emulation's tracks are compared position by position
                    break
                mine, ref = b.b.sys_errors(strand, seq, L), whole.sys_errors(strand, seq, L)
                sl = slice(p_lo, t_hi) if strand == 0 else slice(L - t_hi, L - p_lo)      # the reverse track is indexed L-1-position
                assert np.array_equal(mine[0][sl], ref[0][sl]) and np.array_equal(mine[1][sl], ref[1][sl]), (rank, seq, strand)
                covered += t_hi - p_lo
        first_block += nb
    entered = [s[0] for s in b.states]
    summary = [None] * world
    dist.all_gather_object(summary, dict(rank=rank, range=(lo, hi), rounds=rounds, covered=covered, nonzero_in=any(any(e) for e in entered), calls=len(b.states)))
    if rank == 0:
        assert sum(1 for s in summary if s["covered"] > 0) >= min(world, 4) and any(s["nonzero_in"] for s in summary), summary
        assert max(s["rounds"] for s in summary) >= 2, summary
        print("PREPASS_OK", summary)
    whole.close()
    b.b.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
