"""Multi-GPU `illuminaPE` simulation: one process per GPU, launched as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m reseq_amd.simulate \\
        -R ref.fa -s profile.rsqp -1 r1.fq -2 r2.fq --numReads 100000000 --seed 11

Every rank packs the replicated tables, takes a contiguous range of 1000-position blocks balanced by expected pairs
(`sharding.partition_blocks`) and computes ITS share of the pre-passes (`sharding.sharded_prepare`: bias sums per chunk + one exact
all-reduce, systematic-error chains over its own positions + the chain states at the shard borders), writes its FASTQ shard, and
after one all-gather of the shard sizes every rank copies its shard to its own offset of the output files, all ranks at once
(`sharding.place_shard`); rank 0 appends the adapter-only pairs.  The result is byte for byte the output of a
single-GPU run (`reseq_amd/reseq illuminaPE` with the same arguments): blocks are independent and every random stream is keyed by
(seed, sequence, start, length), not by rank (Simulator.cpp:2384-2401 distributes blocks over threads the same way).
torch.distributed (RCCL) carries the seed, the job totals, the shard sizes and three barriers.
"""
import argparse
import os
import sys
import time

from . import sharding


class GpuBackend:
    """reseq_amd.api.Simulator with reusable device buffers (the product path)."""

    def __init__(self, profile_path, fasta_path, device, replace_n_seed, vcf_path=None, methylation_path=None, sys_error_path=None):
        from . import api
        self.api = api
        self.prof = api.Profile(profile_path)
        self.ref = api.Reference(fasta_path, replace_n_seed)
        self.has_variants = bool(vcf_path)
        if vcf_path:
            self.ref.read_variants(vcf_path)
        self.sim = api.Simulator(self.prof, self.ref, device)
        if methylation_path:
            self.sim.read_methylation(methylation_path)
        self.sys_error_path = sys_error_path
        self.device = device
        self.r1 = self.r2 = None
        self.seq_len = [self.ref.sequence_length(i) for i in range(self.ref.num_sequences())]

    def prepare(self, seed, num_pairs, coverage, ref_bias_mode, base_identifier):
        i = self.sim.prepare(seed, num_pairs, coverage, ref_bias_mode, base_identifier)
        if self.sys_error_path:                                      # every rank loads the same tracks (Simulator.cpp:2800-2811)
            self.sim.read_sys_errors(self.sys_error_path)
        return dict(total_blocks=i.total_blocks, total_pairs=i.total_pairs, adapter_only_pairs=i.adapter_only_pairs, insert_to=i.insert_to)

    def ref_seq_bias(self):
        return self.sim.ref_seq_bias(len(self.seq_len))

    # the sharded pre-pass (not with a systematic-error profile, which replaces the chains anyway)
    @property
    def can_shard_prepare(self):
        return not self.sys_error_path

    def _info(self, i):
        return dict(total_blocks=i.total_blocks, total_pairs=i.total_pairs, adapter_only_pairs=i.adapter_only_pairs, insert_to=i.insert_to)

    def prepare_plan(self, *a):
        return self._info(self.sim.prepare_plan(*a))

    def bias_partials(self, lo, hi):
        return self.sim.bias_partials(lo, hi)

    def prepare_normalization(self, sums, maxes):
        self.sim.prepare_normalization(sums, maxes)

    def prepare_sys_errors(self, lo, hi, in_state):
        return self.sim.prepare_sys_errors(lo, hi, in_state)

    def prepare_finish(self):
        return self._info(self.sim.prepare_finish())

    def pairs(self, lo, hi):
        import numpy as np
        api = self.api
        for _ in range(2):
            n, l1, l2, rc = self.sim.pairs_device(lo, hi, self.r1, self.r2)
            if rc == api.RSQ_OK:
                return n, self.r1.to_numpy(np.uint8, l1).tobytes() if n else b"", self.r2.to_numpy(np.uint8, l2).tobytes() if n else b""
            if rc != api.RSQ_ENOSPC:
                raise api.RsqError(rc, api.lib().rsq_last_error().decode())
            for d in (self.r1, self.r2):
                if d is not None:
                    d.free()
            self.r1, self.r2 = api.DeviceArray(self.device, l1 + l1 // 8 + 4096), api.DeviceArray(self.device, l2 + l2 // 8 + 4096)
        raise RuntimeError("rsq_sim_pairs kept asking for larger buffers")

    def adapter_only_pairs(self, first, n):
        return self.sim.adapter_only_pairs(first, n)

    def close(self):
        for d in (self.r1, self.r2):
            if d is not None:
                d.free()
        self.sim.close()
        self.ref.close()
        self.prof.close()


block_weights = sharding.block_weights


def run_rank(backend, dist, rank, world, out1, out2, seed, num_pairs=0, coverage=0.0, ref_bias_mode=0, base_identifier="", batch_blocks=None, device="cpu"):
    """One rank's share.  `backend` offers prepare / ref_seq_bias / seq_len / pairs / adapter_only_pairs.  Returns (pairs of the whole
    job, seconds of the slowest rank)."""
    if world > 1 and getattr(backend, "can_shard_prepare", False):   # every rank its share of the pre-passes
        info, mine, _ = sharding.sharded_prepare(backend, dist, device, rank, world, seed, num_pairs, coverage, ref_bias_mode, base_identifier)
    else:
        info = backend.prepare(seed, num_pairs, coverage, ref_bias_mode, base_identifier)
        weights = block_weights(backend.seq_len, info["insert_to"], backend.ref_seq_bias())
        assert len(weights) == info["total_blocks"]
        mine = sharding.partition_blocks(info["total_blocks"], world, weights)[rank]
    if not batch_blocks:                                             # about 4 M pairs per call (large launches), at least 2000 blocks
        batch_blocks = int(min(100000, max(2000, 4e6 * info["total_blocks"] / max(1, info["total_pairs"]))))
    t0 = time.perf_counter()
    n_mine = n_bytes = 0
    shard1, shard2 = f"{out1}.rank{rank}", f"{out2}.rank{rank}"
    with open(shard1, "wb") as f1, open(shard2, "wb") as f2:
        for lo, hi in sharding.batches(mine[0], mine[1], batch_blocks):
            n, a, b = backend.pairs(lo, hi)
            n_mine += n
            n_bytes += len(a) + len(b)
            f1.write(a)
            f2.write(b)
    total_pairs, total_bytes, elapsed = sharding.job_totals(dist, device, n_mine, n_bytes, time.perf_counter() - t0)
    # every rank places its shard itself: offsets from the exclusive scan of the shard sizes (one all-gather of two lengths per rank)
    sizes = sharding.gather_sizes(dist, device, [os.path.getsize(shard1), os.path.getsize(shard2)], world)
    if rank == 0:
        for out, col in ((out1, 0), (out2, 1)):
            with open(out, "wb") as f:
                f.truncate(sum(row[col] for row in sizes))
    if dist is not None:
        dist.barrier()                                               # the output files exist at their final size
    for out, shard, col in ((out1, shard1, 0), (out2, shard2, 1)):
        sharding.place_shard(shard, out, sum(row[col] for row in sizes[:rank]))
        os.remove(shard)
    if dist is not None:
        dist.barrier()                                               # every shard is in place
    if rank == 0:
        with open(out1, "ab") as f1, open(out2, "ab") as f2:
            for first in range(0, info["adapter_only_pairs"], 100000):   # Simulator.cpp:2359-2382, as the single-GPU CLI does
                a, b = backend.adapter_only_pairs(first, min(100000, info["adapter_only_pairs"] - first))
                f1.write(a)
                f2.write(b)
    if dist is not None:
        dist.barrier()
    return int(total_pairs) + info["adapter_only_pairs"], elapsed


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("-R", "--refSim", "-r", "--refIn", dest="ref", required=True)
    ap.add_argument("-s", "--statsIn", dest="profile", required=True)
    ap.add_argument("-1", "--firstReadsOut", dest="out1", default="reseq-R1.fq")
    ap.add_argument("-2", "--secondReadsOut", dest="out2", default="reseq-R2.fq")
    ap.add_argument("-V", "--vcfSim", dest="vcf", default=None, help="variants to simulate per allele (substitutions)")
    ap.add_argument("--methylation", default=None, help="extended bed graph with methylation values per region (and allele)")
    ap.add_argument("--readSysError", default=None, help="systematic-error profile written by reseq illuminaPE --writeSysError")
    ap.add_argument("--numReads", type=int, default=0)
    ap.add_argument("-c", "--coverage", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--refBias", choices=["keep", "no", "draw"], default="keep")
    ap.add_argument("--recordBaseIdentifier", default="ReseqRead")
    ap.add_argument("--batchBlocks", type=int, default=0, help="blocks of 1000 start positions per device call (default: about 4 M pairs)")
    a = ap.parse_args(argv)
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    for out in (a.out1, a.out2):                                     # the single-GPU command line compresses these; shards placed by offset cannot be
        if out.endswith((".gz", ".bz2")):
            ap.error(f"{out}: compressed output is not supported by the multi-GPU launcher (write plain FASTQ and compress afterwards)")
    seed = (a.seed if a.seed is not None else int.from_bytes(os.urandom(8), "little")) & 0xFFFFFFFFFFFFFFFF
    if dist is not None:                                             # one seed for the whole job, all 64 bits of it
        t = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.broadcast(t, 0)
        seed = int(t.item()) & 0xFFFFFFFFFFFFFFFF
    backend = GpuBackend(a.profile, a.ref, local_rank, seed, a.vcf, a.methylation, a.readSysError)
    try:
        pairs, seconds = run_rank(backend, dist, rank, world, a.out1, a.out2, seed, a.numReads, a.coverage, {"keep": 0, "no": 1, "draw": 2}[a.refBias],
                                  a.recordBaseIdentifier, a.batchBlocks, f"cuda:{local_rank}")
        if rank == 0:
            print(f">>> Info: Generated {pairs} read pairs on {world} GPU(s), {seconds:.2f} s of generation on the slowest rank", file=sys.stderr)
    finally:
        backend.close()
        if dist is not None:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
