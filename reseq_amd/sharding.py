"""Multi-GPU layout of the hot path.

Blocks of 1000 start positions are independent given the replicated tables, the reference and the systematic-error
tracks: a block's fragments, read ids and Philox counters depend on (seed, sequence, start, length) only.  Ranks therefore
take disjoint contiguous block ranges and there is no data-path collective; the ranks' FASTQ shards in rank order are byte
for byte the single-GPU output (Simulator.cpp:2384-2401 hands blocks to threads the same way).  Every rank copies its shard to
its offset of the output file itself (place_shard), so the merge is N parallel copies, not one serial one.
torch.distributed (RCCL on the GPU box, gloo in the CPU tests) carries only the job totals and the shard sizes.
"""
import os
from typing import Callable, List, Sequence, Tuple


def partition_blocks(total_blocks: int, world: int, weights: Sequence[float] = None) -> List[Tuple[int, int]]:
    """Contiguous block-id ranges [lo, hi) (ids run 1..total_blocks) per rank, balanced by `weights` (expected pairs
    per block; uniform when None).  Every block belongs to exactly one rank; ranks may receive an empty range."""
    if world < 1:
        raise ValueError("world size must be positive")
    if weights is None:
        weights = [1.0] * total_blocks
    if len(weights) != total_blocks:
        raise ValueError("one weight per block")
    total = float(sum(weights))
    bounds = [1]
    acc, b = 0.0, 0
    for r in range(1, world):
        target = total * r / world
        while b < total_blocks and acc + weights[b] / 2 <= target:
            acc += weights[b]
            b += 1
        bounds.append(b + 1)
    bounds.append(total_blocks + 1)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def batches(lo: int, hi: int, batch_blocks: int) -> List[Tuple[int, int]]:
    return [(b, min(hi, b + batch_blocks)) for b in range(lo, hi, batch_blocks)]


def simulate_shard(pairs_fn: Callable[[int, int], Tuple[int, bytes, bytes]], block_range: Tuple[int, int], batch_blocks: int):
    """Runs `pairs_fn(block_lo, block_hi) -> (n_pairs, r1_text, r2_text)` over a rank's range in batches."""
    n, r1, r2 = 0, [], []
    for lo, hi in batches(block_range[0], block_range[1], batch_blocks):
        k, a, b = pairs_fn(lo, hi)
        n += k
        r1.append(a)
        r2.append(b)
    return n, b"".join(r1), b"".join(r2)


def job_totals(dist, device, pairs: int, nbytes: int, elapsed: float):
    """Whole-job aggregate: sum of pairs and bytes, max of the ranks' elapsed time (bench.py's contract)."""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    c = torch.tensor([float(pairs), float(nbytes)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(c[0].item()), float(c[1].item()), float(t.item())


def gather_sizes(dist, device, mine: Sequence[int], world: int) -> List[List[int]]:
    """all-gather of a few byte counts per rank: row r = the values of rank r"""
    if dist is None:
        return [list(mine)]
    import torch
    t = torch.tensor(list(mine), dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [[int(v) for v in row.tolist()] for row in out]


def place_shard(shard_path: str, out_path: str, offset: int, chunk: int = 1 << 26) -> int:
    """copies the whole shard file to `offset` of the (existing, already sized) output file; in-kernel copy where the
    file system offers it (copy_file_range), pread / pwrite otherwise.  Returns the bytes copied."""
    size = os.path.getsize(shard_path)
    src, dst = os.open(shard_path, os.O_RDONLY), os.open(out_path, os.O_WRONLY)
    try:
        done, in_kernel = 0, hasattr(os, "copy_file_range")
        while done < size:
            n = 0
            if in_kernel:
                try:
                    n = os.copy_file_range(src, dst, min(chunk, size - done), done, offset + done)
                except OSError:
                    in_kernel = False
            if not in_kernel:
                buf = os.pread(src, min(chunk, size - done), done)
                n = os.pwrite(dst, buf, offset + done)
            if n <= 0 and in_kernel:
                in_kernel = False
                continue
            done += n
        return done
    finally:
        os.close(src)
        os.close(dst)
