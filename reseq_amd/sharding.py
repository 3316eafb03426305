"""Multi-GPU layout of the hot path.

Blocks of 1000 start positions are independent given the replicated tables, the reference and the systematic-error
tracks: a block's fragments, read ids and Philox counters depend on (seed, sequence, start, length) only.  Ranks therefore
take disjoint contiguous block ranges and there is no data-path collective; the ranks' FASTQ shards in rank order are byte
for byte the single-GPU output (Simulator.cpp:2384-2401 hands blocks to threads the same way).  Every rank keeps the text of its
range in device memory until one all-gather of the sizes has told it its offset, then writes it there itself (rsq_sim_job_generate /
rsq_sim_job_write): N parallel writers, no shard files.
torch.distributed (RCCL on the GPU box, gloo in the CPU tests) carries only the job totals and the text sizes.
"""
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


def record_stretch(size: int, rank: int, world: int) -> Tuple[int, int]:
    """seqToIllumina over several ranks ("shards by input record ranges"): the bytes of the input file in which rank `rank` counts record starts"""
    return size * rank // world, size * (rank + 1) // world


def record_share(counts: Sequence[Sequence[int]], size: int, rank: int) -> Tuple[int, int, int]:
    """counts[r] = (record starts in rank r's stretch, offset of the first of them).  A record belongs to the rank in whose stretch it starts, wherever it ends:
    returns the rank's bytes [begin, end) of the file -- from its first record's start to the first record start of the next rank that has one (the file's end if
    none) -- and the index in the whole input of its first record (it selects the records' random streams, so the ranks' output is the single run's).  The first
    rank begins at byte 0: what stands in front of the first record is its to complain about.  A rank without a record start has an empty share."""
    first_record = sum(int(row[0]) for row in counts[:rank])
    end = size
    for n, first in counts[rank + 1:]:
        if n:
            end = int(first)
            break
    if rank == 0:
        return 0, end, 0
    if not counts[rank][0]:
        return end, end, first_record
    return int(counts[rank][1]), end, first_record


def block_weights(seq_len, insert_to, ref_seq_bias):
    """Expected pairs per block up to a constant: the sequence's reference bias (blocks of a sequence share it); sequences
    shorter than the longest insert have no blocks (Simulator.cpp:1159)."""
    w = []
    for length, bias in zip(seq_len, ref_seq_bias):
        if length >= insert_to:
            w += [float(bias)] * ((length + 999) // 1000)
    return w


def sharded_prepare(backend, dist, device, rank: int, world: int, seed, num_pairs=0, coverage=0.0, ref_bias_mode=0, base_identifier="", agree=None):
    """The pre-passes of a sharded job: every rank computes its share and the ranks exchange small arrays;
    every rank ends with the thresholds and, for the positions its reads can touch, the systematic-error tracks of a single-GPU
    rsq_sim_prepare, bit for bit.  `backend`: prepare_plan / ref_seq_bias / seq_len / bias_partials / prepare_normalization /
    prepare_sys_errors / prepare_finish (api.Simulator's methods).  Returns (info, the rank's block range, rounds of the chain exchange).

    a14 CalculateBiasNormalization: partial sums and maxima per chunk of 8192 start positions; a chunk is computed by the rank whose
    share holds its first position, all other entries are zero, so the all-reduce (SUM) is exact in any order, and every rank then adds
    the chunks up in chunk order -- the additions of the single-GPU run.
    a13 SetSystematicErrors: a chain's state (distance to the start of the error region, its rate) is all that crosses a shard
    border.  Every rank first runs its chunks speculatively to a fixed point; the states at the borders then travel rank to rank (forward
    chains to the right, reverse chains to the left) by all-gathers of two words per rank until no rank's entering state changed; a rank
    whose state changed redoes only the chunks that depend on it.

    `agree(error_or_None, what)`: called after every phase that runs the rank's own work (planning, bias sums, each round of the chains, finishing) and BEFORE the
    collective that follows it -- the launcher's exchange of a "failed" flag (simulate._agree), so that a rank whose phase raised (no memory for the bias sums, a bad
    input file on one node) does not leave the others waiting in the collective; without it a phase's exception simply propagates."""
    import numpy as np
    import torch

    def phase(what, f, *a):
        if agree is None:
            return f(*a)
        try:
            out, error = f(*a), None
        except Exception as e:      # noqa: BLE001 -- raised by agree() once every rank knows
            out, error = None, e
        agree(error, what)
        return out

    def plan():
        info = backend.prepare_plan(seed, num_pairs, coverage, ref_bias_mode, base_identifier)
        total_blocks = info["total_blocks"] if isinstance(info, dict) else info.total_blocks
        insert_to = info["insert_to"] if isinstance(info, dict) else info.insert_to
        weights = block_weights(backend.seq_len, insert_to, backend.ref_seq_bias())
        assert len(weights) == total_blocks
        return partition_blocks(total_blocks, world, weights)[rank]

    lo, hi = phase("planning the pre-pass", plan)
    sums, maxes = phase("summing the coverage bias of its share", backend.bias_partials, lo, hi)
    if dist is not None:
        t = torch.from_numpy(np.stack([sums, maxes])).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                     # every entry is non-zero on one rank
        sums, maxes = t[0].cpu().numpy(), t[1].cpu().numpy()
    phase("normalising the coverage bias", backend.prepare_normalization, sums, maxes)
    in_state, rounds = [0, 0], 0
    while True:
        out = phase("running the systematic-error chains of its share", backend.prepare_sys_errors, lo, hi, in_state)
        rounds += 1
        if dist is None:
            break
        mine = torch.tensor(out, dtype=torch.int64, device=device)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        new_in = [int(everyone[rank - 1][0]) if rank > 0 else 0, int(everyone[rank + 1][1]) if rank + 1 < world else 0]
        changed = torch.tensor([int(new_in != in_state)], dtype=torch.int64, device=device)
        dist.all_reduce(changed, op=dist.ReduceOp.MAX)
        in_state = new_in
        if not int(changed.item()):
            break
        if rounds > world + 2:
            raise RuntimeError("the chain states at the shard borders did not settle")
    return phase("finishing the pre-pass", backend.prepare_finish), (lo, hi), rounds


def sharded_prepare_in_process(backends, seed, num_pairs=0, coverage=0.0, ref_bias_mode=0, base_identifier=""):
    """sharded_prepare for `len(backends)` ranks that live in ONE process (one simulator per device, or a test): the same calls in lock
    step, the collectives replaced by sums and list lookups.  Returns (infos, block ranges, rounds)."""
    import numpy as np
    world = len(backends)
    infos = [b.prepare_plan(seed, num_pairs, coverage, ref_bias_mode, base_identifier) for b in backends]
    get = lambda i, k: i[k] if isinstance(i, dict) else getattr(i, k)
    weights = block_weights(backends[0].seq_len, get(infos[0], "insert_to"), backends[0].ref_seq_bias())
    ranges = partition_blocks(get(infos[0], "total_blocks"), world, weights)
    parts = [b.bias_partials(lo, hi) for b, (lo, hi) in zip(backends, ranges)]
    sums, maxes = np.sum([p[0] for p in parts], axis=0), np.sum([p[1] for p in parts], axis=0)
    for b in backends:
        b.prepare_normalization(sums, maxes)
    in_states, rounds = [[0, 0] for _ in range(world)], 0
    while True:
        outs = [b.prepare_sys_errors(lo, hi, st) for b, (lo, hi), st in zip(backends, ranges, in_states)]
        rounds += 1
        new_in = [[outs[r - 1][0] if r > 0 else 0, outs[r + 1][1] if r + 1 < world else 0] for r in range(world)]
        if new_in == in_states:
            break
        in_states = new_in
        if rounds > world + 2:
            raise RuntimeError("the chain states at the shard borders did not settle")
    return [b.prepare_finish() for b in backends], ranges, rounds
