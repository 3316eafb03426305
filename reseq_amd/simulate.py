"""Multi-GPU `illuminaPE` simulation: one process per GPU, launched as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m reseq_amd.simulate \\
        -R ref.fa -s profile.rsqp -1 r1.fq -2 r2.fq --numReads 100000000 --seed 11

The first rank of a host reads and packs the reference, variant and methylation files and the host's other ranks take the packed result from shared memory
(`load_once_per_host`).  Every rank packs the (small) profile tables, takes a contiguous range of 1000-position blocks balanced by expected pairs
(`sharding.partition_blocks`) and computes ITS share of the pre-passes (`sharding.sharded_prepare`: bias sums per chunk + one exact
all-reduce, systematic-error chains over its own positions + the chain states at the shard borders), simulates its blocks ONCE with the
FASTQ text kept in device memory (`rsq_sim_job_generate`), and after one all-gather of the text sizes writes it straight to its own
byte range of the two output files (`rsq_sim_job_write`: page-locked double buffers, several pwrite threads per file), all ranks at once --
no shard files, no second copy, nothing through Python; rank 0 appends the adapter-only pairs.  The result is byte for byte the output of a
single-GPU run (`reseq_amd/reseq illuminaPE` with the same arguments): blocks are independent and every random stream is keyed by
(seed, sequence, start, length), not by rank (Simulator.cpp:2384-2401 distributes blocks over threads the same way).
torch.distributed (RCCL) carries the seed, the job totals, the shard sizes and, after every step -- each phase of the pre-pass, generating, creating the files,
writing, appending -- whether any rank failed in it (in a barrier's place): no rank enters a collective that another rank will not reach.
"""
import argparse
import os
import sys
import time

from . import sharding


def _launcher_flags(ap):
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend; ranks that own a GPU each talk through RCCL (nccl) -- gloo carries the small "
                    "exchanges on the CPU when ranks share devices (--shareDevice)")
    ap.add_argument("--shareDevice", action="store_true", help="more ranks than devices: rank r runs on device r %% devices, the small exchanges go over gloo on the CPU (RCCL needs a "
                    "device per rank).  Every kernel, buffer and file write is the N-rank path's; several processes share a device's time, so this is not a scaling configuration -- "
                    "it is how the N-rank path is exercised on a host with one GPU")
    ap.add_argument("--distTimeout", type=int, default=600, help="seconds a rank waits in a collective for the others before it gives up (a rank that died takes the job with it)")


class Hooks:
    """What a caller of main() may replace -- nothing in the product does.  tests/simulate_under_test.py puts the host emulation of the kernels where the device would
    be (`make_backend`, `records_sim`, exchanges on the CPU) and lets a named rank die at a named step (`at_step`); none of that is reachable from this module's own
    command line or environment."""
    make_backend = None          # (args, seed, packed_from) -> what run_rank drives
    records_sim = None           # (args, seed) -> what run_records_rank drives
    on_cpu = False               # no device behind the ranks: the exchanges live on the CPU (gloo)
    banner = None                # said on stderr before anything else

    @staticmethod
    def at_step(step, rank):
        pass


def _check_backend(a, ap, hooks):
    if a.backend == "gloo" and not (a.shareDevice or hooks.on_cpu):
        ap.error("--backend gloo needs --shareDevice: ranks that own a GPU each talk through RCCL (nccl)")
    if a.shareDevice and a.backend != "gloo":
        ap.error("--shareDevice needs --backend gloo: RCCL cannot put two ranks on one device")
    if hooks.banner:
        print(hooks.banner, file=sys.stderr)


def _torch_first(hooks):
    """PyTorch before the library: both bring a HIP runtime, and the process must settle on the one torch.distributed (RCCL) and the tensors of the small exchanges
    use -- libreseq_amd.so then binds to the runtime that is loaded.  The other order leaves torch without devices ("No HIP GPUs are available")."""
    if not hooks.on_cpu:
        import torch  # noqa: F401


def _device_of(a, local_rank, hooks):
    """the HIP device of this rank"""
    if hooks.on_cpu:
        return 0
    if not a.shareDevice:
        return local_rank
    from . import api
    return local_rank % api.device_count()


def _start_ranks(a, local_rank, hooks):
    """(dist or None, the device the small exchanges live on).  Under a launcher -- also one that started a single rank -- the same exchanges over RCCL, or over gloo on
    the CPU when the ranks share devices."""
    on_cpu = hooks.on_cpu or a.shareDevice
    if "WORLD_SIZE" not in os.environ:
        return None, "cpu" if on_cpu else f"cuda:{local_rank}"
    import datetime
    import torch
    import torch.distributed as dist
    if on_cpu:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=a.distTimeout))
        return dist, "cpu"
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=a.distTimeout))
    return dist, f"cuda:{local_rank}"


def _profile_flags(ap):
    """the options main.cpp applies to a loaded profile (main.cpp:946-982, ProbabilityEstimates.h:1516-1549): the same names, the spelling of --errorMutliplier included"""
    ap.add_argument("-p", "--probabilitiesIn", dest="ipf", default=None, help="the .reseq.ipf archive that belongs to a .reseq profile (default: <statsIn>.ipf)")
    ap.add_argument("--ipfPrecision", type=float, default=5.0, help="precision the fitted tables must have reached (percent)")
    ap.add_argument("--errorMutliplier", type=float, default=1.0, help="the profile's substitution error rates times this factor")
    ap.add_argument("--noInDelErrors", action="store_true", help="simulate no insertions and deletions")
    ap.add_argument("--noSubstitutionErrors", action="store_true", help="simulate no substitution errors")


def _check_profile_flags(a, ap):
    if not a.ipfPrecision > 0.0:
        ap.error("ipfPrecision must be positive.")
    if a.noSubstitutionErrors and a.errorMutliplier != 1.0:          # main.cpp:946
        ap.error("noSubstitutionErrors and errorMutliplier cannot be combined.")


def load_edited_profile(a):
    """DataStats / ProbabilityEstimates loaded and edited as the command line does before the simulator sees them (reseq_main.cpp load_profile, main.cpp:964-982);
    every rank does the same to its own copy"""
    from . import api
    prof = api.load_profile(a.profile, a.ipf, a.ipfPrecision)
    try:
        if a.noInDelErrors:
            prof.remove_indel_errors()
        if a.noSubstitutionErrors:
            prof.remove_substitution_errors()
        elif a.errorMutliplier != 1.0:
            prof.change_error_rate(a.errorMutliplier)
    except Exception:
        prof.close()
        raise
    return prof


class GpuBackend:
    """reseq_amd.api.Simulator with reusable device buffers (the product path)."""

    def __init__(self, profile, fasta_path, device, replace_n_seed, vcf_path=None, methylation_path=None, sys_error_path=None, packed_from=None, ref_bias_file=None):
        """`profile`: a loaded api.Profile (owned from here on) or the path of one.  `packed_from`: a file another rank of this host wrote with export_reference -- the
        reference, variant and methylation files are not read again"""
        from . import api
        self.api = api
        self.prof = profile if isinstance(profile, api.Profile) else api.load_profile(profile)
        self.has_variants = bool(vcf_path)
        self.ref = self.sim = None
        try:
            if packed_from:
                self.sim = api.Simulator(self.prof, None, device)
                self.sim.import_reference(packed_from)
            else:
                self.ref = api.Reference(fasta_path, replace_n_seed)
                if vcf_path:
                    self.ref.read_variants(vcf_path)
                self.sim = api.Simulator(self.prof, self.ref, device)
                if methylation_path:
                    self.sim.read_methylation(methylation_path)
            if ref_bias_file:                                            # main.cpp:862-908: read by prepare, on every rank
                self.sim.set_ref_bias_file(ref_bias_file)
        except Exception:
            self.close()
            raise
        self.sys_error_path = sys_error_path
        self.device = device
        self.seq_len = self.sim.sequence_lengths()

    def create_sys_error_profile(self, seed, path):
        """--writeSysError (main.cpp:351-397): ONE rank draws the profile and writes it; all ranks then read it like a --readSysError file"""
        self.sim.create_sys_error_profile(seed, path)

    def export_reference(self, path):
        self.sim.export_reference(path)

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

    # the rank's share: simulated once, the text kept in device memory, then written to its place in the final files by the library's writer threads
    def job_generate(self, lo, hi, batch_blocks):
        return self.sim.job_generate(lo, hi, batch_blocks or 0)

    def job_compress(self):
        return self.sim.job_compress()

    def job_write(self, path1, offset1, path2, offset2):
        self.sim.job_write(path1, offset1, path2, offset2)
        self.sim.job_free()

    def adapter_only_pairs(self, first, n):
        return self.sim.adapter_only_pairs(first, n)

    # .gz outputs: text that does not come from the kept job (the adapter-only pairs) as members of the same kind, and what ends a file of such members
    def gzip_members(self, text):
        if not text:
            return b""
        if self.api.get_option("host_gzip"):
            return _gzip_member(text)
        return self.sim.gzip(text)

    def gzip_end(self):
        return self.sim.gzip_end()

    # --gatherOutput: a slice of the kept text as a device tensor of `size` bytes (the first `n` of them text), and a received slice to its place in a file
    def job_slice(self, file, at, n, size):
        import torch
        t = torch.zeros(size, dtype=torch.uint8, device=f"cuda:{self.device}")
        if n:
            self.sim.job_read(file, at, n, t.data_ptr())
        return t

    def write_slice(self, tensor, n, path, offset):
        self.api.dev_pwrite(self.device, tensor.data_ptr(), n, path, offset)

    def job_free(self):
        self.sim.job_free()

    def close(self):
        if self.sim:
            self.sim.close()
        if self.ref:
            self.ref.close()
        self.prof.close()


block_weights = sharding.block_weights


def part_name(path, rank, world):
    """--splitOutput: the file of rank `rank`; the parts in rank order, one after the other, are the single file"""
    return f"{path}.part{rank + 1:0{len(str(world))}d}of{world}"


def _agree(dist, device, error, what):
    """Every rank learns whether any rank failed in the step just done (one all-reduce, which also takes the place of a barrier): the failing rank raises its own
    exception, the others a RuntimeError -- nobody is left waiting in a collective for a rank that has gone"""
    failed = error is not None
    if dist is not None:
        import torch
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = bool(int(flag.item()))
    if error is not None:
        raise error
    if failed:
        raise RuntimeError(f"another rank failed while {what}")


def load_once_per_host(make_backend, dist, device, local_rank, local_world, node, shm_dir="/dev/shm"):
    """One load per host: the host's first rank reads and packs the reference, variant and methylation files (`make_backend(None)`) and exports the result to shared
    memory; the host's other ranks take it from there (`make_backend(path)`) -- rsq_sim_export_reference / rsq_sim_import_reference.  Without a process group, or
    with one rank per host, every rank loads for itself.  The ranks agree after each half, so that a loader that fails (a missing file, a malformed record) takes
    the others with it instead of leaving them waiting for a file that never appears."""
    if dist is None or local_world <= 1:
        return make_backend(None)
    import torch
    token = torch.tensor([int.from_bytes(os.urandom(7), "little")], dtype=torch.int64, device=device)       # one name for the job's files, from rank 0
    dist.broadcast(token, 0)
    path = os.path.join(shm_dir, f"rsq_packed_reference_{int(token.item()):014x}_host{node}")
    loader = local_rank == 0
    backend, error = None, None
    try:
        if loader:
            def load_and_export():
                b = make_backend(None)
                try:
                    b.export_reference(path)
                except Exception:
                    b.close()
                    raise
                return b
            backend, error = _attempt(load_and_export)
        _agree(dist, device, error, "loading and packing the reference for its host")
        if not loader:
            backend, error = _attempt(make_backend, path)
        _agree(dist, device, error, "taking the packed reference of its host")
    except BaseException:
        if backend is not None:                                      # some rank failed: this one gives its device memory back before the job ends
            backend.close()
        raise
    finally:
        if loader:
            for name in (path, path + ".writing"):                   # every rank of the host has it in its own memory by now -- or the job is over either way
                try:
                    os.unlink(name)
                except OSError:
                    pass
    return backend


def _attempt(f, *a):
    try:
        return f(*a), None
    except Exception as e:       # noqa: BLE001 -- handed to _agree, which raises it after the ranks have agreed
        return None, e


def gather_to_first_rank(backend, dist, rank, world, sizes, outs, slice_bytes, exchange_on_cpu=False):
    """The ranks' kept text merged by a collective (BASELINE.json's "RCCL all-gather over xGMI only to merge the emitted FASTQ buffers", as a gather: only one rank
    writes): per file and round every rank contributes one fixed-size slice of its text (rsq_sim_job_read), the first rank receives the N slices (dist.gather: RCCL
    over xGMI on a GPU host) and writes each to its rank's place in the file (rsq_dev_pwrite).  An option, not the default: N ranks writing their own byte ranges
    reach what one writer reaches, and this route moves every byte once more."""
    for f in (0, 1):
        start = [sum(row[f] for row in sizes[:r]) for r in range(world)]
        rounds = -(-max(row[f] for row in sizes) // slice_bytes)
        for k in range(rounds):
            at = k * slice_bytes
            mine = max(0, min(slice_bytes, sizes[rank][f] - at))
            send = backend.job_slice(f, at, mine, slice_bytes)
            # ranks that share devices exchange over gloo on the CPU (--shareDevice): the slice crosses to the host for the collective and back for the writer
            carried = send.cpu() if exchange_on_cpu and send.device.type != "cpu" else send
            recv = [carried.new_empty(slice_bytes) for _ in range(world)] if rank == 0 else None
            if dist is not None:
                dist.gather(carried, recv, dst=0)
            else:
                recv = [carried]
            if rank == 0:
                for r in range(world):
                    n = max(0, min(slice_bytes, sizes[r][f] - at))
                    if n:
                        backend.write_slice(recv[r].to(send.device), n, outs[f], start[r] + at)


def _gzip_member(text):
    """text as one gzip member to append to a compressed output (nothing for no text)"""
    import gzip
    return gzip.compress(text, 6) if text else b""


def run_rank(backend, dist, rank, world, out1, out2, seed, num_pairs=0, coverage=0.0, ref_bias_mode=0, base_identifier="", batch_blocks=None, device="cpu", split_output=False,
             gather_output=False, gather_slice_bytes=256 << 20, compress=False, at_step=Hooks.at_step):
    """One rank's share.  `backend` offers prepare (or the sharded pre-pass) / ref_seq_bias / seq_len / job_generate / job_write / adapter_only_pairs.
    Returns (pairs of the whole job, seconds of generation on the slowest rank).  This function is the launcher: it decides who does what and carries three
    small exchanges; the data never passes through Python."""
    if dist is not None and getattr(backend, "can_shard_prepare", False):   # every rank its share of the pre-passes (its phases agree among the ranks themselves)
        info, mine, _ = sharding.sharded_prepare(backend, dist, device, rank, world, seed, num_pairs, coverage, ref_bias_mode, base_identifier,
                                                 agree=lambda error, what: _agree(dist, device, error, what))
    else:
        def whole_prepare():
            info = backend.prepare(seed, num_pairs, coverage, ref_bias_mode, base_identifier)
            weights = block_weights(backend.seq_len, info["insert_to"], backend.ref_seq_bias())
            assert len(weights) == info["total_blocks"]
            return info, sharding.partition_blocks(info["total_blocks"], world, weights)[rank]
        prepared, error = _attempt(whole_prepare)
        _agree(dist, device, error, "preparing the simulation")
        info, mine = prepared
    t0 = time.perf_counter()
    at_step("generate", rank)
    generated, error = _attempt(backend.job_generate, mine[0], mine[1], batch_blocks)      # the rank's text stays where it was made (HBM) until its place is known
    _agree(dist, device, error, "generating its share")
    n_mine, bytes1, bytes2 = generated
    total_pairs, total_bytes, elapsed = sharding.job_totals(dist, device, n_mine, bytes1 + bytes2, time.perf_counter() - t0)
    # text from outside the kept job as members like the job's own (on the device: BGZF-framed; the emulation and host_gzip: zlib's), and the file's last bytes
    wrap = getattr(backend, "gzip_members", _gzip_member) if compress else (lambda text: text)
    file_end = getattr(backend, "gzip_end", lambda: b"")() if compress else b""
    if compress:
        # .gz outputs: the rank's text becomes gzip members in host memory (rsq_sim_job_compress: a pool of threads per rank); a file of concatenated members is a
        # gzip file, so from here on the COMPRESSED sizes are the shard sizes and everything else stays as it is.  The decompressed files are the single run's.
        packed, error = _attempt(backend.job_compress)
        _agree(dist, device, error, "compressing its share")
        bytes1, bytes2 = packed
    if split_output:
        # One pair of files per rank: buffered writes into ONE file take its inode lock one after the other, whoever writes (8 GB/s for the whole job however
        # many ranks, profiles/r03_g_*), separate files do not (49 GB/s with 8 writers behind one GPU's link).  No exchange of sizes, no barrier: a rank is done
        # when its text is out.  The last rank appends the adapter-only pairs, so that the parts in rank order concatenate to the single-file output.
        p1, p2 = part_name(out1, rank, world), part_name(out2, rank, world)

        def write_parts():
            for p in (p1, p2):
                open(p, "wb").close()
            backend.job_write(p1, 0, p2, 0)
            if rank == world - 1:
                with open(p1, "ab") as f1, open(p2, "ab") as f2:
                    for first in range(0, info["adapter_only_pairs"], 100000):
                        a, b = backend.adapter_only_pairs(first, min(100000, info["adapter_only_pairs"] - first))
                        f1.write(wrap(a))
                        f2.write(wrap(b))
                    f1.write(file_end)
                    f2.write(file_end)

        _agree(dist, device, _attempt(write_parts)[1], "writing its part")
        return int(total_pairs) + info["adapter_only_pairs"], elapsed
    # one all-gather of two lengths per rank: the exclusive scan over the ranks is every rank's offset in the final files
    sizes = sharding.gather_sizes(dist, device, [bytes1, bytes2], world)
    end1, end2 = (sum(row[col] for row in sizes) for col in (0, 1))
    def create_files():                                              # the files exist at the size of the ranks' text before anybody writes into them
        if rank == 0:
            for out, size in ((out1, end1), (out2, end2)):
                with open(out, "wb") as f:
                    f.truncate(size)

    def append_adapter_only_pairs():
        if rank == 0:
            with open(out1, "ab") as f1, open(out2, "ab") as f2:
                for first in range(0, info["adapter_only_pairs"], 100000):   # Simulator.cpp:2359-2382, as the single-GPU CLI does
                    a, b = backend.adapter_only_pairs(first, min(100000, info["adapter_only_pairs"] - first))
                    f1.write(wrap(a))
                    f2.write(wrap(b))
                f1.write(file_end)
                f2.write(file_end)

    # three steps, after each of which the ranks agree that all of them got through (what a barrier stood for, and no rank waits for one that failed)
    _agree(dist, device, _attempt(create_files)[1], "creating the output files")
    at_step("write", rank)
    if gather_output:                                                # one writer, fed by a collective
        def gathered():
            gather_to_first_rank(backend, dist, rank, world, sizes, (out1, out2), gather_slice_bytes, exchange_on_cpu=str(device) == "cpu")
            backend.job_free()
        _agree(dist, device, _attempt(gathered)[1], "gathering the text on the first rank")
    else:
        _agree(dist, device, _attempt(backend.job_write, out1, sum(row[0] for row in sizes[:rank]), out2, sum(row[1] for row in sizes[:rank]))[1],
               "writing its byte range")                             # all ranks at once, each its own byte range
    _agree(dist, device, _attempt(append_adapter_only_pairs)[1], "appending the adapter-only pairs")
    return int(total_pairs) + info["adapter_only_pairs"], elapsed


def run_records_rank(sim, dist, rank, world, input_path, output_path, device="cpu", split_output=False, count=None, compress=False, **pipeline):
    """`seqToIllumina` over several ranks (Simulator::SimulateErrorModelOnly, Simulator.cpp:2900-3014; SURVEY section 8(e): shards by input record ranges).
    Every rank counts the record starts in its stretch of the (plain) input file (`count`: api.count_fasta_records, host code), one all-gather of the two numbers
    tells it which bytes are its records and what the index of its first record is; it runs them through the library's pipeline with the text kept in device
    memory (`sim.error_model_file(..., keep_text=True)`), and after one all-gather of the text sizes writes it at its offset of the one output file
    (`sim.job_write`) -- or, split_output, into its own part file.  The output is byte for byte the single run's: a record's random stream is selected by its
    index in the input.  Returns (records of the whole job, seconds on the slowest rank)."""
    def counted():
        size = os.path.getsize(input_path)
        lo, hi = sharding.record_stretch(size, rank, world)
        return size, count(input_path, lo, hi) if hi > lo else (0, hi)
    got, error = _attempt(counted)
    _agree(dist, device, error, "counting the records of its stretch of the input")
    size, (n_starts, first_start) = got
    counts = sharding.gather_sizes(dist, device, [n_starts, first_start], world)
    begin, end, first_record = sharding.record_share(counts, size, rank)
    t0 = time.perf_counter()

    def simulated():
        if end <= begin and rank:                                    # no record starts in this rank's stretch
            return 0, 0
        return sim.error_model_file(input_path, None, from_=begin, to=end, first_record=first_record, keep_text=True, **pipeline)[:2]
    got, error = _attempt(simulated)
    _agree(dist, device, error, "simulating its records")
    records, nbytes = got
    total_records, _, elapsed = sharding.job_totals(dist, device, records, nbytes, time.perf_counter() - t0)
    if compress:                                                     # a .gz output: the share as gzip members, their size is the shard size (run_rank says why)
        packed, error = _attempt(lambda: sim.job_compress()[0] if nbytes else 0)
        _agree(dist, device, error, "compressing its share")
        nbytes = packed
    file_end = getattr(sim, "gzip_end", lambda: b"")() if compress else b""      # behind device-made members: BGZF's end-of-file member
    if not nbytes:                                                   # nothing kept (an empty share): nothing to write either
        write = lambda path, offset: None
    else:
        write = lambda path, offset: sim.job_write(path, offset, None, 0)
    if split_output:
        part = part_name(output_path, rank, world)

        def write_part():
            open(part, "wb").close()
            write(part, 0)
            if nbytes:
                sim.job_free()
            if rank == world - 1 and file_end:
                with open(part, "ab") as f:
                    f.write(file_end)
        _agree(dist, device, _attempt(write_part)[1], "writing its part")
        return int(total_records), elapsed
    sizes = sharding.gather_sizes(dist, device, [nbytes], world)

    def create_file():
        if rank == 0:
            with open(output_path, "wb") as f:
                f.truncate(sum(row[0] for row in sizes))
    _agree(dist, device, _attempt(create_file)[1], "creating the output file")
    _agree(dist, device, _attempt(write, output_path, sum(row[0] for row in sizes[:rank]))[1], "writing its byte range")
    if nbytes:
        sim.job_free()

    def end_file():
        if rank == 0 and file_end:
            with open(output_path, "ab") as f:
                f.write(file_end)
    _agree(dist, device, _attempt(end_file)[1], "ending the output file")
    return int(total_records), elapsed


def _broadcast_seed(a, dist, device):
    """one seed for the whole job, all 64 bits of it"""
    seed = (a.seed if a.seed is not None else int.from_bytes(os.urandom(8), "little")) & 0xFFFFFFFFFFFFFFFF
    if dist is not None:
        import torch
        t = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed], dtype=torch.int64, device=device)
        dist.broadcast(t, 0)
        seed = int(t.item()) & 0xFFFFFFFFFFFFFFFF
    return seed


def main_records(argv, hooks=Hooks):
    """python -m reseq_amd.simulate seqToIllumina -i in.fa -o out.fq -s profile [--seed N] [--splitOutput]: `reseq seqToIllumina` over the GPUs of a host"""
    ap = argparse.ArgumentParser(prog="reseq_amd.simulate seqToIllumina", description=main_records.__doc__)
    ap.add_argument("-i", "--input", required=True, help="FASTA records with the systematic errors in their id lines; a plain file (ranks read it at offsets)")
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("-s", "--statsIn", dest="profile", required=True)
    _profile_flags(ap)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--splitOutput", action="store_true", help="every rank writes its own file <out>.part<k>of<N> (their concatenation in order is the single file)")
    _launcher_flags(ap)
    a = ap.parse_args(argv)
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if a.output.endswith(".bz2"):
        ap.error(f"{a.output}: bzip2 output is not supported by the multi-GPU launcher (write .gz or plain FASTQ)")
    _check_profile_flags(a, ap)
    _check_backend(a, ap, hooks)
    _torch_first(hooks)
    from . import api
    dist, device = _start_ranks(a, local_rank, hooks)
    seed = _broadcast_seed(a, dist, device)
    prof = sim = None
    try:
        def set_up():
            if hooks.records_sim:
                return None, hooks.records_sim(a, seed)
            p = load_edited_profile(a)
            try:
                s = api.Simulator(p, None, _device_of(a, local_rank, hooks))
                s.prepare(seed)
            except Exception:
                p.close()
                raise
            return p, s
        got, error = _attempt(set_up)
        _agree(dist, device, error, "setting up its simulator")
        prof, sim = got
        records, seconds = run_records_rank(sim, dist, rank, world, a.input, a.output, device, a.splitOutput, count=api.count_fasta_records, compress=a.output.endswith(".gz"))
        if rank == 0:
            if not records:
                print(f"!!! Error: {a.input} does not contain any sequences.", file=sys.stderr)
                if not a.splitOutput:
                    os.remove(a.output)
                sys.exit(1)
            print(f">>> Info: Generated {records} reads on {world} GPU(s), {seconds:.2f} s on the slowest rank", file=sys.stderr)
    finally:
        if sim is not None:
            sim.close()
        if prof is not None:
            prof.close()
        if dist is not None:
            dist.destroy_process_group()


def main(argv=None, hooks=Hooks):
    argv = sys.argv[1:] if argv is None else list(argv)
    if argv and argv[0] in ("seqToIllumina", "replaceQuals"):
        return main_records(argv[1:], hooks)
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("-R", "--refSim", "-r", "--refIn", dest="ref", required=True)
    ap.add_argument("-s", "--statsIn", dest="profile", required=True)
    _profile_flags(ap)
    ap.add_argument("-1", "--firstReadsOut", dest="out1", default="reseq-R1.fq")
    ap.add_argument("-2", "--secondReadsOut", dest="out2", default="reseq-R2.fq")
    ap.add_argument("-V", "--vcfSim", dest="vcf", default=None, help="variants to simulate per allele (substitutions)")
    ap.add_argument("--methylation", default=None, help="extended bed graph with methylation values per region (and allele)")
    ap.add_argument("--readSysError", default=None, help="systematic-error profile written by reseq illuminaPE --writeSysError")
    ap.add_argument("--writeSysError", default=None, help="draw the systematic errors once (the first rank), write them to this file and simulate with them")
    ap.add_argument("--numReads", type=int, default=0)
    ap.add_argument("-c", "--coverage", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--refBias", choices=["keep", "no", "draw", "file"], default=None)
    ap.add_argument("--refBiasFile", default=None, help="reference sequence biases, one `<sequence id> <bias>` per line (implies --refBias file)")
    ap.add_argument("--recordBaseIdentifier", default="ReseqRead")
    ap.add_argument("--batchBlocks", type=int, default=0, help="blocks of 1000 start positions per device call (default: about 12 M pairs)")
    ap.add_argument("--gatherOutput", action="store_true", help="the ranks' text is gathered on the first rank by a collective (RCCL) in slices and written by that rank alone, "
                    "instead of every rank writing its own byte range of the files")
    ap.add_argument("--gatherSliceMB", type=int, default=256, help="--gatherOutput: bytes (MiB) per rank and round of the gather")
    ap.add_argument("--gatherSliceBytes", type=int, default=0, help="--gatherOutput: the same in bytes (small jobs, tests); overrides --gatherSliceMB")
    ap.add_argument("--everyRankLoads", action="store_true", help="every rank reads and packs the reference, variant and methylation files itself (default: the first rank of a host "
                    "does and the host's other ranks take the packed result from shared memory)")
    ap.add_argument("--splitOutput", action="store_true", help="every rank writes its own pair of files <out>.part<k>of<N> (their concatenation in order is the single file): "
                    "writes into one file serialise on its inode lock, 8 GB/s for the whole job; separate files scale with the ranks")
    _launcher_flags(ap)
    a = ap.parse_args(argv)
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    _check_profile_flags(a, ap)
    _check_backend(a, ap, hooks)
    # main.cpp:862-908: --refBias keep|no|draw|file, --refBiasFile implies file; keep is the default
    if a.refBias is None:
        a.refBias = "file" if a.refBiasFile else "keep"
    elif (a.refBias == "file") != bool(a.refBiasFile):
        ap.error("refBiasFile option mandatory if refBias is set to 'file'." if a.refBias == "file" else "refBiasFile option only allowed if refBias is set to 'file'.")
    if a.writeSysError and a.readSysError:                           # main.cpp:351-397
        ap.error("writeSysError and readSysError option are mutually exclusive. Specify the one or the other.")
    if a.numReads and a.coverage:                                    # main.cpp:783-786
        ap.error("numReads and coverage option are mutually exclusive. Specify the one or the other.")
    compress = a.out1.endswith(".gz")                                # gzip members concatenate, so a rank's compressed share has a place in the file like plain text
    if a.out2.endswith(".gz") != compress:
        ap.error("the two output files are either both plain or both .gz")
    if compress and a.gatherOutput:
        ap.error("--gatherOutput moves plain text; with .gz outputs every rank writes its own compressed share")
    for out in (a.out1, a.out2):
        if out.endswith(".bz2"):
            ap.error(f"{out}: bzip2 output is not supported by the multi-GPU launcher (write .gz or plain FASTQ)")
    _torch_first(hooks)
    dist, device = _start_ranks(a, local_rank, hooks)
    seed = _broadcast_seed(a, dist, device)
    if hooks.make_backend:
        make = lambda packed_from: hooks.make_backend(a, seed, packed_from)
    else:
        hip_device = _device_of(a, local_rank, hooks)
        make = lambda packed_from: GpuBackend(load_edited_profile(a), a.ref, hip_device, seed, a.vcf, a.methylation, a.readSysError, packed_from=packed_from,
                                              ref_bias_file=a.refBiasFile)
    if a.everyRankLoads:
        backend = make(None)
    else:
        backend = load_once_per_host(make, dist, device, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), int(os.environ.get("GROUP_RANK", 0)),
                                     shm_dir=os.environ.get("RSQ_SHM_DIR", "/dev/shm"))
    try:
        if a.writeSysError:
            # main.cpp:351-397 WriteSysError: the first rank draws and writes the profile, then every rank reads it as it would a --readSysError file
            def write_profile():
                if rank == 0:
                    print(f">>> Info: Writing systematic error profile to {a.writeSysError}", file=sys.stderr)
                    try:
                        backend.create_sys_error_profile(seed, a.writeSysError)
                    except Exception:
                        if os.path.exists(a.writeSysError):          # main.cpp:392
                            os.remove(a.writeSysError)
                        raise
            _agree(dist, device, _attempt(write_profile)[1], "writing the systematic error profile")
            backend.sys_error_path = a.writeSysError
        pairs, seconds = run_rank(backend, dist, rank, world, a.out1, a.out2, seed, a.numReads, a.coverage, {"keep": 0, "no": 1, "draw": 2, "file": 3}[a.refBias],
                                  a.recordBaseIdentifier, a.batchBlocks, device, a.splitOutput, a.gatherOutput, a.gatherSliceBytes or a.gatherSliceMB << 20, compress,
                                  at_step=hooks.at_step)
        if rank == 0:
            print(f">>> Info: Generated {pairs} read pairs on {world} GPU(s), {seconds:.2f} s of generation on the slowest rank", file=sys.stderr)
    finally:
        backend.close()
        if dist is not None:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
