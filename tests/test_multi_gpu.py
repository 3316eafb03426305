"""The N-rank path, one process per rank under torch.distributed.run, in three modes of the SAME test bodies:

  rccl    (-m gpu; skipped with a reason when fewer than two devices are visible): the product -- `python -m reseq_amd.simulate` on N GPUs over RCCL, compared byte
          for byte with the single-device command line (`reseq illuminaPE` / `reseq seqToIllumina`, Simulator.cpp:2830-2836 starts its own workers the same way and
          :2384-2401 hands them blocks), run with 2 ranks and with min(8, devices) ranks;
  shared  (-m gpu; needs ONE device): the same launcher and the same kernels with 2 and 4 PROCESSES on device 0 (`--backend gloo --shareDevice`: the ranks' small
          exchanges go over gloo on the CPU, because RCCL cannot put two ranks on one device).  Everything of the N-rank path except RCCL itself runs on hardware:
          the packed reference exported to /dev/shm by one process and imported by the others, sharded pre-passes across real shard borders under separate
          processes, N writers into one file at offsets, N compilations racing on one kernel cache, a rank killed with its device context open;
  gloo    (CPU suite): the launcher under tests/simulate_under_test.py --emulate -- tests/hostemu where the device would be (tests/emu_ranks.py) -- compared with
          the same module run as one rank without a launcher.

What is covered: (i) illuminaPE plain, with variants + methylation, --splitOutput, --gatherOutput, .gz; (ii) seqToIllumina; (iii) the sharded pre-pass against the
whole one; (iv) one load per host: only one rank ever opens the FASTA; (v) bench.py --gpus 2; (vi) a rank killed in the middle of the job takes the job with it;
(vii) the options main.cpp applies after loading (error multiplier, no indels / substitutions, --refBiasFile, --writeSysError) on N ranks against the command line."""
import gzip
import json
import os
import pathlib
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

import parity_cases as P
from conftest import read_fasta as conftest_read_fasta
from reseq_amd import api, synth

HERE = pathlib.Path(__file__).resolve().parent
ROOT = HERE.parent
RESEQ = ROOT / "reseq_amd" / "reseq"


def _devices():
    try:
        return api.device_count()
    except Exception:      # noqa: BLE001 -- no library, no device: the gloo mode does not need one
        return 0


def _worlds(mode):
    if mode == "gloo":
        return [2]
    if mode == "shared":
        return [2, 4]
    n = _devices()
    return sorted({2, min(8, n)})


@pytest.fixture(params=["gloo", pytest.param("shared", marks=pytest.mark.gpu), pytest.param("rccl", marks=pytest.mark.gpu)])
def mode(request):
    if request.param == "rccl" and _devices() < 2:
        pytest.skip(f"the N-rank path over RCCL needs at least two visible devices ({_devices()} here); its bodies run with the real kernels in mode `shared` (N processes on "
                    "one device, exchanges over gloo) and over gloo with the host emulation in the CPU suite")
    if request.param == "shared" and _devices() < 1:
        pytest.skip("no device")
    return request.param


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(workdir, **extra):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    e.update(PYTHONPATH=str(ROOT) + os.pathsep + e.get("PYTHONPATH", ""), RSQ_TESTS=str(HERE), HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update({k: str(v) for k, v in extra.items()})
    return e


UNDER_TEST = (str(HERE / "simulate_under_test.py"),)                # reseq_amd.simulate.main with the suite's hooks (emulation, a rank that dies)


def launch(mode, world, args, workdir, target=None, timeout=900, check=True, **env):
    """the target under torch.distributed.run with `world` ranks on this host.  The default target is the product's own entry point, `-m reseq_amd.simulate`, for the
    modes with a device behind the ranks and tests/simulate_under_test.py for the emulation."""
    if target is None:
        target = UNDER_TEST if mode == "gloo" else ("-m", "reseq_amd.simulate")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_port()), *target, *map(str, args)]
    if mode == "gloo":
        cmd += ["--backend", "gloo", "--emulate"]
        env.setdefault("RSQ_SHM_DIR", workdir)                      # the packed reference of the "host" goes where the test can see that it is gone
    elif mode == "shared":
        cmd += ["--backend", "gloo", "--shareDevice"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=_env(workdir, **env), cwd=str(ROOT))
    if check:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    return r


def single(mode, sub, args, workdir, **env):
    """the single-device run the ranks' output is compared with: the command line on one GPU, or the launcher's module as one rank on the emulation"""
    if mode == "gloo":
        cmd = [sys.executable, *UNDER_TEST, *([sub] if sub == "seqToIllumina" else []), *map(str, args), "--emulate"]
    else:
        cmd = [str(RESEQ), sub, *map(str, args)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(workdir, **env), cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    return r


@pytest.fixture(scope="module")
def job(tmp_path_factory):
    """TINY profile (three tiles, adapters, variable read lengths, indels), three sequences of which one is too short for blocks; variants of every kind on two
    alleles with extra starts across block borders, and methylation regions with a rate per allele"""
    work = tmp_path_factory.mktemp("multi_gpu")
    ppath, fpath, seqs = P.make_inputs(work, "job", synth.TINY, [7000, 80, 4210])
    vcf = work / "job.vcf"
    P.write_vcf(vcf, seqs, P._mixed_variant_set(seqs, np.random.default_rng(5), 30, [999, 1000, 1999, 2000, 2999, 3000]))
    names = [n.split(" ")[0] for n, _ in seqs]
    bed = work / "job.bed"
    bed.write_text(f"{names[0]}\t100\t900\t0.3\t0.6\n{names[0]}\t2000\t5000\t0.0\t0.5\n{names[2]}\t50\t2000\t0.0\t1.0\n")
    # without the short sequence: a systematic-error profile is written for every sequence and read for those with blocks only (Simulator.cpp:750-769)
    ppath2, fpath2, _ = P.make_inputs(work, "long_only", synth.TINY, [7000, 4210])
    return dict(work=work, profile=ppath, fasta=fpath, vcf=str(vcf), bed=str(bed), long_only=dict(work=work, profile=ppath2, fasta=fpath2))


def _pe_args(job, tag, extra=(), gz=False):
    ext = ".fq.gz" if gz else ".fq"
    out = [str(job["work"] / f"{tag}_{k}{ext}") for k in (1, 2)]
    return ["-R", job["fasta"], "-s", job["profile"], "-1", out[0], "-2", out[1], "--numReads", 40000, "--seed", 7, "--refBias", "no", "--recordBaseIdentifier", "Job", *extra], out


def _single_pe(mode, job, tag, extra=()):
    args, out = _pe_args(job, f"{'emu' if mode == 'gloo' else 'cli'}_{tag}_one", extra)
    if not all(os.path.exists(o) for o in out):
        single(mode, "illuminaPE", args, job["work"])
    texts = [open(o, "rb").read() for o in out]
    assert texts[0].count(b"\n") % 4 == 0 and texts[0].count(b"\n") == texts[1].count(b"\n") > 4 * 5000 and b":0:Adapter:0:" in texts[0]
    return texts


# ------------------------------------------------------------------------------------------------------------------------ (i)
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("inputs", ["plain", "variants_methylation"])
def test_illumina_pe_on_n_ranks_writes_the_single_device_files(mode, job, inputs):
    _illumina_pe(mode, job, inputs, _worlds(mode))


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_one_rank_under_the_launcher_writes_the_command_line_files(job):
    """What a box with ONE device can run of the RCCL mode: the launcher with a single rank -- process group over nccl, every exchange of the N-rank path with a
    world of one, GpuBackend, the job kept in device memory and written at its offset -- against the command line, for both commands."""
    if _devices() < 1:
        pytest.skip("no device")
    _illumina_pe("rccl", job, "variants_methylation", [1])
    _seq_to_illumina("rccl", job, [1])


def _illumina_pe(mode, job, inputs, worlds):
    extra = ["-V", job["vcf"], "--methylation", job["bed"]] if inputs != "plain" else []
    want = _single_pe(mode, job, inputs, extra)
    if inputs != "plain":
        assert b"_allele1" in want[0]
    for world in worlds:
        tag = f"{mode}_{inputs}_w{world}"
        # every rank writes its own byte range of the two files (batches of three blocks: several device calls per rank)
        args, out = _pe_args(job, tag, [*extra, "--batchBlocks", 3])
        r = launch(mode, world, args, job["work"])
        assert f"on {world} GPU(s)" in r.stderr
        assert [open(o, "rb").read() for o in out] == want, (world, "byte ranges")
        # --splitOutput: a pair of files per rank whose concatenation in rank order is the single output
        args, out = _pe_args(job, tag + "_split", [*extra, "--splitOutput"])
        launch(mode, world, args, job["work"])
        for o, w in zip(out, want):
            digits = len(str(world))
            assert b"".join(open(f"{o}.part{k + 1:0{digits}d}of{world}", "rb").read() for k in range(world)) == w, (world, "split")
        # --gatherOutput: the ranks' text gathered on the first rank by a collective in slices (several rounds, last slices of different lengths), one writer
        args, out = _pe_args(job, tag + "_gather", [*extra, "--gatherOutput", "--gatherSliceBytes", 300_000])
        launch(mode, world, args, job["work"])
        assert [open(o, "rb").read() for o in out] == want, (world, "gathered")
        # .gz: every rank's share as gzip members at its offset (the sizes exchanged are the compressed ones), the adapter-only pairs as a member behind them
        args, out = _pe_args(job, tag, extra, gz=True)
        launch(mode, world, args, job["work"])
        assert [gzip.decompress(open(o, "rb").read()) for o in out] == want, (world, "gz")
        if mode != "gloo":                                          # device-made members are BGZF blocks: the file ends with BGZF's end-of-file member
            assert all(open(o, "rb").read().endswith(api.gzip_eof_member()) for o in out)
    assert not list(pathlib.Path(job["work"]).glob("rsq_packed_reference_*")) and not list(pathlib.Path("/dev/shm").glob("rsq_packed_reference_*"))


# ----------------------------------------------------------------------------------------------------------------------- (ii)
@pytest.mark.timeout(1800)
def test_seq_to_illumina_on_n_ranks_writes_the_single_device_file(mode, job):
    _seq_to_illumina(mode, job, _worlds(mode))


def _seq_to_illumina(mode, job, worlds):
    n = 600 if mode == "gloo" else 40000
    arrays = synth.make_profile(synth.TINY, seed=5)
    rec = synth.make_error_model_input(9, n, 30, arrays, zero_frac=0.7)
    r = rec["rate"].astype(np.int64)                          # what survives the file: odd percents above 86 become the even one below (Simulator.cpp:2439-2442)
    rec["rate"] = np.where(r > 86, r - r % 2, r).astype(np.uint8)
    ids = [f"read {i}/x" if i % 7 == 0 else f"r{i}" for i in range(n)]
    fa = job["work"] / f"{mode}_records.fa"
    fa.write_bytes(P.fasta_of_records(rec, ids, wrap_every=3))
    one = job["work"] / f"{mode}_records_one.fq"
    single(mode, "seqToIllumina", ["-i", fa, "-o", one, "-s", job["profile"], "--seed", 13], job["work"])
    want = one.read_bytes()
    assert want.count(b"\n") == 4 * n and want.startswith(b"@read 0/x ")
    for world in worlds:
        out = job["work"] / f"{mode}_records_w{world}.fq"
        r = launch(mode, world, ["seqToIllumina", "-i", fa, "-o", out, "-s", job["profile"], "--seed", 13], job["work"])
        assert f"Generated {n} reads on {world} GPU(s)" in r.stderr
        assert out.read_bytes() == want, world
        launch(mode, world, ["seqToIllumina", "-i", fa, "-o", out, "-s", job["profile"], "--seed", 13, "--splitOutput"], job["work"])
        digits = len(str(world))
        assert b"".join(open(f"{out}.part{k + 1:0{digits}d}of{world}", "rb").read() for k in range(world)) == want, world
        gz = job["work"] / f"{mode}_records_w{world}.fq.gz"
        launch(mode, world, ["seqToIllumina", "-i", fa, "-o", gz, "-s", job["profile"], "--seed", 13], job["work"])
        assert gzip.decompress(gz.read_bytes()) == want, world
        assert mode == "gloo" or gz.read_bytes().endswith(api.gzip_eof_member())
    # a malformed record on one rank: that rank says what the reference's reader says, the job ends, nobody waits
    text = fa.read_bytes()
    cut = text.index(b">", len(text) * 3 // 4)
    end = text.index(b"\n", cut)
    fields = text[cut:end].rsplit(b" ", 1)
    bad = job["work"] / f"{mode}_records_bad.fa"
    bad.write_bytes(text[:cut] + fields[0] + b" 3" + fields[1][1:] + text[end:])
    r = launch(mode, max(worlds), ["seqToIllumina", "-i", bad, "-o", job["work"] / f"{mode}_bad.fq", "-s", job["profile"], "--seed", 13], job["work"], check=False, timeout=600)
    assert r.returncode != 0 and ("Template segment" in r.stderr or "malformed record" in r.stderr), r.stderr[-3000:]
    assert max(worlds) == 1 or "another rank failed while simulating its records" in r.stderr


# ---------------------------------------------------------------------------------------------------------------------- (iii)
@pytest.mark.timeout(1800)
def test_sharded_pre_pass_equals_the_whole_pre_pass(mode, job):
    for world in sorted({*_worlds(mode), 4} if mode == "gloo" else set(_worlds(mode))):
        r = launch(mode, world, ["prepass", job["work"] / f"{mode}_prepass_w{world}"], job["work"], target=(str(HERE / "multi_gpu_worker.py"),))
        assert "PREPASS_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


# ----------------------------------------------------------------------------------------------------------------------- (iv)
OPEN_LOG_C = r"""
// LD_PRELOAD shim of the test: every open of the file RSQ_OPEN_WATCH names appends "<pid>\n" to the file RSQ_OPEN_LOG names (inotify would do, but it merges the
// identical events of two ranks that open the file at the same moment).  open / open64 / openat / fopen and their 64-bit names: whatever the readers use.
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void note(const char *path) {
    const char *watch = getenv("RSQ_OPEN_WATCH"), *log = getenv("RSQ_OPEN_LOG");
    if (!path || !watch || !log || strcmp(path, watch)) return;
    static int (*real_open)(const char *, int, ...);
    if (!real_open) real_open = (int (*)(const char *, int, ...))dlsym(RTLD_NEXT, "open");
    const int fd = real_open(log, O_WRONLY | O_APPEND | O_CREAT, 0644);
    if (fd < 0) return;
    char line[32];
    const int n = snprintf(line, sizeof line, "%d\n", (int)getpid());
    if (write(fd, line, n) != n) {}
    close(fd);
}
#define OPEN_LIKE(name)                                                                   \
    int name(const char *path, int flags, ...) {                                          \
        static int (*real)(const char *, int, ...);                                       \
        if (!real) real = (int (*)(const char *, int, ...))dlsym(RTLD_NEXT, #name);       \
        va_list ap;                                                                       \
        va_start(ap, flags);                                                              \
        const int mode = va_arg(ap, int);                                                 \
        va_end(ap);                                                                       \
        note(path);                                                                       \
        return real(path, flags, mode);                                                   \
    }
OPEN_LIKE(open)
OPEN_LIKE(open64)
#define OPENAT_LIKE(name)                                                                 \
    int name(int dir, const char *path, int flags, ...) {                                 \
        static int (*real)(int, const char *, int, ...);                                  \
        if (!real) real = (int (*)(int, const char *, int, ...))dlsym(RTLD_NEXT, #name);  \
        va_list ap;                                                                       \
        va_start(ap, flags);                                                              \
        const int mode = va_arg(ap, int);                                                 \
        va_end(ap);                                                                       \
        note(path);                                                                       \
        return real(dir, path, flags, mode);                                              \
    }
OPENAT_LIKE(openat)
OPENAT_LIKE(openat64)
#define FOPEN_LIKE(name)                                                                  \
    FILE *name(const char *path, const char *mode) {                                      \
        static FILE *(*real)(const char *, const char *);                                 \
        if (!real) real = (FILE * (*)(const char *, const char *)) dlsym(RTLD_NEXT, #name); \
        note(path);                                                                       \
        return real(path, mode);                                                          \
    }
FOPEN_LIKE(fopen)
FOPEN_LIKE(fopen64)
"""


def _open_log_shim(work):
    so = pathlib.Path(work) / "openlog.so"
    if not so.exists():
        (pathlib.Path(work) / "openlog.c").write_text(OPEN_LOG_C)
        subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", str(so), str(pathlib.Path(work) / "openlog.c"), "-ldl"], check=True)
    return str(so)


@pytest.mark.timeout(1800)
def test_only_one_rank_of_a_host_opens_the_reference(mode, job):
    """one load per host (simulate.load_once_per_host): with N ranks on the host ONE process opens the FASTA (and the VCF), the others take the packed reference
    from shared memory; with --everyRankLoads all N do.  Who opened what is logged by an LD_PRELOAD shim of the test."""
    world = _worlds(mode)[-1]
    shim = _open_log_shim(job["work"])
    want = _single_pe(mode, job, "variants_methylation", ["-V", job["vcf"], "--methylation", job["bed"]])
    openers = {}
    for how, extra in (("shared", []), ("each", ["--everyRankLoads"])):
        for watched in ("fasta", "vcf"):
            log = pathlib.Path(job["work"]) / f"{mode}_opens_{how}_{watched}.log"
            log.unlink(missing_ok=True)
            args, out = _pe_args(job, f"{mode}_opens_{how}", ["-V", job["vcf"], "--methylation", job["bed"], *extra])
            launch(mode, world, args, job["work"], LD_PRELOAD=shim, RSQ_OPEN_WATCH=job[watched], RSQ_OPEN_LOG=log)
            assert [open(o, "rb").read() for o in out] == want
            openers[how, watched] = {int(x) for x in log.read_text().split()} if log.exists() else set()
    for watched in ("fasta", "vcf"):
        assert len(openers["shared", watched]) == 1, f"{world} ranks of one host: the {watched} file was opened by the processes {openers['shared', watched]}, one must load for all"
        assert len(openers["each", watched]) == world, (watched, openers["each", watched])


# ------------------------------------------------------------------------------------------------------------------------ (v)
@pytest.mark.timeout(1800)
def test_bench_on_two_ranks(mode, job):
    flags = ["--gpus", "2", "--steps", "2", "--warmup", "1"] + (["--backend", "gloo", "--emulate"] if mode == "gloo" else ["--pairs", "1000000", "--no-cpu-baseline"])
    if mode == "shared":
        flags += ["--backend", "gloo", "--shareDevice", "--genome", "1000000", "--pairs", "400000"]
    for scaling in ("weak", "strong"):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags, "--scaling", scaling], capture_output=True, text=True, timeout=1500, env=_env(job["work"]), cwd=str(ROOT))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-5000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == scaling and len(line["ms_per_step_per_rank"]) == 2
        assert line["ms_per_step"] >= max(line["ms_per_step_per_rank"]) * 0.999 and line["value"] > 0
        assert "barrier" in line["config"]["collectives"] and ("nccl" if mode == "rccl" else "gloo") in line["config"]["collectives"]
        if mode == "shared":
            assert line["config"]["ranks_share_devices"] and "not a scaling" in line["config"]["note"]
        if mode != "gloo":
            assert "roofline" in line and line["roofline"]["frac"] > 0


# ----------------------------------------------------------------------------------------------------------------------- (vi)
@pytest.mark.timeout(900)
@pytest.mark.parametrize("step", ["generate", "write"])
def test_a_rank_that_dies_takes_the_job_with_it(mode, job, step):
    """SIGKILL of one rank at the named step (simulate._fault): the launcher ends the others, the command returns non-zero well inside the collective timeout, no rank
    stays behind and the packed reference is gone from shared memory"""
    args, out = _pe_args(job, f"{mode}_killed_{step}")
    t0 = time.time()
    r = launch(mode, 2, [*args, "--distTimeout", 120], job["work"], target=UNDER_TEST, check=False, timeout=600, RSQ_FAULT_INJECT=f"{step}:1")
    took = time.time() - t0
    assert r.returncode != 0, r.stderr[-3000:]
    assert took < 240, took
    assert "Generated" not in r.stderr
    leftover = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert not [l for l in leftover.splitlines() if f"{mode}_killed_{step}" in l and "ps -eo" not in l], leftover
    assert not list(pathlib.Path(job["work"]).glob("rsq_packed_reference_*")) and not list(pathlib.Path("/dev/shm").glob("rsq_packed_reference_*"))


# ---------------------------------------------------------------------------------------------------------------------- (vii)
@pytest.mark.timeout(1800)
def test_post_load_options_on_n_ranks_write_the_single_device_files(mode, job):
    """main.cpp:964-982 (--errorMutliplier, --noInDelErrors, --noSubstitutionErrors; ProbabilityEstimates.h:1516-1549), :862-908 (--refBiasFile) and :351-397
    (--writeSysError): what `reseq illuminaPE` does with them on one device, the launcher does on every rank -- the files are the command line's byte for byte, and
    every option changes them"""
    plain = _single_pe(mode, job, "plain")
    names = [n.split(" ")[0] for n, _ in conftest_read_fasta(job["fasta"])]
    bias = job["work"] / "job_ref_bias.txt"
    # the forms of test/ref-bias-test.txt: a leading '>', a comment behind the name, a name that is no sequence; every sequence needs a line
    bias.write_text(f">{names[2]} something   0.25\nnot_a_sequence 0.5\n{names[0]} 2.0\n{names[1]} 1.0\n")
    world = _worlds(mode)[-1]
    cases = {
        "multiplier_no_indels": ["--errorMutliplier", 2.5, "--noInDelErrors"],
        "no_substitutions": ["--noSubstitutionErrors"],
        "ref_bias_file": ["--refBiasFile", bias],
    }
    for tag, flags in cases.items():
        extra = flags
        base = [a for a in _pe_args(job, "unused")[0]]
        if tag == "ref_bias_file":                                  # --refBias no of the common arguments and --refBiasFile exclude each other
            k = base.index("--refBias")
            del base[k:k + 2]
        one = [str(job["work"] / f"{'emu' if mode == 'gloo' else 'cli'}_{tag}_one_{k}.fq") for k in (1, 2)]
        many = [str(job["work"] / f"{mode}_{tag}_w{world}_{k}.fq") for k in (1, 2)]

        def with_outputs(args, outs):
            args = list(args)
            args[args.index("-1") + 1], args[args.index("-2") + 1] = outs
            return args
        if not all(os.path.exists(o) for o in one):
            single(mode, "illuminaPE", with_outputs([*base, *extra], one), job["work"])
        want = [open(o, "rb").read() for o in one]
        assert want != plain and want[0].count(b"\n") > 4 * 5000, tag
        launch(mode, world, with_outputs([*base, *extra, "--batchBlocks", 3], many), job["work"])
        assert [open(o, "rb").read() for o in many] == want, tag
    # --writeSysError: the first rank draws and writes the profile, every rank then simulates with it; file and reads are the command line's
    long_only = job["long_only"]
    args, out = _pe_args(long_only, f"{mode}_write_sys_one", ["--writeSysError", job["work"] / f"{mode}_sys_one.prof"])
    single(mode, "illuminaPE", args, job["work"])
    want = [open(o, "rb").read() for o in out]
    args, out = _pe_args(long_only, f"{mode}_write_sys_w{world}", ["--writeSysError", job["work"] / f"{mode}_sys_w{world}.prof"])
    launch(mode, world, args, job["work"])
    assert (job["work"] / f"{mode}_sys_w{world}.prof").read_bytes() == (job["work"] / f"{mode}_sys_one.prof").read_bytes()
    assert [open(o, "rb").read() for o in out] == want and want[0].count(b"\n") > 4 * 5000
    # and --readSysError of that file gives the same reads again
    args, out = _pe_args(long_only, f"{mode}_read_sys_w{world}", ["--readSysError", job["work"] / f"{mode}_sys_one.prof"])
    launch(mode, world, args, job["work"])
    assert [open(o, "rb").read() for o in out] == want


def test_the_launcher_refuses_what_the_command_line_refuses(job, tmp_path):
    """argument rules of main.cpp:946 (noSubstitutionErrors with errorMutliplier), :862-908 (refBias file / refBiasFile), :351-397 (write / readSysError), :783-786
    (numReads / coverage) -- before any device or process group is touched"""
    args = _pe_args(job, "refused")[0]
    for extra, message in ((["--noSubstitutionErrors", "--errorMutliplier", "2"], "noSubstitutionErrors and errorMutliplier cannot be combined."),
                           (["--refBias", "file"], "refBiasFile option mandatory if refBias is set to 'file'."),
                           (["--refBiasFile", "x.txt"], "refBiasFile option only allowed if refBias is set to 'file'."),          # the common arguments say --refBias no
                           (["--writeSysError", "a", "--readSysError", "b"], "writeSysError and readSysError option are mutually exclusive."),
                           (["--coverage", "3"], "numReads and coverage option are mutually exclusive."),
                           (["--ipfPrecision", "0"], "ipfPrecision must be positive."),
                           (["--backend", "gloo"], "--backend gloo needs --shareDevice"),
                           (["--shareDevice"], "--shareDevice needs --backend gloo")):
        argv = [a for a in args]
        if extra[0] == "--refBias":
            k = argv.index("--refBias")
            del argv[k:k + 2]
        r = subprocess.run([sys.executable, "-m", "reseq_amd.simulate", *map(str, argv), *extra], capture_output=True, text=True, timeout=300, env=_env(tmp_path), cwd=str(ROOT))
        assert r.returncode == 2 and message in r.stderr, (extra, r.stderr[-1500:])
    # the product's entry point knows neither the emulation nor the fault switch
    r = subprocess.run([sys.executable, "-m", "reseq_amd.simulate", *map(str, args), "--emulate"], capture_output=True, text=True, timeout=300, env=_env(tmp_path), cwd=str(ROOT))
    assert r.returncode == 2 and "unrecognized arguments: --emulate" in r.stderr


# --------------------------------------------------------------------------------------------------------------------- (viii)
def _cli_pe(job, tag, extra=(), gz=False, check=True, source=None):
    args, out = _pe_args(source or job, f"cli_{tag}", extra, gz)
    r = subprocess.run([str(RESEQ), "illuminaPE", *map(str, args)], capture_output=True, text=True, timeout=900, env=_env(job["work"]), cwd=str(ROOT))
    if check:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    return r, out


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_reseq_illumina_pe_on_several_workers_in_one_process(job):
    """`reseq illuminaPE --gpus N`: N host threads in the command line's own process, one simulator each on device worker % devices (Simulator.cpp:2830-2836 starts
    its workers itself; main.cpp:436 -j) -- no launcher, no Python, no process group.  With 2, 3 and 8 workers (on a one-device box: all on device 0, which is also
    the test of the header's promise that simulators of one process may be driven from a thread each) the files are the single worker's byte for byte: plain, with
    variants + methylation, as .gz (decompressed), with --writeSysError, with the post-load options; -j asks for workers but never more than there are devices."""
    if _devices() < 1:
        pytest.skip("no device")
    plain = _single_pe("shared", job, "plain")
    meth_extra = ["-V", job["vcf"], "--methylation", job["bed"]]
    meth = _single_pe("shared", job, "variants_methylation", meth_extra)
    for n in (2, 3, 8):
        r, out = _cli_pe(job, f"workers{n}_plain", ["--gpus", n])
        assert f"Simulating with {n} workers on {min(n, _devices())} device(s)" in r.stderr
        assert [open(o, "rb").read() for o in out] == plain, n
        r, out = _cli_pe(job, f"workers{n}_meth", [*meth_extra, "--gpus", n])
        assert [open(o, "rb").read() for o in out] == meth, n
    r, out = _cli_pe(job, "workers3_gz", [*meth_extra, "--gpus", 3], gz=True)
    assert [gzip.decompress(open(o, "rb").read()) for o in out] == meth
    assert all(open(o, "rb").read().endswith(api.gzip_eof_member()) for o in out)
    r, out = _cli_pe(job, "workers1_gz", meth_extra, gz=True)
    assert [gzip.decompress(open(o, "rb").read()) for o in out] == meth and all(open(o, "rb").read().endswith(api.gzip_eof_member()) for o in out)
    r, out = _cli_pe(job, "workers1_host_gz", [*meth_extra, "--hostGzip"], gz=True)
    assert [gzip.decompress(open(o, "rb").read()) for o in out] == meth and not any(open(o, "rb").read().endswith(api.gzip_eof_member()) for o in out)
    r, out = _cli_pe(job, "workers2_host_gz", ["--gpus", 2, "--rsqOption", "host_gzip:1"], gz=True)
    assert [gzip.decompress(open(o, "rb").read()) for o in out] == plain
    # options that change the profile or the pre-pass, on one worker and on four
    bias = job["work"] / "workers_ref_bias.txt"
    names = [n.split(" ")[0] for n, _ in conftest_read_fasta(job["fasta"])]
    bias.write_text(f"{names[0]} 2.0\n{names[1]} 1.0\n{names[2]} 0.25\n")
    for tag, flags in (("edits", ["--errorMutliplier", 2.5, "--noInDelErrors"]), ("bias", ["--refBiasFile", bias])):
        base = list(_pe_args(job, "unused")[0])
        if tag == "bias":
            k = base.index("--refBias")
            del base[k:k + 2]
        outs = {}
        for n in (1, 4):
            args = list(base)
            outs[n] = [str(job["work"] / f"cli_workers{n}_{tag}_{k}.fq") for k in (1, 2)]
            args[args.index("-1") + 1], args[args.index("-2") + 1] = outs[n]
            r = subprocess.run([str(RESEQ), "illuminaPE", *map(str, args), *map(str, flags), "--gpus", str(n)], capture_output=True, text=True, timeout=900, env=_env(job["work"]), cwd=str(ROOT))
            assert r.returncode == 0, r.stderr[-4000:]
        texts = {n: [open(o, "rb").read() for o in outs[n]] for n in outs}
        assert texts[1] == texts[4] and texts[1] != plain, tag
    long_only = job["long_only"]
    r, one = _cli_pe(job, "workers1_sys", ["--writeSysError", job["work"] / "cli_workers1_sys.prof"], source=long_only)
    r, four = _cli_pe(job, "workers4_sys", ["--writeSysError", job["work"] / "cli_workers4_sys.prof", "--gpus", 4], source=long_only)
    assert (job["work"] / "cli_workers1_sys.prof").read_bytes() == (job["work"] / "cli_workers4_sys.prof").read_bytes()
    assert [open(o, "rb").read() for o in one] == [open(o, "rb").read() for o in four]
    # -j: the reference's worker count, at most one worker per device here
    r, out = _cli_pe(job, "threads16_plain", ["-j", 16])
    want_workers = min(16, _devices())
    assert (f"Simulating with {want_workers} workers" in r.stderr) == (want_workers > 1)
    assert [open(o, "rb").read() for o in out] == plain
    # --device D: the one worker's device
    r, out = _cli_pe(job, "device0_plain", ["--device", _devices() - 1])
    assert [open(o, "rb").read() for o in out] == plain
    r, out = _cli_pe(job, "device_none", ["--device", 4096], check=False)
    assert r.returncode != 0 and "device index out of range" in r.stderr
    # refused: --gpus 0, a .bz2 output with several workers; a worker's failure ends the command and leaves no output behind
    r, out = _cli_pe(job, "workers0", ["--gpus", 0], check=False)
    assert r.returncode != 0 and "gpus must be between 1 and 1024." in r.stderr
    args, out = _pe_args(job, "cli_workers_bz2", ["--gpus", 2])
    args[args.index("-1") + 1] += ".bz2"
    r = subprocess.run([str(RESEQ), "illuminaPE", *map(str, args)], capture_output=True, text=True, timeout=900, env=_env(job["work"]), cwd=str(ROOT))
    assert r.returncode != 0 and "bzip2 output is written by one worker only" in r.stderr
    r, out = _cli_pe(job, "workers2_bad_meth", ["--gpus", 2, "--methylation", job["work"] / "no_such_file.bed"], check=False)
    assert r.returncode != 0 and "worker" in r.stderr and not any(os.path.exists(o) for o in out)


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_reseq_seq_to_illumina_on_several_workers_in_one_process(job):
    """`reseq seqToIllumina --gpus N` (Simulator::SimulateErrorModelOnly's worker threads, Simulator.cpp:2900-3014 with -j): the plain input cut into the workers'
    stretches, a record belonging to the worker in whose stretch it begins, every worker's text kept on its device and written at its offset -- the single worker's
    file byte for byte with 2, 3 and 8 workers (records wrapped over lines, so that stretches end inside records), as .gz, with more workers than records; a
    malformed record ends the command with the reference's words and leaves no output; a stream or a compressed input runs with one worker"""
    if _devices() < 1:
        pytest.skip("no device")
    n = 20000
    arrays = synth.make_profile(synth.TINY, seed=5)
    rec = synth.make_error_model_input(9, n, 30, arrays, zero_frac=0.7)
    r = rec["rate"].astype(np.int64)
    rec["rate"] = np.where(r > 86, r - r % 2, r).astype(np.uint8)
    ids = [f"read {i}/x" if i % 7 == 0 else f"r{i}" for i in range(n)]
    fa = job["work"] / "cli_workers_records.fa"
    fa.write_bytes(P.fasta_of_records(rec, ids, wrap_every=3))

    def run(out, *extra, inp=fa, check=True):
        r = subprocess.run([str(RESEQ), "seqToIllumina", "-i", str(inp), "-o", str(out), "-s", job["profile"], "--seed", "13", *map(str, extra)], capture_output=True, text=True,
                           timeout=900, env=_env(job["work"]), cwd=str(ROOT))
        if check:
            assert r.returncode == 0, r.stderr[-4000:]
        return r
    one = job["work"] / "cli_workers_records_one.fq"
    run(one)
    want = one.read_bytes()
    assert want.count(b"\n") == 4 * n and want.startswith(b"@read 0/x ")
    for workers in (2, 3, 8):
        out = job["work"] / f"cli_workers_records_{workers}.fq"
        r = run(out, "--gpus", workers)
        assert f"Simulating with {workers} workers" in r.stderr and f"Generated {n} reads." in r.stderr
        assert out.read_bytes() == want, workers
    gz = job["work"] / "cli_workers_records_3.fq.gz"
    run(gz, "--gpus", 3)
    assert gzip.decompress(gz.read_bytes()) == want and gz.read_bytes().endswith(api.gzip_eof_member())
    run(gz, "--gpus", 2, "--hostGzip")
    assert gzip.decompress(gz.read_bytes()) == want
    # more workers than records; an input without records
    few = job["work"] / "cli_workers_few.fa"
    few.write_bytes(P.fasta_of_records({k: v[:3] for k, v in rec.items()}, ids[:3], wrap_every=3))
    run(job["work"] / "cli_workers_few_1.fq", inp=few)
    run(job["work"] / "cli_workers_few_8.fq", "--gpus", 8, inp=few)
    assert (job["work"] / "cli_workers_few_8.fq").read_bytes() == (job["work"] / "cli_workers_few_1.fq").read_bytes()
    empty = job["work"] / "cli_workers_empty.fa"
    empty.write_bytes(b"")
    r = run(job["work"] / "cli_workers_empty.fq", "--gpus", 2, inp=empty, check=False)
    assert r.returncode != 0 and "does not contain any sequences." in r.stderr and not (job["work"] / "cli_workers_empty.fq").exists()
    # a malformed record in one worker's share
    text = fa.read_bytes()
    cut = text.index(b">", len(text) * 3 // 4)
    end = text.index(b"\n", cut)
    fields = text[cut:end].rsplit(b" ", 1)
    bad = job["work"] / "cli_workers_bad.fa"
    bad.write_bytes(text[:cut] + fields[0] + b" 3" + fields[1][1:] + text[end:])
    r = run(job["work"] / "cli_workers_bad.fq", "--gpus", 4, inp=bad, check=False)
    assert r.returncode != 0 and "Template segment is 3 not 1 or 2" in r.stderr and not (job["work"] / "cli_workers_bad.fq").exists()
    # a compressed input has one reader: one worker, said so
    packed = job["work"] / "cli_workers_records.fa.gz"
    packed.write_bytes(gzip.compress(text))
    r = run(job["work"] / "cli_workers_from_gz.fq", "--gpus", 2, inp=packed)
    assert "one worker runs" in r.stderr and (job["work"] / "cli_workers_from_gz.fq").read_bytes() == want
