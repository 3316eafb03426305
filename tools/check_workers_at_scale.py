#!/usr/bin/env python3
"""`reseq illuminaPE --gpus N` against one worker at the size of BASELINE's configs[3] / [4]: the files' SHA-256 must be equal.  The tests compare the same on
thousands of bases; here the workers' shares hold hundreds of sequences and a hundred thousand variants, the pre-passes are shared among the workers' threads and the
text of every worker is gigabytes.  One device serves all workers when there is only one (which checks the path, not the speed).

    python tools/check_workers_at_scale.py [drosophila|human] [scale] [workers] [--gz] [--launcher]

--launcher: the N workers are N PROCESSES under torch.distributed.run (python -m reseq_amd.simulate --backend gloo --shareDevice when the box has fewer devices than
ranks, RCCL else) instead of threads of the binary: one load per host through /dev/shm, sharded pre-passes over gloo / RCCL, N writers into one file.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reseq_amd import workloads  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "drosophila"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 5
gz = "--gz" in sys.argv
launcher = "--launcher" in sys.argv
tmp = tempfile.mkdtemp(prefix="rsq_workers_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
ppath = os.path.join(tmp, "p0.rsqp")
extra = []
if which == "drosophila":
    fasta, lengths = workloads.drosophila_sized(tmp, scale)
    workloads.p0_profile(ppath, n_ref_seqs=len(lengths))
else:
    job = workloads.human_sized(tmp, scale)
    fasta, lengths = job["fasta"], job["lengths"]
    workloads.p0_profile(ppath, n_ref_seqs=len(lengths))
    extra = ["-V", job["vcf"], "--methylation", job["bed"]]
exe = os.path.join(ROOT, "reseq_amd", "reseq")
ext = ".fq.gz" if gz else ".fq"


def sha(path):
    h = hashlib.sha256()
    if gz:
        import gzip
        opener = gzip.open
    else:
        opener = open
    with opener(path, "rb") as f:
        for chunk in iter(lambda: f.read(64 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


out = {"workers_are": "processes under torch.distributed.run (reseq_amd.simulate)" if launcher else "threads of the reseq binary (--gpus N)", "config": f"{which}-sized at scale {scale}: {sum(lengths)} bp in {len(lengths)} sequences, P0, coverage 30" + (", variants + methylation" if extra else "") + (", .gz" if gz else ""), "runs": {}}
for n in (1, workers):
    r1, r2 = (os.path.join(tmp, f"w{n}_{k}{ext}") for k in (1, 2))
    t0 = time.perf_counter()
    common = ["-R", fasta, "-s", ppath, "-1", r1, "-2", r2, "--coverage", "30", "--seed", "7", *extra]
    if launcher and n > 1:
        import socket
        from reseq_amd import api
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        share = ["--backend", "gloo", "--shareDevice"] if api.device_count() < n else []
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "reseq_amd.simulate", *common, *share]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    else:
        r = subprocess.run([exe, "illuminaPE", *common, "--gpus", str(n)], capture_output=True, text=True)
    wall = time.perf_counter() - t0
    if r.returncode:
        raise SystemExit(r.stderr[-3000:])
    pairs = [l for l in r.stderr.splitlines() if "Generated" in l][-1]
    out["runs"][f"workers_{n}"] = {"wall_s": round(wall, 2), "bytes": [os.path.getsize(r1), os.path.getsize(r2)], "sha256": [sha(r1), sha(r2)], "said": pairs.strip()}
    os.remove(r1), os.remove(r2)
a, b = out["runs"]["workers_1"], out["runs"][f"workers_{workers}"]
out["equal"] = a["sha256"] == b["sha256"] and (gz or a["bytes"] == b["bytes"])
print(json.dumps(out))
import shutil
shutil.rmtree(tmp, ignore_errors=True)
sys.exit(0 if out["equal"] else 1)
