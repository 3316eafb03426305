#!/bin/bash
# round 4: the suite, and python -m reseq_amd.simulate --gatherOutput under the launcher at world size 1 (dist.gather over RCCL on device tensors) against the command line's files
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04t; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
python - > $out/simulate_gather_nccl_world1.log 2>&1 <<'PY'
import os, subprocess, sys, pathlib, tempfile
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import parity_cases as P
from reseq_amd import synth
work = pathlib.Path(tempfile.mkdtemp())
ppath, fpath, _ = P.make_inputs(work, "nccl1", synth.P0, [300000], prof_seed=103741084)
args = ["-R", fpath, "-s", ppath, "--numReads", "600000", "--seed", "13"]
subprocess.run([os.path.join(root, "reseq_amd", "reseq"), "illuminaPE"] + args + ["-1", str(work / "a1.fq"), "-2", str(work / "a2.fq")], check=True)
env = dict(os.environ, PYTHONPATH=root, NCCL_DEBUG="WARN")
for tag, extra in (("b", ["--gatherOutput", "--gatherSliceMB", "8"]), ("c", [])):
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29521", "-m", "reseq_amd.simulate"] + args +
                       ["-1", str(work / f"{tag}1.fq"), "-2", str(work / f"{tag}2.fq")] + extra, env=env, cwd=root, capture_output=True, text=True)
    print(tag, extra, "RC", r.returncode, r.stderr[-1500:])
    print("FILES_EQUAL_CLI", all((work / f"a{k}.fq").read_bytes() == (work / f"{tag}{k}.fq").read_bytes() for k in (1, 2)), "bytes", (work / "a1.fq").stat().st_size)
PY
tail -3 $out/pytest_gpu.log; grep -E "FILES_EQUAL|RC" $out/simulate_gather_nccl_world1.log
