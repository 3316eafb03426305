#!/bin/bash
# round 4, first call: the suite as it stands, the bench line, and the RCCL calls of the N-rank path executed on the one leased GPU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04a; rm -rf $out; mkdir -p $out
( nproc; cat /sys/fs/cgroup/cpu.max; df -h / /tmp /dev/shm 2>&1; mount | grep -v -e proc -e sysfs -e cgroup | head -30; free -g; rocm-smi --showmeminfo vram 2>&1 | head; ls /opt/rocm/lib | grep hiprtc ) > $out/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
# one rank with a process group: RCCL init, barrier, all_reduce, all_gather on device tensors
NCCL_DEBUG=INFO timeout 600 python bench.py --dist-single --steps 3 --warmup 1 > $out/bench_dist_single_nccl.json 2> $out/bench_dist_single_nccl.err
# the same through the launcher the driver uses
NCCL_DEBUG=WARN timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_torchrun1.json 2> $out/bench_torchrun1.err
# bare --gpus 2 on a box with one GPU: two ranks are started, the second has no device -- the log shows that the launch happened
timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > $out/bench_gpus2_on_one_gpu.out 2>&1; echo "rc $?" >> $out/bench_gpus2_on_one_gpu.out
# the launcher module under torchrun with one rank (broadcast of the seed, sharded pre-pass with its all-reduce / all-gathers, sizes, agreement flags) against the CLI
python - > $out/simulate_nccl_world1.log 2>&1 <<'PY'
import os, subprocess, sys, pathlib, tempfile
root = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import parity_cases as P
from reseq_amd import synth
work = pathlib.Path(tempfile.mkdtemp())
ppath, fpath, _ = P.make_inputs(work, "nccl1", synth.TINY, [5000, 80, 3210])
args = ["-R", fpath, "-s", ppath, "--numReads", "30000", "--seed", "13", "--refBias", "no"]
subprocess.run([os.path.join(root, "reseq_amd", "reseq"), "illuminaPE"] + args + ["-1", str(work / "a1.fq"), "-2", str(work / "a2.fq")], check=True)
env = dict(os.environ, PYTHONPATH=root, NCCL_DEBUG="INFO")
r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29519", "-m", "reseq_amd.simulate"] + args +
                   ["-1", str(work / "b1.fq"), "-2", str(work / "b2.fq"), "--batchBlocks", "3"], env=env, cwd=root, capture_output=True, text=True)
print(r.stdout[-6000:]); print(r.stderr[-12000:])
same = all((work / f"a{k}.fq").read_bytes() == (work / f"b{k}.fq").read_bytes() for k in (1, 2))
print("RC", r.returncode, "FILES_EQUAL_CLI", same, "bytes", (work / "a1.fq").stat().st_size)
PY
