#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04l; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
timeout 1500 python tools/run_config5.py 1.0 loads 8 > $out/c5_loads8.json 2> $out/c5_loads8.err
tail -3 $out/pytest_gpu.log
