#!/bin/bash
# the -m gpu suite + the default bench line: bash tools/gpu_tests.sh <tag>
tag=${1:-tests}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
tail -3 $out/pytest_gpu.log
