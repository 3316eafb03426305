#!/bin/bash
# round 4, call b: the family geometry + the read kernel compiled for the profile: parity on both routes, bench on both routes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04b; rm -rf $out; mkdir -p $out
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery"
timeout 600 $B > $out/bench_spec.json 2> $out/bench_spec.err
timeout 600 $B --option specialize=0 > $out/bench_generic.json 2> $out/bench_generic.err
timeout 600 $B --tiles 96 > $out/bench_spec_tiles96.json 2> $out/bench_spec_tiles96.err
timeout 600 $B --tiles 96 --option specialize=0 > $out/bench_generic_tiles96.json 2> $out/bench_generic_tiles96.err
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
