#!/bin/bash
# Collects the evidence bench.py's roofline block cites: kernel-trace statistics of the default bench command and the PMC
# passes (each counter group in its own run, no tracing domains combined with --pmc).  Run on the GPU box from the repo
# root: bash profiles/collect.sh <tag>; results land in gpurun_out/<tag>/ and are summarised into profiles/<tag>_*.
tag=${1:-r01}; shift          # further arguments go to every bench.py command (e.g. --tiles 96)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
# --no-host-delivery: the value_to_host leg launches the same kernels on smaller batches, which would mix into the per-kernel averages
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-delivery --no-other-configs $*"
# the un-profiled run first: it compiles the read kernel for the profile (hiprtc) and leaves the code object in the kernel cache.  Compiled for the first time UNDER
# rocprofv3 the same sources gave a kernel with 6.7 % more vector instructions (round 5: 21.73 G against 20.36 G per launch, and every later run of that box took it
# from the cache) -- the profiler must meet the code object the un-profiled runs use.
$B > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- $B > $out/bench_under_rocprof.json 2> $out/stats.err
P="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-delivery --no-other-configs $*"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $out/pmc_sq -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr GRBM_GUI_ACTIVE -d $out/pmc_sq2 -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $out/pmc_l2 -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o p --output-format csv -- $P > /dev/null 2>&1
python profiles/summarise.py $out $tag
