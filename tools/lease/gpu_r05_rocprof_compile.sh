#!/bin/bash
# does the run-time compilation give another code object under rocprofv3?  two cache directories, one filled by a profiled run, one by a plain run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_rocprof_compile; rm -rf $out; mkdir -p $out
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-delivery --no-other-configs"
XDG_CACHE_HOME=/tmp/cache_prof rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- $B > $out/prof.json 2> $out/prof.err
XDG_CACHE_HOME=/tmp/cache_plain $B > $out/plain.json 2> $out/plain.err
XDG_CACHE_HOME=/tmp/cache_prof $B > $out/plain_with_prof_cache.json 2> /dev/null
XDG_CACHE_HOME=/tmp/cache_plain rocprofv3 --kernel-trace --stats -d $out/stats2 -o s --output-format csv -- $B > $out/prof_with_plain_cache.json 2> /dev/null
ls -la /tmp/cache_prof/reseq_amd /tmp/cache_plain/reseq_amd; md5sum /tmp/cache_prof/reseq_amd/* /tmp/cache_plain/reseq_amd/*
rocprofv3 --kernel-trace -d $out/envp -o e -- env 2>/dev/null | sort > $out/env_prof.txt; env | sort > $out/env_plain.txt; diff $out/env_plain.txt $out/env_prof.txt | head -40
for f in prof plain plain_with_prof_cache prof_with_plain_cache; do python -c "
import json,sys; d=json.loads([l for l in open('$out/$f.json') if l.startswith('{')][-1]); print('$f', round(d['value']/1e6,2), d['roofline']['avg_launch_ms'], d['config']['fill_plan']['read_kernel_note'])"; done
cp /tmp/cache_prof/reseq_amd/*fill_reads* $out/prof.hsaco 2>/dev/null; cp /tmp/cache_plain/reseq_amd/*fill_reads* $out/plain.hsaco 2>/dev/null
