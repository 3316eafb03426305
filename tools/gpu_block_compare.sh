#!/bin/bash
# the configurations that run the read kernels, for comparing builds (e.g. RSQ_FILL_BLOCK=1024 against 768): bash tools/gpu_block_compare.sh <tag>
tag=${1:-blk}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json,re
m=re.findall(r'\{\"metric\".*\}', sys.stdin.read())
d=json.loads(m[-1]); print('$1', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms/step; read kernel', round(d['roofline']['avg_launch_ms'],2), 'ms')"; }
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery 2>$out/b1.err | tee $out/bench.json | line default
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery --tiles 96 2>$out/b96.err | tee $out/bench_tiles96.json | line tiles96
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery --option specialize=0 2>$out/bl.err | tee $out/bench_library.json | line library_instantiation
python tools/bench_error_model.py 8000000 > $out/config3.json 2>$out/c3.err; python -c "
import json; d=json.load(open('$out/config3.json')); print('config3 arrays', round(d['reads_per_s']/1e6,1), 'text', round(d['with_fastq_text_on_device']['reads_per_s']/1e6,1), 'fasta one call', round(d['from_fasta_text_parsed_on_device']['one_call']['reads_per_s']/1e6,1), d['from_fasta_text_parsed_on_device']['one_call']['kernel_ms_summed'])"
python tools/run_config5.py 0.1 > $out/config5_tenth.json 2>$out/c5.err; python -c "
import json; d=json.load(open('$out/config5_tenth.json')); print('config5 1/10', {k: d[k] for k in d if 'pairs_per_s' in k or k in ('kernel_ms', 'gpu_seconds')})" 2>/dev/null || tail -c 600 $out/config5_tenth.json
python tools/run_config4.py > $out/config4.json 2>$out/c4.err; tail -c 500 $out/config4.json
