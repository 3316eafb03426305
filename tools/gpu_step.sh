# GPU check of the working tree: the -m gpu suite, then the headline bench (no CPU baseline), the kernel times of its last batch
mkdir -p gpurun_out/step
if [ -z "$NO_TESTS" ]; then python -m pytest tests -m gpu -x -q > gpurun_out/step/tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/step/tests.log; fi
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery"
run() { tag=$1; shift; $B "$@" > gpurun_out/step/bench_$tag.json 2> gpurun_out/step/bench_$tag.err || tail -3 gpurun_out/step/bench_$tag.err
python -c "
import json
d=json.load(open('gpurun_out/step/bench_$tag.json'))
print('$tag', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k: round(v,2) for k,v in d['kernel_ms_last_batch'].items()}, d['config'].get('read_kernel_launches_per_step'))
"; }
for spec in "$@"; do run ${spec%%:*} ${spec#*:}; done
