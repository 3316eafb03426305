mkdir -p gpurun_out/r03c
python -m pytest tests/test_parity_gpu.py -x -q -k "tiles or error_model or p0_reads or tiny" > gpurun_out/r03c/tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r03c/tests.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery"
run() { tag=$1; shift; $B "$@" > gpurun_out/r03c/bench_$tag.json 2> gpurun_out/r03c/bench_$tag.err
python -c "
import json
d=json.load(open('gpurun_out/r03c/bench_$tag.json'))
print('$tag', round(d['value']/1e6,1), round(d['ms_per_step'],1), {k: round(v,2) for k,v in d['kernel_ms_last_batch'].items()})
"; }
run t1
run t1_binned --option image_tiles=1

run t3 --tiles 3
run t96 --tiles 96

