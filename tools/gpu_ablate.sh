# read-kernel time with parts of the per-position step cut out (exp/ablate_read_step.patch builds, exp/build/libabl_<n>.so): wrong output, only the time is of interest
mkdir -p gpurun_out/abl
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery"
for n in 0 "$@"; do
  if [ $n = 0 ]; then lib=""; else lib="--lib exp/build/libabl_$n.so"; fi
  $B $lib > gpurun_out/abl/b$n.json 2> gpurun_out/abl/b$n.err || tail -2 gpurun_out/abl/b$n.err
  python -c "
import json
d=json.load(open('gpurun_out/abl/b$n.json'))
print('ablate $n', round(d['ms_per_step'],1), {k: round(v,2) for k,v in d['kernel_ms_last_batch'].items()})"
done
