# PC sampling of the headline bench (rocprofv3, beta): where the read kernel's waves are, instruction by instruction.  usage: tools/gpu_pcsample.sh [stochastic|host_trap] [bench args]
method=${1:-stochastic}; shift
out=/root/repo/gpurun_out/pcs_$method
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ "$method" = stochastic ]; then unit=cycles; interval=1048576; else unit=time; interval=200; fi
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval --output-format csv -d $out -o pcs -- \
  python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-delivery "$@" > $out/bench.json 2> $out/bench.err
echo "rc $?"; tail -3 $out/bench.err; ls -la $out $out/* | head -30
f=$(find $out -name "*pc_sampling*csv" | head -1)
if [ -n "$f" ]; then head -5 $f; wc -l $f; python /root/repo/tools/pcsample_summary.py $f > $out/summary.txt 2>&1; head -80 $out/summary.txt; fi
# the raw samples are large: keep the summary and the header
for g in $(find $out -name "*.csv" -size +4M); do head -1000 $g > $g.head; rm $g; done
