# randomised parity runs on the GPU after the round's changes to the chains, the variants' chain states and the readers; logs kept under gpurun_out/stress
# usage: tools/gpu_stress_r03.sh <trials tiny> <trials p0>; every tool runs with the library's defaults and again with short chain chunks and an odd run-up
mkdir -p gpurun_out/stress
run() { tag=$1; shift; (timeout 1500 "$@"; echo "exit $?") > gpurun_out/stress/$tag.log 2>&1; echo "== $tag"; tail -2 gpurun_out/stress/$tag.log; }
W="python tools/with_options.py chain_chunk=64,chain_warmup=23"
run plain_tiny python tools/stress_plain.py ${1:-24} gpu tiny
run plain_p0 python tools/stress_plain.py ${2:-6} gpu p0
run variants_tiny python tools/stress_variants.py ${1:-24} gpu tiny
run variants_p0 python tools/stress_variants.py ${2:-6} gpu p0
run sharded_prepare python tools/stress_sharded_prepare.py ${1:-24}
run plain_tiny_chunk64 $W tools/stress_plain.py ${1:-24} gpu tiny
run variants_tiny_chunk64 $W tools/stress_variants.py ${1:-24} gpu tiny
run variants_p0_chunk64 $W tools/stress_variants.py ${2:-6} gpu p0
run sharded_prepare_chunk64 $W tools/stress_sharded_prepare.py ${1:-24}
