# round-3 evidence: full collections (kernel statistics + PMC + traffic) of the headline bench and of the 96-tile bench, the default bench line with the CPU
# baseline, configs[2..4] tools
tag=${1:-r03_f}
bash profiles/collect.sh ${tag} > gpurun_out/${tag}_collect.log 2>&1
bash profiles/collect.sh ${tag}_tiles96 --tiles 96 > gpurun_out/${tag}_tiles96_collect.log 2>&1
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python tools/run_config4.py > gpurun_out/${tag}_config4.json 2> gpurun_out/${tag}_config4.err
python tools/run_config5.py 0.1 > gpurun_out/${tag}_config5_tenth.json 2> gpurun_out/${tag}_config5_tenth.err
python tools/bench_error_model.py 50000000 > gpurun_out/${tag}_config3_50M.json 2> gpurun_out/${tag}_config3_50M.err
mkdir -p gpurun_out/profiles_${tag}; cp profiles/${tag}* gpurun_out/profiles_${tag}/
tail -2 gpurun_out/${tag}_collect.log; for f in gpurun_out/${tag}_bench_default.json gpurun_out/${tag}_config4.json gpurun_out/${tag}_config5_tenth.json gpurun_out/${tag}_config3_50M.json; do head -c 600 $f; echo; done
