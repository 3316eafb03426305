#!/bin/bash
# round 5, the evidence of the round's last sources in one lease: bash tools/lease/gpu_r05_evidence.sh <tag>
#   the GPU suite, the default bench command (with its other_configs, cpu_baseline and value_to_host legs), the counter collection of profiles/collect.sh,
#   configs[2] in small calls, the device gzip, configs[4] at full size in calls of the job's default size.  Summaries are made from gpurun_out/<tag> afterwards
#   (python profiles/summarise.py gpurun_out/<tag> <tag>; the JSON lines are copied into profiles/ by name).
tag=${1:-r05_z}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
bash profiles/collect.sh $tag > /tmp/collect.log 2>&1; mv /tmp/collect.log $out/collect.log      # first: it empties gpurun_out/<tag>
timeout 1500 python -m pytest tests -m gpu -x -q -rs > $out/pytest_gpu.log 2>&1; echo "rc $?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
( time timeout 1800 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time; cat $out/bench_default.time | tr '\n' ' '; echo
timeout 600 python tools/bench_error_model.py 8000000 > $out/bench_error_model.json 2> $out/bench_error_model.err
timeout 300 python tools/time_small_calls.py > $out/small_calls.json 2> $out/small_calls.err
timeout 600 python tools/bench_gzip.py 10000000 > $out/gzip_10M_pairs.json 2> $out/gzip.err
timeout 1300 python tools/run_config5.py 1.0 batches 120000,60000 > $out/config5_full.json 2> $out/config5_full.err
python - "$out" <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads([l for l in open(o + "/bench_default.json") if l.startswith("{")][-1])
print("bench", round(d["value"] / 1e6, 2), "M pairs/s", round(d["ms_per_step"], 2), "ms;", d.get("kernel_ms_last_batch"))
for k, v in d.get("other_configs", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("reads_per_s", "pairs_per_s", "batching_invariant")})
print("value_to_host", {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a in ("value", "unit")}) for k, v in d.get("value_to_host", {}).items()} if isinstance(d.get("value_to_host"), dict) else d.get("value_to_host"))
c = json.load(open(o + "/config5_full.json"))
print("configs[4] full", round(c["pairs_per_s_gpu"] / 1e6, 1), c["batching_invariant"], {k: (v["gpu_s"], v["kernel_ms"]) for k, v in c["runs"].items()})
PY
