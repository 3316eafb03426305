#!/bin/bash
# round 5, first lease after the multi-GPU tests and the other_configs leg: the new -m gpu tests, then the whole default bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_b; rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests/test_multi_gpu.py tests/test_parity_gpu.py -m gpu -x -q -rs -k "multi_gpu or packed_reference or one_rank or fasta" > $out/pytest_new.log 2>&1; echo "rc $?" >> $out/pytest_new.log
tail -15 $out/pytest_new.log
( time timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err ) 2> $out/bench.time
tail -3 $out/bench.err; cat $out/bench.time
python - "$out" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1] + "/bench.json") if l.startswith("{")][-1])
print("value", d["value"] / 1e6, "ms", d["ms_per_step"])
for k, v in d.get("other_configs", {}).items():
    print(k, {a: b for a, b in v.items() if a not in ("runs", "workload")})
PY
