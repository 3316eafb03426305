#!/bin/bash
# a bench line and the read kernel's VALU / LDS counters of the current build: bash tools/gpu_quick.sh <tag> [bench.py flags]
tag=${1:-quick}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery --no-other-configs $*"
timeout 600 $B > $out/bench.json 2> $out/bench.err
P="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-delivery --no-other-configs $*"
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $out/pmc -o p --output-format csv -- $P > /dev/null 2> $out/pmc.err
python - "$out" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fill_re" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    res[k] = dict(c, kernel_ms_at_2_4GHz=cyc / 2.4e6, valu_busy=c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc) if cyc else None,
                  lds_array_busy=c.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc) if cyc else None, lds_conflict_share=c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                  lds_cycles_per_instruction=c.get("SQ_LDS_IDX_ACTIVE", 0) / max(c.get("SQ_INSTS_LDS", 1), 1))
json.dump(res, open(out + "/counters.json", "w"), indent=1)
try:
    d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
    print("BENCH", round(d["value"] / 1e6, 2), "M pairs/s", round(d["ms_per_step"], 2), "ms/step; read kernel", round(d["roofline"]["avg_launch_ms"], 2), "ms;", d["config"]["fill_plan"])
except Exception as e:
    print("no bench line", e, open(out + "/bench.err").read()[-2000:])
print(json.dumps(res, indent=1))
PY
