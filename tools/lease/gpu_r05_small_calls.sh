#!/bin/bash
# round 5: what a small call of the records kernel spends its time on: kernel trace and cycles of tools/bench_error_model.py (8 M records, also in blocks of 48 MB)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_small; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python tools/bench_error_model.py 8000000 > $out/bench.json 2> $out/err.txt
python - "$out" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
f = glob.glob(out + "/stats/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"].split("(")[0][:50], r["Grid_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{k:52s} grid {g:>9s} n {len(v):4d} avg {sum(v)/len(v)/1e3:9.1f} us  min {min(v)/1e3:9.1f}  total {sum(v)/1e6:8.2f} ms")
d = json.loads([l for l in open(out + "/bench.json") if l.startswith("{")][-1])
print({k: {a: b for a, b in v.items() if a != "seconds"} for k, v in d["from_fasta_text_parsed_on_device"].items() if isinstance(v, dict)})
PY
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $out/pmc -o p --output-format csv -- python tools/bench_error_model.py 8000000 > /dev/null 2> $out/err2.txt
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fill_rec" in r["Kernel_Name"]:
            acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, cs in acc.items():
    print("grid", g, {k: (len(v), round(sum(v) / len(v))) for k, v in cs.items()}, "cycles per launch (GRBM / 8 XCD)", round(sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"]) / 8))
PY
