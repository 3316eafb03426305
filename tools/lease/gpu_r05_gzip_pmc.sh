#!/bin/bash
# the gzip kernel under the counters: bash tools/lease/gpu_r05_gzip_pmc.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_gzip_pmc; rm -rf $out; mkdir -p $out
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $out/$tag -o p --output-format csv -- python tools/bench_gzip.py 4000000 > /dev/null 2> $out/err_$tag.txt
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gzip" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, "launches", len(next(iter(cs.values()))), {n: round(v) for n, v in sorted(c.items())})
    if cyc:
        print("   ms", cyc / 2.4e6, "valu busy", c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc), "lds busy", c.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), "conflict share", c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, c.get("SQ_LDS_IDX_ACTIVE", 1)),
              "wait share", c.get("SQ_WAIT_INST_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)), "lds wait share", c.get("SQ_WAIT_INST_LDS", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)))
PY
