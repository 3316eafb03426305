#!/bin/bash
# the variant job's read kernel under the counters, with and without the plain / variant split: bash tools/lease/gpu_r05_var_pmc.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_var_pmc; rm -rf $out; mkdir -p $out
for split in 1 0; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $out/split$split -o p --output-format csv -- python tools/run_config5.py 0.1 --option split_plain=$split > $out/split$split.json 2> $out/split$split.err
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for split in (1, 0):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/split{split}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "fill_re" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r.get("VGPR_Count"), r.get("Scratch_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print("split", split, k, {n: (len(v), round(sum(v) / len(v))) for n, v in sorted(cs.items())})
PY
