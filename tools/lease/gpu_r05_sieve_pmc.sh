#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_sieve_pmc; rm -rf $out; mkdir -p $out
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $out/$tag -o p --output-format csv -- python tools/run_config5.py 0.1 batches 320000 > /dev/null 2> $out/err_$tag.txt
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sieve_finish" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:40], r["Grid_Size"], r["VGPR_Count"] if "VGPR_Count" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, cs in sorted(acc.items()):
    print(g, {k: round(sum(v) / len(v)) for k, v in sorted(cs.items())})
PY
