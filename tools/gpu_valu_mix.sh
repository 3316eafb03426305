cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/valu_mix; rm -rf $out; mkdir -p $out
P="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-delivery"
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INSTS_SALU\|SQ_INSTS_SMEM\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_INSTS_[A-Z0-9_]*" | sort -u > $out/counters.txt
cat $out/counters.txt | tr '\n' ' '
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $out/p1 -o p --output-format csv -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $out/p2 -o p --output-format csv -- $P > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/valu_mix/p*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fill_reads" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
steps = 47.5e6
for k, v in sorted(acc.items()):
    print(f"{k:28} {sum(v)/len(v):14.4g}  per wave-step {sum(v)/len(v)/steps:8.1f}")
PY
