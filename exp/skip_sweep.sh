cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export RSQ_FILL_MODE=7
for x in 0 1 2 4 8 15; do
export RSQ_LIB=$PWD/exp/skip_$x.so
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM -d gpurun_out/skip/$x -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/skip/$x/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fill_reads" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("skip=$x", {k: round(sum(v)/len(v)/3.9e6,1) for k,v in sorted(acc.items())})
PY
done
