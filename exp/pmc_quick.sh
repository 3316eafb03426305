cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/q
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE -d gpurun_out/q/a -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TA_BUSY_avr SQ_WAIT_INST_ANY SQ_WAVE_CYCLES TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/q/b -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/q/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    if "k_fill_reads" in k: print(k, {c: round(sum(v)/len(v)/(3.9e6*3),1) for c,v in sorted(acc[k].items())})
    if "k_sieve" in k or "k_format" in k: print(k, {c: round(sum(v)/len(v)/1e6,2) for c,v in sorted(acc[k].items())})
PY
