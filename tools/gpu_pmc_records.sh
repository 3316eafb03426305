cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_rec; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $out/a -o p --output-format csv -- python tools/bench_error_model.py 8000000 > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_avr SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $out/b -o p --output-format csv -- python tools/bench_error_model.py 8000000 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_rec/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fill_rec" in r["Kernel_Name"] and int(r["Grid_Size"]) > 200000:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    c = {n: max(v) for n, v in cs.items()}       # the large (8 M record) launches
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(k, {n: round(v / 1e6, 1) for n, v in c.items()})
    print(" ms", cyc / 2.4e6, "valu busy", c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), "lds busy", c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), "ta busy", c.get("TA_BUSY_avr", 0) / cyc, "vmem rd per wave-step", c["SQ_INSTS_VMEM_RD"] / (8e6 * 150 / 64))
PY
