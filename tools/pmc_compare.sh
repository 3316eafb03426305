#!/bin/bash
# Counters of the read kernel for several bench.py variants side by side (each counter group in its own run; no tracing domains with --pmc).
#   bash tools/pmc_compare.sh <outdir> "<name>:<bench args>" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1; shift; mkdir -p $out
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  P="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-delivery $args"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $out/$name/pmc_sq -o p --output-format csv -- $P > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr GRBM_GUI_ACTIVE -d $out/$name/pmc_sq2 -o p --output-format csv -- $P > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_WAIT_INST_LDS TCP_TOTAL_CACHE_ACCESSES_sum TCC_MISS_sum -d $out/$name/pmc_sq3 -o p --output-format csv -- $P > /dev/null 2>&1
done
python - "$out" "$@" <<'PY'
import collections, csv, glob, json, sys
out, specs = sys.argv[1], sys.argv[2:]
table = collections.OrderedDict()
for spec in specs:
    name = spec.split(":", 1)[0]
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(f"{out}/{name}/pmc_*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "k_fill_reads" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    table[name] = {c: sum(v) / len(v) for c, v in acc.items()}
json.dump(table, open(f"{out}/pmc_compare.json", "w"), indent=1)
names = list(table)
print(f"{'counter':32}" + "".join(f"{n:>18}" for n in names))
for c in sorted({c for t in table.values() for c in t}):
    print(f"{c:32}" + "".join(f"{table[n].get(c, float('nan')):18.4g}" for n in names))
PY
