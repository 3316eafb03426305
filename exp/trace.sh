cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/trace -o t --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
ls -la gpurun_out/trace
python - <<'PY'
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob("gpurun_out/trace/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")))
rows.sort()
# last 2 batches: print timeline of the last ~40 events
last=rows[-45:]
t0=last[0][0]
prev_end=None
for s,e,n in last:
    gap=(s-prev_end)/1e3 if prev_end else 0
    print(f"{(s-t0)/1e3:10.1f} us  dur {(e-s)/1e3:9.1f} us  gap {gap:8.1f} us  {n}")
    prev_end=max(prev_end or e, e)
PY
