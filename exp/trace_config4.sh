cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace4
rocprofv3 --kernel-trace --stats -d gpurun_out/trace4 -o t --output-format csv -- python tools/run_config4.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/trace4/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r["Name"].split("(")[0][:40], r["Calls"], round(float(r["TotalDurationNs"])/1e6,1), "ms total", round(float(r["AverageNs"])/1e3,1), "us avg", r["Percentage"])
PY
