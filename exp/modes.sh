cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,2), 'Mpairs/s  fill ms', round(d['roofline']['avg_launch_ms'],2), d['kernel_ms_last_batch'])"; }
run "default"
RSQ_RATE_ROWS=2 run "R2"
RSQ_RATE_ROWS=1 run "R1"
