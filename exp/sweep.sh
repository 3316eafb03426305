cd $GRAFT_REPO_ROOT
for lib in exp/lib_*.so; do for m in 7 15; do
RSQ_LIB=$PWD/$lib RSQ_FILL_MODE=$m timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || tail -2 /tmp/b.err
python -c "
import json;d=json.load(open('/tmp/b.json'));print('$lib', $m, round(d['value']/1e6,2),'Mpairs/s fill avg ms', round(d['roofline']['avg_launch_ms'],2))"
done; done
