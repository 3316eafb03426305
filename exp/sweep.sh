cd $GRAFT_REPO_ROOT
for lib in exp/lib_*.so; do
timeout 300 python bench.py --lib $PWD/$lib --steps 2 --warmup 1 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || tail -2 /tmp/b.err
python -c "
import json;d=json.loads(open('/tmp/b.json').read().strip().split('\n')[-1]);print('$lib', round(d['value']/1e6,2),'Mpairs/s', d['kernel_ms_last_batch'])"
done
