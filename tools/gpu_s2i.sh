# wall time of reseq seqToIllumina on N records with several parser-thread counts (tools/time_seq_to_illumina.py builds the input once per call)
N=${1:-20000000}
for t in 6 16 32; do
python - "$N" "$t" <<'PY'
import sys, re, subprocess, json
n, t = sys.argv[1], sys.argv[2]
src = open("tools/time_seq_to_illumina.py").read().replace('"--seed", "5"]', '"--seed", "5", "--parseThreads", "%s"]' % t)
open("/tmp/ts2i.py", "w").write(src.replace('os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', 'os.getcwd()'))
r = subprocess.run([sys.executable, "/tmp/ts2i.py", n], capture_output=True, text=True)
line = r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-500:]
try:
    d = json.loads(line); print("threads", t, "wall", d["wall_s"], "M reads/s", round(d["reads_per_s_wall"]/1e6, 2), d["first_records_equal_oracle"], d["records_in_the_middle_equal_oracle"])
except Exception: print(line)
PY
done
