#!/bin/bash
# `reseq seqToIllumina` on 20 M records in /dev/shm under a few settings: bash tools/gpu_s2i.sh <tag>
tag=${1:-s2i}; out=gpurun_out/$tag; mkdir -p $out
run() { name=$1; shift; python tools/time_seq_to_illumina.py 20000000 "$@" > $out/s2i_$name.json 2> $out/s2i_$name.err; python - $out/s2i_$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], [round(t, 2) for t in d["wall_s"]], d["stages"])
PY
}
run default
run readers2 --readThreads 2
run readers12 --readThreads 12
run batch1 --batchBlocks 1
run block16 --blockKB 16384
RSQ_S2I_OUT=/dev/null run devnull
lscpu | grep -i "numa\|socket\|model name" > $out/lscpu.txt
RSQ_S2I_OUT=/tmp/rsq_s2i_out.fq run overlay
rm -f /tmp/rsq_s2i_out.fq
python tools/bench_error_model.py 8000000 > $out/config3_8M.json 2> $out/config3_8M.err; python -c "
import json,sys; d=json.load(open('$out/config3_8M.json')); print(json.dumps(d['from_fasta_text_parsed_on_device'], indent=1)); print(d['reads_per_s'], d['with_fastq_text_on_device']['reads_per_s'])"; tail -3 $out/config3_8M.err
