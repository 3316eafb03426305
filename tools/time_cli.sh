#!/bin/bash
# end-to-end time of the command line on the bench workload (E. coli-sized reference, 10 M pairs), FASTQ written to /dev/shm
set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0, '.')
from reseq_amd import synth
synth.write_profile('/dev/shm/p0.rsqp', synth.make_profile(synth.P0, seed=103741084))
synth.write_fasta('/dev/shm/ecoli.fa', synth.make_reference(2, [4641652], gc=0.508))
PY
for i in 1 2; do
  s=$(date +%s%N); reseq_amd/reseq illuminaPE -R /dev/shm/ecoli.fa -s /dev/shm/p0.rsqp -1 /dev/shm/r1.fq -2 /dev/shm/r2.fq --numReads 10000000 --seed 11 --traceStages 2>&1 | tail -3; echo "wall $(( ($(date +%s%N) - s) / 1000000 )) ms"
done
ls -la /dev/shm/r1.fq /dev/shm/r2.fq; rm -f /dev/shm/r1.fq /dev/shm/r2.fq /dev/shm/ecoli.fa /dev/shm/p0.rsqp
