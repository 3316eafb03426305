#!/bin/bash
# round 5: .gz outputs made on the device, through the command line: 20 M seqToIllumina records to .fq and .fq.gz (and host zlib for comparison on 4 M), illuminaPE 10 M pairs to .fq.gz
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_gz_cli; rm -rf $out; mkdir -p $out
RSQ_S2I_OUT=/dev/shm/rsq_s2i_out.fq.gz timeout 900 python tools/time_seq_to_illumina.py 20000000 > $out/s2i_20M_gz_device.json 2> $out/err1.txt; tail -1 $out/s2i_20M_gz_device.json | cut -c1-1500
rm -f /dev/shm/rsq_s2i_out.fq.gz
timeout 900 python tools/time_seq_to_illumina.py 20000000 > $out/s2i_20M_plain.json 2> $out/err2.txt; tail -1 $out/s2i_20M_plain.json | cut -c1-800
RSQ_S2I_OUT=/dev/shm/rsq_s2i_out_host.fq.gz timeout 900 python tools/time_seq_to_illumina.py 4000000 --rsqOption host_gzip:1 > $out/s2i_4M_gz_host.json 2> $out/err3.txt; tail -1 $out/s2i_4M_gz_host.json | cut -c1-800
rm -f /dev/shm/rsq_s2i_out_host.fq.gz
python tools/bench_gzip.py 10000000 > $out/gzip_10M_pairs.json 2> $out/err4.txt; cat $out/gzip_10M_pairs.json
