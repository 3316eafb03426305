#!/bin/bash
# round 4: N writers behind one GPU into one pair of files, buffered / direct, against a pair per process; tmpfs and the box's disk-backed file system
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04w; rm -rf $out; mkdir -p $out
mkdir -p /var/tmp/rsq_w
( df -T / /tmp /var/tmp /dev/shm; cat /proc/mounts | grep -E " / | /tmp | /var" ) > $out/file_systems.txt 2>&1
for n in 1 2 4 8; do
  timeout 900 python tools/measure_rank_writes.py $n /var/tmp/rsq_w 5000000 > $out/disk_$n.json 2> $out/disk_$n.err
  timeout 900 python tools/measure_rank_writes.py $n /dev/shm 5000000 > $out/shm_$n.json 2> $out/shm_$n.err
done
