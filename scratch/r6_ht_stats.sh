#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for N in 8000 12000; do
  rm -rf /tmp/pq; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -- python3 $R/scratch/r5_ht2.py $N > /tmp/pq.log 2>&1
  grep "n=" /tmp/pq.log | tail -1 > $R/gpurun_out/r6_ht_stats_$N.txt
  python3 $R/scratch/kstats.py /tmp/pq 16 >> $R/gpurun_out/r6_ht_stats_$N.txt 2>&1
done
cat $R/gpurun_out/r6_ht_stats_8000.txt | cut -c1-150
