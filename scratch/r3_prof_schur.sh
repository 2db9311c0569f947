#!/bin/bash
# kernel-time profile of the Schur leg with and without the aggregated lazy updates
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export STARNEIG_AMD_TUNING=1
for a in 0 1; do
  export SN_SCHUR_AGG=$a
  rm -rf /tmp/prof_$a
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$a -- python3 $R/scratch/schur_only.py 20000 > $R/gpurun_out/r3_prof_schur_$a.log 2>&1
  echo "== AGG=$a" >> $R/gpurun_out/r3_prof_schur.txt
  tail -1 $R/gpurun_out/r3_prof_schur_$a.log >> $R/gpurun_out/r3_prof_schur.txt
  python3 $R/scratch/kstats.py /tmp/prof_$a 16 >> $R/gpurun_out/r3_prof_schur.txt 2>&1
done
cat $R/gpurun_out/r3_prof_schur.txt
