#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export STARNEIG_AMD_TUNING=1 SN_SCHUR_HS_PRIO=1 SN_SCHUR_AGG=${1:-0}
rm -rf /tmp/trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python3 $R/scratch/schur_only.py 20000 > $R/gpurun_out/r3_trace.log 2>&1
tail -1 $R/gpurun_out/r3_trace.log
python3 $R/scratch/chase_timeline.py /tmp/trace
