#!/bin/bash
cd /tmp && export TMPDIR=/tmp STARNEIG_AMD_TUNING=1 SN_SCHUR_EARLY_CHASE=1 && rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 $GRAFT_REPO_ROOT/scratch/schur_trace.py 20000 2>&1 | grep "^schur"; python3 $GRAFT_REPO_ROOT/scratch/step_timeline.py /tmp/st
