#!/bin/bash
# per-kernel averages of the Hessenberg leg under rocprofv3 (n = 20000), plus the leg's time without the profiler
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python scratch/hess_only.py 20000 3 2>&1 | grep -v amdgpu.ids | tail -3
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_h -- python3 $GRAFT_REPO_ROOT/scratch/hess_only.py 20000 > /tmp/h.log 2>&1
cd "$GRAFT_REPO_ROOT"
python3 scratch/kstats.py /tmp/p_h 16 | tee gpurun_out/r6_col_${1:-a}.txt
