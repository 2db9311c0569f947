#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hessenberg.py -m gpu -q -x 2>&1 | tail -2
python scratch/hess_only.py 20000 3 2>&1 | grep -v amdgpu.ids | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_h -- python3 $GRAFT_REPO_ROOT/scratch/hess_only.py 20000 > /tmp/h.log 2>&1
python3 scratch/kstats.py /tmp/p_h 8
