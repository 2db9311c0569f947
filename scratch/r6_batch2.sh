#!/bin/bash
# round 6, second batch: the whole GPU suite on the fence-free ordered sums, the two-stage Hessenberg-triangular path at
# the sizes of round 5's table, its kernel statistics at n = 8000, its PMC traffic, the queue-order reproducer
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r6_gpu_tests_a.log 2>&1
tail -16 gpurun_out/r6_gpu_tests_a.log
python scratch/r5_ht2.py 1500 2500 4000 6000 8000 12000 8000 12000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_ht_twostage_sizes.txt
cd /tmp; rm -rf /tmp/p_ht
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ht -- python3 $GRAFT_REPO_ROOT/scratch/r5_ht2.py 8000 > /tmp/ht.log 2>&1
cd "$GRAFT_REPO_ROOT"
python3 scratch/kstats.py /tmp/p_ht 14 | tee gpurun_out/r6_ht_twostage_kernel_stats_n8000.txt
bash scratch/r6_ht2_pmc.sh 4000 > gpurun_out/r6_ht2_pmc_stdout.log 2>&1; tail -5 gpurun_out/r6_ht2_pmc_stdout.log
cd scratch/micro
for cfg in "4 0 20000 65536 1" "4 8 20000 65536 1" "8 8 20000 65536 1" "4 16 20000 16384 1" "8 16 40000 4096 0"; do timeout 300 ./queue_order $cfg; done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r6_queue_order.txt
