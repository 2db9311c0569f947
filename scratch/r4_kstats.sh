#!/bin/bash
# kernel statistics of the bench command at HEAD (two timed steps + one warm-up = three reductions)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_profiles; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > $O/bench_under_rocprof.log 2>&1
cp $(ls /tmp/p_bench/*/*kernel_stats.csv | head -1) $O/hess_schur_n20000_kernel_stats.csv
python3 scratch/kstats.py /tmp/p_bench 25 > $O/hess_schur_n20000_summary.txt 2>&1
tail -1 $O/bench_under_rocprof.log >> $O/hess_schur_n20000_summary.txt
python3 scratch/step_timeline.py /tmp/p_bench > $O/schur_step_anatomy.txt 2>&1
head -30 $O/hess_schur_n20000_summary.txt | cut -c1-160
cat $O/schur_step_anatomy.txt
