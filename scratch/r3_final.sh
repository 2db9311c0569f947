#!/bin/bash
# End-of-round measurement set (run on the GPU box): the GPU test suite, the bench line, and the kernel
# statistics of the bench command under rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gpu_tests.log
python3 bench.py --steps 2 --warmup 1 > $O/bench_line.json 2> $O/bench_err.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > $O/bench_under_rocprof.log 2>&1
cp $(ls /tmp/p_bench/*/*kernel_stats.csv | head -1) $O/hess_schur_n20000_kernel_stats.csv
python3 $R/scratch/kstats.py /tmp/p_bench 24 > $O/hess_schur_n20000_summary.txt 2>&1
tail -3 $O/gpu_tests.log; cut -c1-400 $O/bench_line.json; head -30 $O/hess_schur_n20000_summary.txt
