#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run6.log; : > $L
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests.log 2>&1; tail -5 gpurun_out/r3_gpu_tests.log >> $L
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/r3_bench_line.json 2>gpurun_out/r3_bench.err; tail -c 3000 gpurun_out/r3_bench_line.json >> $L
timeout 600 python bench.py --steps 2 --warmup 1 --force-sharded --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > gpurun_out/r3_bench_sharded.json 2>>gpurun_out/r3_bench.err; tail -c 1500 gpurun_out/r3_bench_sharded.json >> $L
tail -5 gpurun_out/r3_bench.err >> $L
cat $L
