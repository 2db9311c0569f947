#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python bench.py --force-sharded --steps 1 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > gpurun_out/r5_bench_forced.json 2> gpurun_out/r5_bench_forced.err; tail -c 1500 gpurun_out/r5_bench_forced.json; tail -3 gpurun_out/r5_bench_forced.err
timeout 600 python bench.py --workload ht --steps 1 --warmup 0 > gpurun_out/r5_bench_ht_line.json 2> gpurun_out/r5_bench_ht.err; tail -c 1200 gpurun_out/r5_bench_ht_line.json
STARNEIG_AMD_TUNING=1 SN_HT_TWOSTAGE=1 timeout 600 python bench.py --workload ht --steps 1 --warmup 0 --cpu-ht-n 0 > gpurun_out/r5_bench_ht_twostage_line.json 2> gpurun_out/r5_bench_ht2.err; tail -c 1200 gpurun_out/r5_bench_ht_twostage_line.json
timeout 600 python bench.py --n 40000 --steps 1 --warmup 0 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > gpurun_out/r5_bench_n40000.json 2> gpurun_out/r5_bench_n40000.err; tail -c 1500 gpurun_out/r5_bench_n40000.json
