#!/bin/bash
# the N > 1 code path of bench.py on the one-GPU box: 2 ranks sharing cuda:0 over gloo; then world 1 through RCCL
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time SN_BENCH_ONE_GPU=1 SN_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --steps 1 --warmup 1 --cpu-n 0 --cpu-port-n 0 ) > gpurun_out/r4_bench_2ranks.json 2> gpurun_out/r4_bench_2ranks.err
echo "2 ranks rc=$?"; tail -c 1500 gpurun_out/r4_bench_2ranks.json; tail -5 gpurun_out/r4_bench_2ranks.err
( time timeout 1200 python bench.py --force-sharded --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 ) > gpurun_out/r4_bench_forced.json 2> gpurun_out/r4_bench_forced.err
echo "forced rc=$?"; tail -c 1500 gpurun_out/r4_bench_forced.json; tail -5 gpurun_out/r4_bench_forced.err
