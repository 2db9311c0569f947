#!/bin/bash
cd "$(dirname "$0")/.."
echo "=== device-side exchange tests"
timeout 400 python -m pytest tests/test_gpu_node_team.py -m gpu -q -k "device_side or cannot_allocate" 2>&1 | tail -3
echo "=== Schur tests with the register-resident accumulated factor"
timeout 900 python -m pytest tests/test_gpu_schur.py tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -4
echo "=== bench: chase kernel with U in registers (default) against U in LDS"
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('UREG', d['ms_per_step'], c['hessenberg_s'], c['schur_s'], c['residual_u'], c['schur_sweeps'], c['schur_aeds'])"
STARNEIG_AMD_TUNING=1 SN_SCHUR_CHASE_ULDS=1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('ULDS', d['ms_per_step'], c['hessenberg_s'], c['schur_s'], c['residual_u'], c['schur_sweeps'], c['schur_aeds'])"
echo "=== generalized CTest sweep"
STARNEIG_AMD_TUNING=1 SN_SCHUR_PROFILE=1 timeout 2000 python -m pytest tests/test_gpu_testdriver.py -m gpu -q -s -k "generalized" --durations=25 2>&1 | grep -E "^aed=|\[qz level 0\]|passed|failed|FAILED|Error|assert|s call" | cut -c1-330 | tail -90
echo "=== bench.py --gpus 2 on one GPU over gloo (the N > 1 line: roofline, collectives by kind, cpu_baseline)"
SN_BENCH_ONE_GPU=1 SN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 > gpurun_out/r5_bench_2ranks_gloo.json 2> gpurun_out/r5_bench_2ranks_gloo.err
tail -c 3500 gpurun_out/r5_bench_2ranks_gloo.json; tail -5 gpurun_out/r5_bench_2ranks_gloo.err
