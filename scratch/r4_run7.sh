#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -X faulthandler -m pytest tests/test_gpu_node_team.py -m gpu -q -x > gpurun_out/r4_node_team.log 2>&1
head -c 3000 gpurun_out/r4_node_team.log; tail -5 gpurun_out/r4_node_team.log
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r4_gpu_tests.log 2>&1
tail -25 gpurun_out/r4_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r4_bench_line.json 2> gpurun_out/r4_bench_err.log
tail -c 3000 gpurun_out/r4_bench_line.json
