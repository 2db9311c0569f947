#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for mode in 0 1 3 0 1 3; do
  echo -n "bench --force-sharded, mode $mode: "
  SN_STREAM_MODE=$mode python bench.py --force-sharded --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --secondary 0 --host-api 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['hessenberg_s'], d['config']['schur_s'])"
done
for mode in 0 1 3; do
  echo -n "bench single, mode $mode: "
  SN_STREAM_MODE=$mode python bench.py --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --secondary 0 --host-api 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['hessenberg_s'], d['config']['schur_s'])"
done
