#!/bin/bash
cd "$(dirname "$0")/.."
for k in 1 2; do
  python bench.py --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --secondary 0 --host-api 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('single ', d['config']['hessenberg_s'], d['config']['schur_s'])"
  python bench.py --force-sharded --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --secondary 0 --host-api 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('forced ', d['config']['hessenberg_s'], d['config']['schur_s'])"
done
