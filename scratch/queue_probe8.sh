#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for mode in 0 1 0 1; do
  echo "secondary, mode $mode: "
  SN_STREAM_MODE=$mode python bench.py --workload secondary 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('[') or l.startswith('{')][-1])
d = d if isinstance(d, list) else d.get('secondary', d)
for e in d: print('   ', e.get('pencil','')[:12], e.get('n'), e.get('hessenberg_triangular_s'), e.get('qz_s'), e.get('seconds_per_step'))
"
done
