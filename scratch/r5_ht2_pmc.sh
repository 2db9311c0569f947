#!/bin/bash
# HBM traffic of the two reflector-application kernels of stage 2 of the two-stage Hessenberg-triangular path
# (n = 4000, the whole reduction): rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE, in separate passes, against the
# algorithmic bytes.  Run on the GPU box; result in gpurun_out/r5_ht2_pmc.json.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmh; mkdir -p /tmp/pmh
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmh/fetch -- python3 $R/scratch/r5_ht2.py 4000 > gpurun_out/r5_ht2_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmh/write -- python3 $R/scratch/r5_ht2.py 4000 > gpurun_out/r5_ht2_pmc_write.log 2>&1
python3 - <<'PY' > gpurun_out/r5_ht2_pmc.json
import csv, glob, json, collections
import numpy as np
def load(d, counter):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                for key in ("ht2_apply_left_kernel", "ht2_apply_right_kernel", "ht2_gen_kernel"):
                    if key in r["Kernel_Name"]:
                        tot[key][0] += float(r["Counter_Value"]); tot[key][1] += 1
    return tot
f, w = load("/tmp/pmh/fetch", "FETCH_SIZE"), load("/tmp/pmh/write", "WRITE_SIZE")
n, r = 4000, 64
j = np.arange(n - 2, dtype=np.int64)[:, None]; t = np.arange((n - 3) // r + 1, dtype=np.int64)[None, :]
p = j + 1 + r * t; live = p <= n - 2; p1 = np.minimum(p + r, n); ln = p1 - p; c0 = np.where(t == 0, j, p - r)
left = 8.0 * float((ln * ((n - c0 - 1) + (n - p)) * live).sum()); right = 8.0 * float((ln * (np.minimum(p1 + r, n) + p1) * live).sum())
out = {"command": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --kernel-trace -- python3 scratch/r5_ht2.py 4000",
       "note": "FETCH_SIZE counts 64 B per 128-B request on gfx950 (x 2), both counters in KiB; algorithmic bytes: every entry a step touches read once and written once",
       "kernels": {}}
for key, alg in (("ht2_apply_left_kernel", left), ("ht2_apply_right_kernel", right)):
    fb, wb = 2.0 * f[key][0] * 1024.0, w[key][0] * 1024.0
    out["kernels"][key] = {"dispatches": f[key][1], "fetched_bytes": fb, "written_bytes": wb, "algorithmic_read_bytes": alg, "algorithmic_written_bytes": alg,
                           "fetch_ratio": fb / alg, "write_ratio": wb / alg, "traffic_ratio": (fb + wb) / (2 * alg)}
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r5_ht2_pmc.json; tail -2 gpurun_out/r5_ht2_pmc_fetch.log | cut -c1-200
