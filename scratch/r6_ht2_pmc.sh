#!/bin/bash
# HBM traffic of stage 2 of the two-stage Hessenberg-triangular path (n = 4000, the whole reduction): rocprofv3 --pmc
# FETCH_SIZE, then WRITE_SIZE, in separate passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half the bytes of a
# wide stream on gfx950), per kernel, against the algorithmic bytes of bench.py's two_stage_roofline.
# Result: gpurun_out/r6_ht2_pmc.json
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-4000}
rm -rf /tmp/pmh; mkdir -p /tmp/pmh
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmh/fetch -- python3 $R/scratch/r5_ht2.py $N > $R/gpurun_out/r6_ht2_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmh/write -- python3 $R/scratch/r5_ht2.py $N > $R/gpurun_out/r6_ht2_pmc_write.log 2>&1
cd $R
python3 - $N <<'PY' > gpurun_out/r6_ht2_pmc.json
import csv, glob, json, collections, sys
import numpy as np
n = int(sys.argv[1])
KEYS = ("ht2_geng_left_kernel", "ht2_apply_right_kernel", "ht2_wy_right_kernel", "ht2_genh_kernel", "ht2_group_wy_kernel")
def load(d, counter):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                for key in KEYS:
                    if key in r["Kernel_Name"]:
                        tot[key][0] += float(r["Counter_Value"]); tot[key][1] += 1
    return tot
f, w = load("/tmp/pmh/fetch", "FETCH_SIZE"), load("/tmp/pmh/write", "WRITE_SIZE")
r = gs = 64
j = np.arange(n - 2, dtype=np.int64)[:, None]; t = np.arange((n - 3) // r + 1, dtype=np.int64)[None, :]
p = j + 1 + r * t; live = p <= n - 2; p1 = np.minimum(p + r, n); ln = p1 - p; c0 = np.where(t == 0, j, p - r)
top = (j // gs) * gs + 1
left = 8.0 * float((ln * ((n - c0 - 1) + (n - p)) * live).sum())
right = 8.0 * float((ln * ((np.minimum(p1 + r, n) - top) + (p1 - top)) * live).sum())
out = {"command": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --kernel-trace -- python3 scratch/r5_ht2.py %d" % n,
       "note": "FETCH_SIZE counts 64 B per 128-B request on gfx950 (x 2), both counters in KiB; algorithmic bytes: every entry a step "
               "touches read once and written once.  ht2_geng_left_kernel: the left application (its first `count` workgroups are the "
               "factorisations, which read the copied blocks); ht2_apply_right_kernel: rows from the group's top on; ht2_wy_right_kernel: "
               "stage 1's right applications, Q / Z of both stages and the deferred rows of A and B (not separable by kernel name)",
       "n": n, "kernels": {}}
for key, alg in (("ht2_geng_left_kernel", left), ("ht2_apply_right_kernel", right), ("ht2_wy_right_kernel", None), ("ht2_genh_kernel", None)):
    fb, wb = 2.0 * f[key][0] * 1024.0, w[key][0] * 1024.0
    e = {"dispatches": f[key][1], "fetched_bytes": fb, "written_bytes": wb}
    if alg:
        e.update({"algorithmic_read_bytes": alg, "algorithmic_written_bytes": alg, "fetch_ratio": fb / alg, "write_ratio": wb / alg,
                  "traffic_ratio": (fb + wb) / (2 * alg)})
    out["kernels"][key] = e
tl = out["kernels"]["ht2_geng_left_kernel"]; tr = out["kernels"]["ht2_apply_right_kernel"]
out["chase_traffic_over_algorithmic"] = (tl["fetched_bytes"] + tl["written_bytes"] + tr["fetched_bytes"] + tr["written_bytes"]) / (2 * (left + right))
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r6_ht2_pmc.json | head -60; tail -2 gpurun_out/r6_ht2_pmc_fetch.log | cut -c1-200
