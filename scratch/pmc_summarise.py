# Summarises the rocprofv3 --pmc passes behind profiles/r2_*.json (run on the GPU box).
#   gemv traffic : FETCH_SIZE and WRITE_SIZE passes over scratch/pmc_run.py (SN_HESS_MAX_PANELS=2)
#   dgemm MFMA   : SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE passes over scratch/gemm_bench.py
import csv, glob, json, sys, collections
def load(d, counter, kernel_substr):
    per = collections.OrderedDict()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
                key = int(r["Dispatch_Id"])
                per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    return per
out = {}
root = sys.argv[1]
fetch = load(root + "/pmc_fetch", "FETCH_SIZE", "hess_gemv_kernel")
write = load(root + "/pmc_write", "WRITE_SIZE", "hess_gemv_kernel")
if fetch and write:
    n, nb = 20000, 312
    alg = 0.0
    launches = len(fetch)
    for k in range(launches):
        p, j = divmod(k, nb)
        R0 = p * nb + 1
        alg += 8.0 * (n - R0) * (n - R0 - j)
    fk, wk = sum(fetch.values()), sum(write.values())
    fb, wb = 2.0 * fk * 1024.0, wk * 1024.0      # FETCH_SIZE counts 64 B per 128-B request on gfx950: x2
    out["gemv"] = {"kernel": "hess_gemv_kernel<16,true,true>", "launches": launches,
        "command": "SN_HESS_MAX_PANELS=2 rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --kernel-trace -- python3 scratch/pmc_run.py",
        "algorithmic_bytes": alg, "FETCH_SIZE_kb_sum": fk, "WRITE_SIZE_kb_sum": wk,
        "fetch_bytes_corrected_x2": fb, "write_bytes": wb, "traffic_bytes": fb + wb,
        "traffic_over_algorithmic": (fb + wb) / alg}
busy = load(root + "/pmc_mfma1", "SQ_VALU_MFMA_BUSY_CYCLES", "dgemm")
act = load(root + "/pmc_mfma2", "GRBM_GUI_ACTIVE", "dgemm")
if busy and act:
    # dispatch order of scratch/gemm_bench.py: each bench() = 1 warm-up + reps launches (split-K adds a memset)
    names = {}
    for f in glob.glob(root + "/pmc_mfma1/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "dgemm" in r["Kernel_Name"]:
                names[int(r["Dispatch_Id"])] = (r["Kernel_Name"][:80], r.get("Grid_Size", ""))
    import math
    groups = collections.OrderedDict()
    for k in busy:
        if k in act:
            # same kernel and grid, different k (the rank-312 and rank-624 updates): split by the MFMA count
            key = names[k] + (round(math.log2(max(busy[k], 1.0)) * 2) / 2,)
            g = groups.setdefault(key, [0.0, 0.0, 0])
            g[0] += busy[k]; g[1] += act[k]; g[2] += 1
    out["dgemm"] = [{"kernel": k[0], "grid": k[1], "dispatches": v[2], "mfma_busy_cycles": v[0], "grbm_gui_active": v[1],
                     "mfma_busy_per_dispatch": v[0] / v[2],
                     "mfma_utilisation": v[0] / (v[1] / 8 * 256 * 4)} for k, v in groups.items()]
print(json.dumps(out, indent=1))
