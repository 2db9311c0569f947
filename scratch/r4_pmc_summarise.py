# Summarises the round-4 rocprofv3 --pmc passes (run on the GPU box by scratch/r4_profiles.sh).
#   python3 r4_pmc_summarise.py <root>  ->  JSON on stdout
#     mfma_in_situ : SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over the WHOLE n = 20000 Hessenberg
#                    reduction (all 65 panels), per GEMM kernel and per panel for the fused update
#     gemv_traffic : FETCH_SIZE / WRITE_SIZE over the first two panels (SN_HESS_MAX_PANELS=2)
#     mfma_alone   : the same two counters over scratch/gemm_bench.py
# utilisation = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
import csv, glob, json, sys, collections

def load(d, counter, want):
    per = collections.OrderedDict()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(w in r["Kernel_Name"] for w in want):
                k = int(r["Dispatch_Id"])
                v = per.setdefault(k, [r["Kernel_Name"].split("(")[0][-72:], int(r.get("Grid_Size", 0) or 0), 0.0])
                v[2] += float(r["Counter_Value"])
    return per

def util(busy, act):
    return busy / (act / 8 * 256 * 4) if act else None

root = sys.argv[1]
out = {}
busy, act = load(root + "/insitu_mfma1", "SQ_VALU_MFMA_BUSY_CYCLES", ["dgemm"]), load(root + "/insitu_mfma2", "GRBM_GUI_ACTIVE", ["dgemm"])
if busy and act:
    groups = collections.OrderedDict()
    fused = []
    for k in sorted(busy):
        if k not in act: continue
        name, grid, b = busy[k]
        g = groups.setdefault(name, [0.0, 0.0, 0])
        g[0] += b; g[1] += act[k][2]; g[2] += 1
        # the fused trailing update is the 128 x 128 N,T kernel with the most MFMA work per workgroup (k = 2 nb)
        if "ILi128ELi128" in name or "<128, 128" in name: fused.append((k, grid, b, act[k][2]))
    out["mfma_in_situ"] = {
        "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES (then GRBM_GUI_ACTIVE) --kernel-trace -- python3 scratch/pmc_run.py 20000   (the whole reduction, 65 panels)",
        "per_kernel": [{"kernel": n, "dispatches": v[2], "mfma_busy_cycles": v[0], "grbm_gui_active": v[1], "mfma_utilisation": util(v[0], v[1])} for n, v in groups.items()],
    }
    # per workgroup MFMA cycles separate the k = 624 fused update from the k = 312 updates of the same kernel
    if fused:
        per_wg = [b / max(g / 256, 1) for _, g, b, _ in fused]
        top = max(per_wg)
        sel = [(k, g, b, a) for (k, g, b, a), w in zip(fused, per_wg) if w > 0.75 * top]
        out["mfma_in_situ"]["fused_trailing_update"] = {
            "dispatches": len(sel), "mfma_busy_cycles": sum(x[2] for x in sel), "grbm_gui_active": sum(x[3] for x in sel),
            "mfma_utilisation": util(sum(x[2] for x in sel), sum(x[3] for x in sel)),
            "first_panels": [round(util(x[2], x[3]), 4) for x in sel[:5]], "last_panels": [round(util(x[2], x[3]), 4) for x in sel[-5:]]}
fetch = load(root + "/pmc_fetch", "FETCH_SIZE", ["hess_gemv_kernel"])
write = load(root + "/pmc_write", "WRITE_SIZE", ["hess_gemv_kernel"])
if fetch and write:
    n, nb = 20000, 312
    alg = 0.0
    for k in range(len(fetch)):
        p, j = divmod(k, nb)
        R0 = p * nb + 1
        alg += 8.0 * (n - R0) * (n - R0 - j)
    fk, wk = sum(v[2] for v in fetch.values()), sum(v[2] for v in write.values())
    fb, wb = 2.0 * fk * 1024.0, wk * 1024.0      # FETCH_SIZE counts 64 B per 128-B request on gfx950: x2
    out["gemv_traffic"] = {"kernel": "hess_gemv_kernel<16,true,true>", "launches": len(fetch),
        "command": "SN_HESS_MAX_PANELS=2 rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --kernel-trace -- python3 scratch/pmc_run.py",
        "algorithmic_bytes": alg, "fetch_bytes_corrected_x2": fb, "write_bytes": wb, "traffic_over_algorithmic": (fb + wb) / alg}
busy, act = load(root + "/alone_mfma1", "SQ_VALU_MFMA_BUSY_CYCLES", ["dgemm"]), load(root + "/alone_mfma2", "GRBM_GUI_ACTIVE", ["dgemm"])
if busy and act:
    groups = collections.OrderedDict()
    for k in sorted(busy):
        if k not in act: continue
        name, grid, b = busy[k]
        key = (name, grid, round(b / 1e9, 1))
        g = groups.setdefault(key, [0.0, 0.0, 0])
        g[0] += b; g[1] += act[k][2]; g[2] += 1
    out["mfma_alone"] = [{"kernel": k[0], "grid": k[1], "dispatches": v[2], "mfma_utilisation": util(v[0], v[1])} for k, v in groups.items()]
print(json.dumps(out, indent=1))
