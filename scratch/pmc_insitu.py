# MFMA utilisation per kernel from pairs of rocprofv3 --pmc passes (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)
# over the same command: the Hessenberg GEMM updates in situ (first two panels of n = 20000, with and
# without the side stream) and the aggregated Schur update kernels alone.
#   utilisation = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
import csv, glob, json, sys, collections
def load(d, counter):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = int(r["Dispatch_Id"])
                v = per.setdefault(k, [r["Kernel_Name"].split("(")[0][-70:], r.get("Grid_Size", ""), 0.0])
                v[2] += float(r["Counter_Value"])
    return per
def pair(root, a, b, want):
    busy, act = load(root + "/" + a, "SQ_VALU_MFMA_BUSY_CYCLES"), load(root + "/" + b, "GRBM_GUI_ACTIVE")
    groups = collections.OrderedDict()
    for k in sorted(busy):
        if k in act and any(w in busy[k][0] for w in want):
            g = groups.setdefault((busy[k][0], busy[k][1]), [0.0, 0.0, 0])
            g[0] += busy[k][2]; g[1] += act[k][2]; g[2] += 1
    return [{"kernel": k[0], "grid": k[1], "dispatches": v[2], "mfma_busy_cycles": v[0], "grbm_gui_active": v[1],
             "mfma_utilisation": v[0] / (v[1] / 8 * 256 * 4) if v[1] else None} for k, v in groups.items()]
root = sys.argv[1]
out = {"hessenberg_updates_in_situ_side_stream_on": pair(root, "insitu_mfma1", "insitu_mfma2", ["dgemm"]),
       "hessenberg_updates_in_situ_side_stream_off": pair(root, "noside_mfma1", "noside_mfma2", ["dgemm"]),
       "aggregated_schur_updates_alone": pair(root, "agg_mfma1", "agg_mfma2", ["agg_"])}
print(json.dumps(out, indent=1))
