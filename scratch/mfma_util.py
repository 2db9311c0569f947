# MFMA utilisation of the dgemm kernel from two rocprofv3 --pmc passes (scratch/gemm_bench.py):
#   pass 1: SQ_VALU_MFMA_BUSY_CYCLES   pass 2: GRBM_GUI_ACTIVE
# utilisation = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
import csv, sys, glob, collections
def load(d, counter):
    out = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "dgemm_kernel" in r["Kernel_Name"]:
                out[(r["Kernel_Name"][:60], r["Grid_Size"] if "Grid_Size" in r else "")] += float(r["Counter_Value"])
    return out
a = load(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES")
b = load(sys.argv[2], "GRBM_GUI_ACTIVE")
for k in a:
    if k in b and b[k] > 0:
        print(k, "MFMA busy %.3e  GRBM %.3e  utilisation %.1f %%" % (a[k], b[k], 100.0 * a[k] / (b[k] / 8 * 256 * 4)))
