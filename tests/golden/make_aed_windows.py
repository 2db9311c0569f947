"""How tests/golden/aed_windows_lcg20000.npz was made: two AED windows (160 x 160, with the coupling
entry and the deflation threshold that came with them) as they arrive at the host window kernel
during the reduction of BASELINE config 3 (n = 20000, LCG seed 2019).

On a GPU box, with the test library standing in for the product (scratch/schur_configs.py):
    SN_USE_TEST_LIB=1 SN_AED_DUMP=$PWD/gpurun_out/aed_windows.bin python scratch/schur_configs.py 20000 160,106,-1
then here:
    python tests/golden/make_aed_windows.py gpurun_out/aed_windows.bin
(records 3 and 20 of the dump: a window early in the reduction and one from the middle)."""
import sys, struct, os
import numpy as np

raw = open(sys.argv[1], "rb").read()
off = 0; wins = []
while off < len(raw):
    nw, = struct.unpack_from("<i", raw, off); sub, thres = struct.unpack_from("<dd", raw, off + 4); off += 20
    T = np.frombuffer(raw, dtype=np.float64, count=nw * nw, offset=off).reshape(nw, nw).T.copy(); off += 8 * nw * nw
    wins.append((sub, thres, T))
sel = [wins[3], wins[20]]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aed_windows_lcg20000.npz")
np.savez_compressed(out, subs=np.array([w[0] for w in sel]), thres=np.array([w[1] for w in sel]),
                    windows=np.stack([w[2] for w in sel]))
print(out, os.path.getsize(out))
