"""Replays AED windows captured from a real reduction (SN_AED_DUMP, test library) through the host
window kernel, on any machine: time per call, deflation counts, and a fingerprint of the results
(so that a faster kernel can be checked for identical or equivalent output).
usage: python scratch/aed_replay.py gpurun_out/aed_windows.bin [reps] [helpers]"""
import sys, os, time, struct, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import starneig_amd.lib as lib
L = lib.load_test_hooks()
dp = C.POINTER(C.c_double)
P = lambda a: a.ctypes.data_as(dp)
L.sn_internal_aed_window.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, C.c_double, C.c_double, dp, dp, dp, C.POINTER(C.c_int)]
L.sn_internal_aed_window.restype = C.c_int
raw = open(sys.argv[1], "rb").read()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
PAD = int(os.environ.get("PAD", "0"))
if len(sys.argv) > 3 and sys.argv[3] != "0":
    L.sn_internal_helper_session(1)
wins = []; off = 0
while off < len(raw):
    nw, = struct.unpack_from("<i", raw, off); sub, thres = struct.unpack_from("<dd", raw, off + 4); off += 20
    T = np.frombuffer(raw, dtype=np.float64, count=nw * nw, offset=off).reshape(nw, nw).T.copy(order="F"); off += 8 * nw * nw
    wins.append((nw, sub, thres, T))
print(len(wins), "windows, nw", sorted(set(w[0] for w in wins)))
best = 1e9
for r in range(reps):
    tot = 0.0; nds = []; fp = 0.0; worst = 0.0
    for nw, sub, thres, T0 in wins:
        ld = nw + PAD
        Tb = np.zeros((ld, nw), order="F"); Tb[:nw] = T0; T = Tb[:nw]
        Zb = np.zeros((ld, nw), order="F"); Z = Zb[:nw]
        spike = np.zeros(nw); sr = np.zeros(nw); si = np.zeros(nw); out = (C.c_int * 3)()
        t = time.perf_counter()
        L.sn_internal_aed_window(nw, P(Tb), ld, P(Zb), ld, sub, thres, P(spike), P(sr), P(si), out)
        tot += time.perf_counter() - t
        nds.append(out[0])
        if r == 0:
            # similarity residual of the window (the spike column is outside T): Z^T T0 Z = T up to the
            # Hessenberg part's own transformation, which is included in Z
            R = Z.T @ T0 @ Z - T
            worst = max(worst, np.abs(np.tril(R, -2)).max() if False else np.linalg.norm(R) / np.linalg.norm(T0))
            fp += float(np.abs(T).sum()) + float(np.abs(spike).sum())
    best = min(best, tot)
    if r == 0:
        print("deflated total", sum(nds), "first", nds[:12], "worst similarity residual %.2e" % worst, "fingerprint %.12e" % fp)
    print("rep %d: %.3f ms per window" % (r, 1e3 * tot / len(wins)))
print("best %.3f ms per window" % (1e3 * best / len(wins)))
