import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import starneig_amd as S, oracle as O
sys.path.insert(0, "tests")
dp = C.POINTER(C.c_double)
L = S.lib.load_test_hooks()
L.sn_internal_aed_window.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, C.c_double, C.c_double, dp, dp, dp, C.POINTER(C.c_int)]
def P(a): return a.ctypes.data_as(dp)
for nw in (128, 192, 256, 384):
    W0 = np.asfortranarray(O.random_hessenberg(nw, seed=11, ld=nw))
    thres = 2.0**-52 * np.linalg.norm(W0) * 20
    best = 1e9
    for rep in range(3):
        T = W0.copy(order="F"); Z = np.zeros((nw, nw), order="F")
        spike = np.zeros(nw); sr = np.zeros(nw); si = np.zeros(nw); out = (C.c_int * 3)()
        t = time.perf_counter()
        L.sn_internal_aed_window(nw, P(T), nw, P(Z), nw, 1e-3, thres, P(spike), P(sr), P(si), out)
        best = min(best, time.perf_counter() - t)
    print(nw, "%.1f ms" % (best * 1e3), "deflated", out[0], "shifts", out[1])
