"""A/B of two builds of the library on ONE box: the Hessenberg leg (and optionally the Schur leg) at n, alternating
between the in-tree library and the one named by SN_AB_LIB, each in a child process of its own.
usage: python scratch/ab_hess.py n reps [schur]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import starneig_amd.lib as L
if os.environ.get("SN_AB_LIB"):
    L.LIB_PATH = os.environ["SN_AB_LIB"]
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(8, 1, S.NO_MESSAGES)
n = int(sys.argv[2]); reps = int(sys.argv[3]); schur = len(sys.argv) > 4
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
for r in range(reps):
    tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t = time.time()
    rc, st = S.hessenberg_device(tH, tQ, n=n, stats=True, sample_every=16)
    torch.cuda.synchronize(); dt = time.time() - t
    bw = st["gemv_sampled_bytes"] / (st["gemv_sampled_ms"] * 1e-3) / 1e9 if st["gemv_sampled_ms"] else 0
    line = "hess n=%d %.3f s gemv %.0f GB/s" % (n, dt, bw)
    if schur:
        torch.cuda.synchronize(); t = time.time()
        rc, re, im, st2 = S.schur_device(tH, tQ, n=n)
        torch.cuda.synchronize(); line += "  schur %.3f s" % (time.time() - t)
    print(os.environ.get("SN_AB_LIB", "in-tree")[-24:], line, flush=True)
"""
n, reps = sys.argv[1], sys.argv[2]
for rnd in range(2):
    for lib in ("", os.path.join(ROOT, "scratch", "ab", os.environ.get("AB_NAME", "libstarneig_amd_r5.so"))):
        env = dict(os.environ)
        if lib: env["SN_AB_LIB"] = lib
        else: env.pop("SN_AB_LIB", None)
        subprocess.run([sys.executable, "-c", CHILD, ROOT, n, reps] + sys.argv[3:], env=env)
