"""Does a Hessenberg + Schur run earlier in the process (its streams, its workspaces) change the time of the
two-stage Hessenberg-triangular reduction?  (bench.py's secondary line measured 6.05 s at n = 8000 after the
n = 20000 run, the stand-alone script 5.47 s on the same box.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
import starneig_amd as S
S.node_init(S.USE_ALL, 1, S.NO_MESSAGES)
def ht(n):
    tA, tB = S.device_matrix(n), S.device_matrix(n)
    S.lcg_fill_device(tA, n, n, seed=2019); S.lcg_fill_device(tB, n, n, seed=77)
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t0 = time.time()
    rc, st = S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
    torch.cuda.synchronize()
    return time.time() - t0, st
what = sys.argv[1] if len(sys.argv) > 1 else "none"
if what in ("sep", "both"):
    n = 6000
    tA, tQ = S.device_matrix(n), S.device_matrix(n)
    S.lcg_fill_device(tA, n, n, seed=2019); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    S.hessenberg_device(tA, tQ, n=n)
    S.schur_device(tA, tQ, n=n)
    torch.cuda.synchronize()
    del tA, tQ
if what in ("qz", "both"):
    n = 1000
    tA, tB = S.device_matrix(n), S.device_matrix(n)
    S.lcg_fill_device(tA, n, n, seed=2019); S.lcg_fill_device(tB, n, n, seed=77)
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
    S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
    S.gep_schur_device(tA, tB, tQ, tZ, n=n)
    torch.cuda.synchronize()
    import threading, subprocess
    print("threads of this process:", subprocess.run(["bash", "-c", f"ls /proc/{os.getpid()}/task | wc -l"], capture_output=True, text=True).stdout.strip(),
          "load:", open("/proc/loadavg").read().split()[:3], flush=True)
for rep in range(2):
    dt, st = ht(8000)
    print(f"after {what}: n=8000 {dt:.3f} s (QR {st['qr_ms']/1e3:.3f}, stage 1 {st['stage1_ms']/1e3:.3f}, stage 2 {(st['rotation_ms'] - st['stage1_ms'])/1e3:.3f})", flush=True)
S.node_finalize()
