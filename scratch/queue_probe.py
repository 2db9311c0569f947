"""Does the Schur leg depend on what else in the process created streams (hardware-queue assignment)?
usage: queue_probe.py SETUP   with SETUP in plain | pg | pg_barrier | hiK | normK | lowK  (K dummy streams created first)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
setup = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
keep = []
if setup.startswith("pg"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    if setup == "pg_barrier":
        dist.barrier(); torch.cuda.synchronize()
elif setup[:2] == "hi":
    keep = [torch.cuda.Stream(priority=-1) for _ in range(int(setup[2:]))]
elif setup[:4] == "norm":
    keep = [torch.cuda.Stream(priority=0) for _ in range(int(setup[4:]))]
elif setup[:3] == "low":
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    for _ in range(int(setup[3:])):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, 1) == 0
        keep.append(h)
for s in keep:
    if hasattr(s, "synchronize"):
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
torch.cuda.synchronize()
import starneig_amd as S
S.node_init(S.USE_ALL, 1, S.NO_MESSAGES)
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n, seed=2019, mode=0)
tA = torch.empty_like(tA0); tQ = S.device_matrix(n)
res = []
for rep in range(3):
    tA.copy_(tA0); S.set_matrix_device(tQ, n, n, 0.0, 1.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc, st = S.hessenberg_device(tA, tQ, n=n, stats=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rc, re, im, sst = S.schur_device(tA, tQ, n=n)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res.append(f"{t1 - t0:.3f}+{t2 - t1:.3f} (wait {sst['gpu_wait_s']:.2f} aed {sst['aed_host_s']:.2f})")
print(f"{setup:11s}: " + "  ".join(res), flush=True)
