"""Why is the Schur leg 0.13 s faster behind the sharded Hessenberg path (world 1) than behind the single-GPU one?
Same process, every combination of {hessenberg_device, hessenberg_sharded} x {schur_device, schur_sharded}."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import starneig_amd as S
from starneig_amd import distributed as D
S.node_init(S.USE_ALL, 1, S.NO_MESSAGES)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n, seed=2019, mode=0)
tA = torch.empty_like(tA0); tQ = S.device_matrix(n)
for rep in range(2):
    for hs in (0, 1):
        for ss in (0, 1):
            tA.copy_(tA0); S.set_matrix_device(tQ, n, n, 0.0, 1.0); torch.cuda.synchronize()
            t0 = time.perf_counter()
            if hs: rc, st = D.hessenberg_sharded(tA, tQ, n=n)
            else: rc, st = S.hessenberg_device(tA, tQ, n=n, stats=True)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if ss: rc, re, im, sst = D.schur_sharded(tA, tQ, n=n)
            else: rc, re, im, sst = S.schur_device(tA, tQ, n=n)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            print(f"hessenberg {'sharded' if hs else 'single '} {t1 - t0:.3f} s | schur {'sharded' if ss else 'single '} {t2 - t1:.3f} s "
                  f"sweeps {sst['sweeps']} aeds {sst['aeds']} aed_host {sst['aed_host_s']:.3f} wait {sst['gpu_wait_s']:.3f} "
                  f"device {sst['total_ms'] / 1e3:.3f}", flush=True)
dist.destroy_process_group()
