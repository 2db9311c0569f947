# does an initialised RCCL process group slow the Schur leg down?  (world 1)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29688")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
mode = sys.argv[1]
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
if mode != "none":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    if mode == "used":
        t = torch.ones(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
import starneig_amd as S
S.node_init(1, 1, S.NO_MESSAGES)
n = 20000
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
S.hessenberg_device(tH, tQ, n=n)
torch.cuda.synchronize(); t = time.time()
rc, real, imag, st = S.schur_device(tH, tQ, n=n)
torch.cuda.synchronize()
print(mode, "schur %.2fs" % (time.time() - t), "aed_host %.2f wait %.2f" % (st["aed_host_s"], st["gpu_wait_s"]), flush=True)
