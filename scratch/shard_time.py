"""What ONE rank of an N-GPU Schur leg does, timed on one GPU: a replica of H, n/N rows of Q, and (new)
only its own column tiles of the deflated part of H.  usage: shard_time.py n world [rank]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import starneig_amd.lib as S

n = int(sys.argv[1]); world = int(sys.argv[2]); rank = int(sys.argv[3]) if len(sys.argv) > 3 else world // 2
S.node_init(64, 1, S.NO_MESSAGES)
tA0 = S.device_matrix(n)
S.lcg_fill_device(tA0, n, n, seed=2019, mode=0)
tH0 = tA0.clone(); tQ0 = S.device_matrix(n)
S.set_matrix_device(tQ0, n, n, 0.0, 1.0)
assert S.hessenberg_device(tH0, tQ0, n=n) == 0
chunk = -(-(-(-n // world)) // 128) * 128
r0 = min(n, rank * chunk); rows = min(n, r0 + chunk) - r0
for label, w in (("rows of Q only", 1), ("rows of Q + own tiles of H", world), ("rows of Q only", 1), ("rows of Q + own tiles of H", world)):
    tH, tQ = tH0.clone(), tQ0.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc, real, imag, st = S.schur_sharded_device(tH, (tQ.data_ptr() + 8 * r0, tQ.shape[1]), rows, rank if w > 1 else 0, w, n=n)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"n={n} rank {rank} of {world} ({rows} rows of Q), {label}: {t:.3f} s  rc={rc} sweeps={st['sweeps']} aeds={st['aeds']} "
          f"aed_host={st['aed_host_s']:.2f}s", flush=True)
tH, tQ = tH0.clone(), tQ0.clone()
torch.cuda.synchronize(); t0 = time.perf_counter()
rc, real, imag, st = S.schur_device(tH, tQ, n=n)
torch.cuda.synchronize(); print(f"single GPU, everything: {time.perf_counter() - t0:.3f} s")
