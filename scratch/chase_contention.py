"""What slows the chase kernel in situ (164-184 us per launch under the lazy updates, 97 us alone)?
The kernel alone, beside an HBM-bound copy stream, beside an fp64 GEMM stream (MFMA + power), beside small-footprint
compute.  usage: chase_contention.py [chains]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
L = S.lib.load_test_hooks()
L.sn_internal_chase_bench.restype = C.c_double
L.sn_internal_chase_bench.argtypes = [C.c_int, C.c_int, C.c_int]
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 29
side = torch.cuda.Stream()
x = torch.empty(1 << 29, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)      # 4 GB each
a = torch.randn(8192, 8192, dtype=torch.float64, device="cuda"); b = torch.randn_like(a); c = torch.empty_like(a)
small = torch.randn(2048, 2048, dtype=torch.float64, device="cuda"); small2 = torch.empty_like(small)

def background(kind):
    with torch.cuda.stream(side):
        if kind == "copy":
            for _ in range(40): y.copy_(x)                  # ~8 GB of traffic each
        elif kind == "gemm":
            for _ in range(12): torch.matmul(a, b, out=c)   # ~1.1 TFLOP each
        elif kind == "small gemm":
            for _ in range(600): torch.matmul(small, small, out=small2)

for kind in ("none", "copy", "gemm", "small gemm", "none"):
    torch.cuda.synchronize()
    if kind != "none":
        background(kind); background(kind) if kind != "copy" else None
    t0 = time.perf_counter()
    us = L.sn_internal_chase_bench(chains, 100, 0)
    busy = not side.query()
    torch.cuda.synchronize()
    print(f"chase kernel ({chains} chains) beside {kind:10s}: {us:6.1f} us per launch   (background still running at the end: {busy})", flush=True)
