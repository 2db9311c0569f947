# One lazy Q update launch of the Schur leg alone on the GPU: time, HBM rate, flop rate
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
L = S.lib.load_test_hooks()
L.sn_internal_qupdate_bench.restype = C.c_double
L.sn_internal_qupdate_bench.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
for rbm, chains in [(r, c) for r in (128, 64) for c in (1, 4, 10, 20)]:
    us = L.sn_internal_qupdate_bench(20000, chains, 20, rbm)
    byt = 2.0 * 20000 * 96 * 8 * chains; fl = 2.0 * 20000 * 96 * 96 * chains
    print(f"Q update, row tile {rbm:3d}, {chains:2d} windows: {us:7.1f} us = {byt / us / 1e6:.2f} TB/s, {fl / us / 1e6:.1f} TFLOP/s", flush=True)
