# One wavefront of aggregated tile updates alone on the GPU (schur_agg.h): microseconds and TFLOP/s.
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
from starneig_amd import lib as L_
L_.LIB_PATH = L_.TEST_LIB_PATH
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
L = L_.load()
L.sn_internal_agg_bench.restype = C.c_double
L.sn_internal_agg_bench.argtypes = [C.c_int] * 5
for mode in (0, 1):
    for rows, ntiles, W in ((20000, 8, 446), (20000, 4, 446), (20000, 1, 446), (20000, 8, 396), (20000, 8, 246), (8000, 8, 446)):
        us = L.sn_internal_agg_bench(mode, rows, ntiles, W, 10)
        fl = 2.0 * rows * W * W * ntiles
        by = 16.0 * rows * W * ntiles
        print(f"mode {mode} rows {rows} tiles {ntiles} W {W}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  {by / us / 1e3:7.1f} GB/s", flush=True)
