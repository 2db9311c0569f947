import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
L = S.lib.load_test_hooks()
L.sn_internal_chase_bench.restype = C.c_double
L.sn_internal_chase_bench.argtypes = [C.c_int, C.c_int, C.c_int]
for chains in (1, 29):
    print(f"chase kernel, {chains:2d} chains: {L.sn_internal_chase_bench(chains, 20, 0):.1f} us per launch", flush=True)
for dbg, what in ((1, "no left phase"), (2, "no right bulk"), (3, "no left, no right bulk"), (4, "no reflector lane"), (7, "barriers + load/store only")):
    print(f"  variant {dbg} ({what}): {L.sn_internal_chase_bench(8, 20, dbg):.1f} us", flush=True)
