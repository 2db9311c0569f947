# Duration of one chain-kernel launch (256 rotations; variants 8/9: 64) -- see sn_internal_ht_chain_bench.
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
L = S.lib.load_test_hooks()
L.sn_internal_ht_chain_bench.restype = C.c_double
L.sn_internal_ht_chain_bench.argtypes = [C.c_int, C.c_int]
names = {0: "4 group waves + 12 followers", 1: "4 group waves, no followers", 2: "4 + 4 followers", 3: "no loads", 4: "no result stores",
         5: "no safe-range branch", 6: "no loads, no stores, no branch", 7: "6 + no barrier, no apply", 8: "one wave, 64 rotations",
         9: "one wave, 64 rotations, bare"}
names[10] = 'timestamps, full kernel'; names[12] = '8 group waves, no followers, 512 rotations'; names[0] = '4 group waves + 11 followers'
for v in (sys.argv[1:] and [int(a) for a in sys.argv[1:]]) or range(10):
    us = L.sn_internal_ht_chain_bench(v, 20)
    steps = {8: 64, 12: 512}.get(v, 256)
    print(f"variant {v} ({names[v]}): {us:.1f} us per launch = {us * 1e3 / steps:.0f} ns per rotation", flush=True)
