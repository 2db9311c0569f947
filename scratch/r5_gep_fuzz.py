"""Fuzz of the blocked AED of the QZ path (GepDriver::large_aed): random sizes, AED windows, shift counts, pencils
(the driver's random pencil, a well-conditioned one, planted zeros on R's diagonal, decoupled blocks) through
starneig_GEP_SM_Schur_expert's device twin with aed_parallel_hard_limit = 1; every run under the reference's hooks
(generalized Schur form, eigenvalues hook at 10^3 / 10^4 u, residuals and orthogonality below 500 u).
python scratch/r5_gep_fuzz.py [runs] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
import starneig_amd as S
import oracle as O
from helpers import to_device, to_host, torch_check_pencil
S.node_init(8, 1, S.NO_MESSAGES)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for it in range(runs):
    n = int(rng.choice([257, 400, 700, 1100, 1500, 2200, 3000]))
    kind = rng.choice(["lcg", "wellcond", "zeros", "decouple"])
    aed = int(rng.choice([130, 200, 333, 500, 777, 1000])); aed = min(aed, n - 1)
    shifts = -1 if rng.rand() < 0.6 else int(rng.choice([16, 40, 90]))
    thr = rng.choice([-1.0, -3.0])
    if kind == "wellcond":
        H0, R0 = O.random_pencil_wellcond(n)
    else:
        H0, R0 = O.random_pencil(n)
    H0 = H0.copy(order="F"); R0 = R0.copy(order="F")
    if kind == "zeros":
        for i in rng.choice(n, size=max(2, n // 100), replace=False):
            R0[i, i] = 0.0
    if kind == "decouple":
        for i in rng.choice(np.arange(1, n - 1), size=3, replace=False):
            H0[i + 1, i] = 0.0
    conf = S.schur_init_conf()
    conf.aed_window_size = aed
    conf.aed_parallel_soft_limit = conf.aed_parallel_hard_limit = 1
    conf.left_threshold = float(thr)
    if shifts > 0:
        conf.shift_count = min(shifts, (aed * 9 // 10) // 2 * 2)
    tH, tR = to_device(H0), to_device(R0)
    tQ, tZ = to_device(O.identity(n)), to_device(O.identity(n))
    t0 = time.time()
    rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n, conf=conf)
    torch.cuda.synchronize(); dt = time.time() - t0
    msg = f"run {it}: n={n} {kind} aed={aed} shifts={shifts} thr={thr}: rc={rc} {dt:.2f} s sweeps {st['sweeps']} aeds {st['aeds']}"
    ok = rc == 0
    if ok:
        ra, oq, oz = torch_check_pencil(tQ, tH, tZ, to_device(H0), n)
        rb, _, _ = torch_check_pencil(tQ, tR, tZ, to_device(R0), n)
        Sm, Tm = to_host(tH), to_host(tR)
        form = O.check_gep_schur_form(Sm, Tm)
        er, ei, eb = O.gep_extract_eigenvalues(Sm, Tm)
        hook = O.eigenvalues_check((er, ei, eb), (ar, ai, be))
        msg += f" residuals {ra:.0f}/{rb:.0f} orth {oq:.0f}/{oz:.0f} form {form} hook w{hook['warnings']} f{hook['failures']} inf {int((be == 0).sum())}"
        ok = max(ra, rb, oq, oz) < 500 and form == 0 and hook["failures"] == 0 and hook["warnings"] == 0
    if not ok:
        bad += 1
    print(("OK   " if ok else "BAD  ") + msg, flush=True)
print(f"{bad} bad of {runs}")
S.node_finalize()
