# Robustness sweep of the round-2 features (blocked AED, reordering, infinite eigenvalues) over odd
# sizes, window sizes and thresholds; prints one line per case, "BAD" marks a violated check.
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import starneig_amd as S, oracle as O
from helpers import to_host, to_device
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(int(os.environ.get("SN_CORES", "16")), 1, S.NO_MESSAGES)
bad = 0
def flag(ok): 
    global bad
    if not ok: bad += 1
    return "ok " if ok else "BAD"
rng = np.random.RandomState(1)
# blocked AED
for n, nw, ns, thr in [(700, 301, 200, -1.0), (1234, 350, 300, -1.0), (1234, 513, 400, -3.0), (3001, 700, 500, -1.0),
                       (3001, 1000, 900, -1.0), (2000, 2000, 1000, -1.0), (5000, 400, 300, -3.0)]:
    tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n, seed=n)
    tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    S.hessenberg_device(tH, tQ, n=n)
    conf = S.schur_init_conf(); conf.aed_window_size, conf.shift_count, conf.left_threshold = nw, ns, thr
    t = time.time(); rc, real, imag, st = S.schur_device(tH, tQ, n=n, conf=conf); torch.cuda.synchronize(); dt = time.time() - t
    _, c = S.check_device(tQ, tH, tA0, n=n)
    form = O.check_schur_form(to_host(tH))
    ok = rc == 0 and c["residual_u"] < 500 and c["orthogonality_u"] < 500 and c["below_subdiagonal"] == 0 and form == 0 \
        and abs(real.sum() - float(torch.diagonal(tA0[:, :n]).sum())) < 1e-8 * n
    print(flag(ok), f"blocked AED n={n} nw={nw} ns={ns} thr={thr}: rc={rc} {dt:.2f}s aeds={st['aeds']} sweeps={st['sweeps']} res={c['residual_u']:.0f} orth={c['orthogonality_u']:.0f} form={form}", flush=True)
# reordering with random selections
for n, frac, win, vpc in [(333, 0.1, -1, -1), (1000, 0.9, -1, -1), (1501, 0.5, 20, 7), (2500, 0.02, -1, -1), (800, 0.5, 128, 126)]:
    tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n, seed=3 * n)
    tS = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    S.hessenberg_device(tS, tQ, n=n)
    rc, real, imag, _ = S.schur_device(tS, tQ, n=n)
    sel = (rng.rand(n) < frac).astype(np.int32)
    for i in range(n - 1):
        if imag[i] > 0: sel[i] = sel[i + 1] = max(sel[i], sel[i + 1])
    want = sorted(zip(real[sel == 1], np.abs(imag[sel == 1])))
    k = int(sel.sum())
    conf = S.reorder_init_conf(); conf.window_size, conf.values_per_chain = win, vpc
    rc, r2, i2, st = S.reorder_schur_device(tS, tQ, sel, n=n, conf=conf)
    _, c = S.check_device(tQ, tS, tA0, n=n)
    got = sorted(zip(r2[:k], np.abs(i2[:k])))
    err = max([abs(a[0] - b[0]) + abs(a[1] - b[1]) for a, b in zip(want, got)] + [0.0]) / max(1e-300, np.abs(real).max())
    ok = rc == 0 and c["residual_u"] < 500 and c["orthogonality_u"] < 500 and O.check_schur_form(to_host(tS)) == 0 \
        and int(sel.sum()) == k and np.all(sel[:k] == 1) and err < 1e-9
    print(flag(ok), f"reorder n={n} frac={frac} win={win} vpc={vpc}: rc={rc} k={k} windows={st['windows']} res={c['residual_u']:.0f} orth={c['orthogonality_u']:.0f} eig err {err:.1e}", flush=True)
# infinite eigenvalues at random positions
for n, nz in [(257, 3), (1000, 25), (2222, 40)]:
    H0, R0 = O.random_pencil_wellcond(n, seed=n)
    zs = sorted(rng.choice(n, nz, replace=False).tolist())
    for k in zs: R0[k, k] = 0.0
    tH, tR = to_device(H0), to_device(R0); tH0, tR0 = tH.clone(), tR.clone()
    tQ, tZ = S.device_matrix(n, ld=H0.shape[0]), S.device_matrix(n, ld=H0.shape[0])
    S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
    rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n)
    _, ca = S.check_pencil_device(tQ, tH, tZ, tH0, n=n); _, cb = S.check_pencil_device(tQ, tR, tZ, tR0, n=n)
    ninf = int((be == 0).sum())
    ok = rc == 0 and ca["residual_u"] < 500 and cb["residual_u"] < 500 and ca["orthogonality_q_u"] < 500 and ca["orthogonality_z_u"] < 500 \
        and O.check_gep_schur_form(to_host(tH), to_host(tR)) == 0 and 1 <= ninf <= nz
    print(flag(ok), f"QZ inf n={n} zeros={nz}: rc={rc} infinite={ninf} resA={ca['residual_u']:.0f} resB={cb['residual_u']:.0f}", flush=True)
print("violations:", bad)
