import sys, os, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, starneig_amd as S
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
    S.node_init(1,1,S.NO_MESSAGES)
    n = 1000
    tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
    tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    S.hessenberg_device(tH, tQ, n=n); torch.cuda.synchronize()
    rc, real, imag, st = S.schur_device(tH, tQ, n=n); torch.cuda.synchronize()
    np.save(sys.argv[2], tH.cpu().numpy()[:, :n].T); np.save(sys.argv[2] + "q", tQ.cpu().numpy()[:, :n].T)
    sys.exit(0)
for k in (3, 5, 8, 11, 14):
    env = dict(os.environ, SN_SCHUR_MAX_SWEEPS=str(k))
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/par.npy"], env=env)
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/ser.npy"], env=dict(env, SN_SCHUR_SERIAL="1"))
    a = np.load("/tmp/par.npy"); b = np.load("/tmp/ser.npy"); d = np.abs(a - b)
    qa = np.load("/tmp/par.npyq.npy"); qb = np.load("/tmp/ser.npyq.npy")
    idx = np.argwhere(d > 1e-9 * np.abs(b).max())
    print("sweeps", k, "H max diff", d.max(), "count", len(idx), "Q max diff", np.abs(qa - qb).max())
    if len(idx):
        print(" rows", idx[:, 0].min(), idx[:, 0].max(), "cols", idx[:, 1].min(), idx[:, 1].max())
        # histogram of differing rows / cols
        rows = np.unique(idx[:, 0]); cols = np.unique(idx[:, 1])
        print(" distinct rows", len(rows), rows[:20], "... distinct cols", len(cols), cols[:20])
