"""Reproducer / stress run of the block-column sharded Hessenberg reduction through the one-process
team (csrc/node_team.hip) with virtual ranks on ONE device ("an unexplained wrong
result in the config-4 code path").

    python scratch/r5_shard_stress.py                 # every (stream mode, fold) cell, 10 repetitions each
    python scratch/r5_shard_stress.py child <reps> <gpus> <n>   # one cell (the environment selects it)

Each cell runs in a process of its own (the developer switches are read once):
  SN_STREAM_MODE 0 pooled streams / 1 critical streams on hardware queues of their own
  SN_HESS_FOLD   0 ticket fold behind agent-scope release / acquire (shipped)
                 1 ticket fold on sc1 accesses alone (rounds 3-4)
                 2 fold by a launch of its own (no in-launch hand-off at all)
Per repetition: max|H - H_oracle| / ||A||_F in units of the test tolerance, and whether H and Q are
bit-identical to the first repetition.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(reps, gpus, n):
    import numpy as np
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    import starneig_amd as S
    import oracle as O
    from helpers import elementwise_tolerance
    os.environ["STARNEIG_AMD_VIRTUAL_GPUS"] = str(gpus)
    S.node_init(4, gpus, S.NO_MESSAGES)
    assert S.lib.load().starneig_node_get_gpus() == gpus
    A0 = O.random_fullpos(n)
    Ao = A0.copy(order="F"); Qo = O.identity(n)
    O.hessenberg(Ao, Qo)
    tol = elementwise_tolerance(n)
    nrm = np.linalg.norm(A0[:n])
    first = None
    out = []
    for r in range(reps):
        A = A0.copy(order="F"); Q = O.identity(n)
        t0 = time.time()
        rc = S.SEP_SM_Hessenberg(n, A, A.shape[0], Q, Q.shape[0])
        dt = time.time() - t0
        err = float(np.abs(A[:n] - Ao[:n]).max() / nrm / tol)
        below = int(O.count_below_subdiagonal(A))
        if first is None:
            first = (A.copy(), Q.copy())
        same = bool(np.array_equal(A, first[0]) and np.array_equal(Q, first[1]))
        bad_cols = []
        if err > 1.0:
            bad = np.argwhere(np.abs(A[:n] - Ao[:n]) / nrm > tol)
            bad_cols = sorted(set(int(c) for c in bad[:, 1]))[:8]
        out.append({"rc": rc, "err_over_tol": round(err, 4), "below": below, "bits_equal_first": same,
                    "s": round(dt, 2), "first_bad_cols": bad_cols})
    S.node_finalize()
    print("RESULT " + json.dumps(out))


def main():
    reps = int(os.environ.get("REPS", "10"))
    cells = [(m, f) for m in (0, 1) for f in (1, 0, 2)]
    summary = []
    for mode, fold in cells:
        env = dict(os.environ, STARNEIG_AMD_TUNING="1", SN_STREAM_MODE=str(mode), SN_HESS_FOLD=str(fold))
        t0 = time.time()
        p = subprocess.run([sys.executable, __file__, "child", str(reps), "4", "2000"], env=env,
                           capture_output=True, text=True, timeout=1500)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        res = json.loads(line[0][7:]) if line else None
        wrong = None if res is None else sum(1 for r in res if r["err_over_tol"] > 1.0 or r["below"] or r["rc"])
        diff = None if res is None else sum(1 for r in res if not r["bits_equal_first"])
        summary.append({"stream_mode": mode, "fold": fold, "rc": p.returncode, "wrong": wrong,
                        "not_bit_identical_to_first": diff, "wall_s": round(time.time() - t0, 1), "runs": res})
        print(json.dumps(summary[-1]), flush=True)
        if res is None:
            print(p.stdout[-2000:], p.stderr[-2000:], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        main()
