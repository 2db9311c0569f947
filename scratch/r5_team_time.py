"""Sharded Hessenberg reduction through the one-process team with virtual ranks on ONE device: the per-column
exchange on the host (round 4: stream synchronise + two barriers + a kernel per column) against the device-side
exchange (round 5: peer stores + flags).  python scratch/r5_team_time.py [n] [ranks...]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n, gpus):
    import numpy as np
    import torch
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
    import starneig_amd as S
    import oracle as O
    os.environ["STARNEIG_AMD_VIRTUAL_GPUS"] = str(gpus)
    S.node_init(8, gpus, S.NO_MESSAGES)
    A0 = O.random_fullpos(n)
    out = []
    for rep in range(3):
        A = A0.copy(order="F"); Q = O.identity(n)
        t0 = time.time()
        rc = S.SEP_SM_Hessenberg(n, A, A.shape[0], Q, Q.shape[0])
        out.append(round(time.time() - t0, 3))
        assert rc == 0
    res = O.residual_u(Q, A, A0) if n <= 4000 else -1.0
    S.node_finalize()
    print("RESULT " + json.dumps({"n": n, "ranks": gpus, "mode": os.environ.get("STARNEIG_AMD_TEAM_EXCHANGE", "default"),
                                  "host_api_s": out, "residual_u": res}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
        ranks = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
        for g in ranks:
            for mode in (["default"] if g == 1 else ["host", "device"]):
                env = dict(os.environ)
                if mode != "default":
                    env["STARNEIG_AMD_TEAM_EXCHANGE"] = mode
                p = subprocess.run([sys.executable, __file__, "child", str(n), str(g)], env=env, capture_output=True, text=True, timeout=1200)
                line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
                print(line[0][7:] if line else ("FAILED " + p.stderr[-800:]), flush=True)
