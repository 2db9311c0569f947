"""Worker of the multi-process sharded-Hessenberg test (launched by torch.distributed.run).
All ranks share cuda:0 and talk over gloo, so the N > 1 code path -- block-column ownership,
the per-column all-reduce of y, the panel broadcast, the W all-reduce, the final assembly --
runs on a single-GPU box.  Rank 0 compares with the single-GPU path and the oracle checks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def main():
    n, pw = int(sys.argv[1]), int(sys.argv[2])
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    torch.cuda.set_device(0)
    if backend == "nccl":          # (one rank per GPU: RCCL does not share a device between ranks)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.zeros(1, device="cuda")
    import starneig_amd as S
    from starneig_amd import distributed as D
    S.node_init(1, 1, S.NO_MESSAGES)
    tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n, seed=2019)
    tA = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    rc, st = D.hessenberg_sharded(tA, tQ, n=n, panel_width=pw)
    torch.cuda.synchronize()
    assert rc == 0, rc
    # every rank holds the same assembled result
    chk = torch.stack([tA.abs().sum(), tQ.abs().sum()])
    chk = chk if backend == "nccl" else chk.cpu()
    ref = chk.clone(); dist.broadcast(ref, src=0)
    assert torch.allclose(chk, ref, rtol=1e-13), (rank, chk, ref)
    if rank == 0:
        rc, c = S.check_device(tQ, tA, tA0, n=n)
        assert rc == 0 and c["below_subdiagonal"] == 0, c
        assert c["residual_u"] < 500 and c["orthogonality_u"] < 500, c
        tB = tA0.clone(); tQ1 = S.device_matrix(n); S.set_matrix_device(tQ1, n, n, 0.0, 1.0)
        assert S.hessenberg_device(tB, tQ1, n=n, panel_width=pw) == 0
        scale = torch.linalg.norm(tA0[:, :n]).item()
        diff = (tA[:, :n] - tB[:, :n]).abs().max().item() / scale
        assert diff <= 8 * np.sqrt(n) * 2.0 ** -52, diff
        if backend == "nccl":
            assert st["collectives"] == "RCCL, called from the library", st["collectives"]
        print(f"DIST-OK world={world} n={n} pw={pw} residual={c['residual_u']:.1f}u "
              f"diff={diff / 2.0**-52:.1f}u collectives={st['collectives']}", flush=True)
    dist.barrier()
    # ---- Schur leg: replicas of H, row-sharded accumulation of Q ---------------------------------
    tH0, tQ0 = tA.clone(), tQ.clone()
    rc, real, imag, sst = D.schur_sharded(tA, tQ, n=n)
    torch.cuda.synchronize()
    assert rc == 0, rc
    chk = torch.stack([tA.abs().sum(), tQ.abs().sum()])
    chk = chk if backend == "nccl" else chk.cpu()
    ref = chk.clone(); dist.broadcast(ref, src=0)
    assert torch.equal(chk, ref), (rank, chk, ref)          # replicas are bit-identical
    if rank == 0:
        rc, c = S.check_device(tQ, tA, tA0, n=n)
        assert rc == 0 and c["below_subdiagonal"] == 0, c
        assert c["residual_u"] < 500 and c["orthogonality_u"] < 500, c
        # equals the single-GPU Schur reduction of the same Hessenberg matrix bit for bit
        rc, real1, imag1, _ = S.schur_device(tH0, tQ0, n=n)
        assert rc == 0
        assert torch.equal(tH0, tA) and torch.equal(tQ0[:, :n], tQ[:, :n])
        assert np.array_equal(real, real1) and np.array_equal(imag, imag1)
        print(f"DIST-SCHUR-OK world={world} n={n} residual={c['residual_u']:.1f}u rows={sst['q_rows']}",
              flush=True)
    dist.barrier()
    D.shutdown()
    S.node_finalize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
