"""CPU (world_size 2, gloo): the ownership plan of the sharded reduction and the collective
plumbing (ctypes callbacks -> torch.distributed on buffer slices) without any GPU compute."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

from starneig_amd.distributed import owned_column_blocks, owned_h_columns, owned_q_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ownership_partitions_the_matrix():
    for n, block, world in [(20000, 312, 8), (1000, 96, 3), (333, 40, 2), (7, 8, 4)]:
        cols = []
        for r in range(world):
            cols += [c for a, b in owned_column_blocks(n, block, world, r) for c in range(a, b)]
        assert sorted(cols) == list(range(n))
        rows = []
        for r in range(world):
            lo, hi = owned_q_rows(n, world, r)
            rows += list(range(lo, hi))
        assert sorted(rows) == list(range(n))
        # Schur leg: the 128-column tiles of the deflated part of H, tile T with rank T mod world
        hcols = owned_h_columns(n, world)
        assert sorted(int(c) for cs in hcols for c in cs) == list(range(n))
        for r, cs in enumerate(hcols):
            assert all((int(c) // 128) % world == r for c in cs)
    # load balance of the trailing matrix at n=20000 on 8 GPUs: every rank owns 8 +- 1 blocks
    counts = [len(owned_column_blocks(20000, 312, 8, r)) for r in range(8)]
    assert max(counts) - min(counts) <= 1


WORKER = textwrap.dedent('''
    import ctypes as C, os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from starneig_amd.distributed import Collectives
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    y = torch.arange(10, dtype=torch.float64) * (rank + 1)
    p = torch.full((6,), float(rank + 5), dtype=torch.float64)
    coll = Collectives({0: y, 1: p})
    # drive the callbacks exactly as the C library does: (ctx, buffer, offset, count[, root])
    coll.allreduce_cb(None, 0, 2, 5)
    coll.broadcast_cb(None, 1, 1, 3, 1)
    exp = torch.arange(10, dtype=torch.float64) * (rank + 1)
    exp[2:7] = torch.arange(2, 7, dtype=torch.float64) * 3          # ranks 0,1 -> factors 1+2
    assert torch.equal(y, exp), (rank, y)
    ep = torch.full((6,), float(rank + 5), dtype=torch.float64); ep[1:4] = 6.0
    assert torch.equal(p, ep), (rank, p)
    assert coll.calls == {"allreduce": 1, "broadcast": 1, "bytes": 64}
    # assembly of the Schur form from the owners' column tiles (schur_sharded): each rank holds
    # garbage in the tiles it does not own, the right values in its own
    from starneig_amd.distributed import assemble_h_tiles, owned_h_columns
    n, ld = 700, 704
    full = torch.arange(n * ld, dtype=torch.float64).reshape(n, ld)
    mine = torch.full((n, ld), -1.0 - rank, dtype=torch.float64)
    own = owned_h_columns(n, 2)[rank]
    mine[own] = full[own]
    assemble_h_tiles(mine, n)
    assert torch.equal(mine, full), rank
    print("PLUMBING-OK", rank, flush=True)
    dist.destroy_process_group()
''') % ROOT


def test_collective_callbacks_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("PLUMBING-OK") == 2
