"""Multi-GPU driver of the sharded Hessenberg reduction: one process per GPU over RCCL / xGMI.

The C library does all the compute.  Its two collectives (all-reduce-sum, broadcast) go to RCCL
directly from C++ on the reduction's own stream when the process group runs on "nccl" (= RCCL):
this module only carries the communicator's unique id to the ranks (one torch.distributed
broadcast) and checks the new communicator with a small all-reduce and broadcast before using
it.  Otherwise (gloo, a sub-group, SN_NATIVE_RCCL=0, or a failed check) the library calls back
into this module and the collectives are torch.distributed calls on buffers allocated here --
one Python round trip per panel column.  torch is plumbing only (device memory, the stream, the
process group).

Communication per reduction of an n x n matrix with panel width nb:
  per panel : broadcast of the owner's panel columns  (ld*nb doubles)
              all-reduce of W = A(0:i+1, .) V T        ((i+1)*nb doubles)
  per column: all-reduce of the partial y = A v        (<= n doubles)
  at the end: assembly of A and Q: every block column of A and every row block of Q is broadcast
              once by its owner (an all-gather: (N-1)/N of 2 * ld*n doubles arrive at each rank)
"""
import ctypes as C
import os
import sys

from . import lib

ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_long, C.c_long)
BROADCAST_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_int)


def owned_column_blocks(n, block, world, rank):
    """Column blocks (first, last_exclusive) of an n-column matrix owned by `rank`:
    block b = c // block belongs to rank b % world (the rule of hessenberg_sharded_device)."""
    return [(b * block, min(n, (b + 1) * block))
            for b in range((n + block - 1) // block) if b % world == rank]


def owned_q_rows(n, world, rank):
    """Contiguous row block of Q owned by `rank` (multiples of 128 rows)."""
    chunk = ((n + world - 1) // world + 127) // 128 * 128
    return min(n, rank * chunk), min(n, (rank + 1) * chunk)


_native = {"state": None}       # None: not tried; False: unavailable; (rank, world): communicator ready


def native_rccl(group=None):
    """True when the library's own RCCL communicator is ready for the default process group."""
    import torch
    import torch.distributed as dist
    if group is not None or os.environ.get("SN_NATIVE_RCCL", "1") == "0":
        return False
    rank, world = dist.get_rank(), dist.get_world_size()
    if _native["state"] is not None:
        return _native["state"] == (rank, world)
    _native["state"] = False
    try:
        if dist.get_backend() != "nccl":
            return False
        L = lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())
        ident = torch.zeros(128, dtype=torch.uint8, device=dev)
        # every rank opens RCCL and makes an id (only rank 0's is used): ncclCommInitRank is a collective,
        # so no rank may enter it unless all of them can
        host = (C.c_ubyte * 128)()
        loaded = L.starneig_amd_rccl_unique_id(host) == 0
        if rank == 0 and loaded:
            ident.copy_(torch.tensor(list(host), dtype=torch.uint8))
        ok = torch.tensor([1 if loaded else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            return False
        dist.broadcast(ident, src=0)
        raw = bytes(ident.cpu().tolist())
        good = L.starneig_amd_rccl_init(rank, world, raw) == 0
        if good:
            # the new communicator against known answers before it carries a reduction
            stream = torch.cuda.current_stream().cuda_stream
            t = torch.full((64,), float(rank + 1), dtype=torch.float64, device=dev)
            good = L.starneig_amd_rccl_allreduce_sum(t.data_ptr(), 64, stream) == 0
            b = torch.full((64,), float(rank), dtype=torch.float64, device=dev)
            good = good and L.starneig_amd_rccl_broadcast(b.data_ptr(), 64, world - 1, stream) == 0
            torch.cuda.synchronize()
            good = good and bool((t == world * (world + 1) / 2.0).all()) and bool((b == float(world - 1)).all())
        agree = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 1:
            _native["state"] = (rank, world)
            return True
        L.starneig_amd_rccl_finalize()
    except Exception as e:          # no RCCL, an old library: the callback path still works
        sys.stderr.write(f"[starneig-amd] native RCCL not used: {e!r}\n")
    return False


def shutdown():
    """Destroys the library's RCCL communicator (call before torch.distributed.destroy_process_group)."""
    if _native["state"]:
        lib.load().starneig_amd_rccl_finalize()
    _native["state"] = None


class Collectives:
    """Maps the library's (buffer id, offset, count) requests onto torch.distributed calls."""

    def __init__(self, buffers, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.buffers = buffers          # id -> flat 1-D tensor
        self.calls = {"allreduce": 0, "broadcast": 0, "bytes": 0}
        self.allreduce_cb = ALLREDUCE_FN(self._allreduce)
        self.broadcast_cb = BROADCAST_FN(self._broadcast)

    def _global_rank(self, group_rank):
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)

    def _allreduce(self, ctx, buf, off, cnt):
        try:
            t = self.buffers[buf][off:off + cnt]
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self.calls["allreduce"] += 1
            self.calls["bytes"] += 8 * cnt
        except BaseException as e:      # an exception must not unwind through C: a rank that
            sys.stderr.write(f"[starneig-amd] collective failed: {e!r}\n")   # drops out would
            sys.stderr.flush()                                               # hang the others
            os._exit(3)

    def _broadcast(self, ctx, buf, off, cnt, root):
        try:
            t = self.buffers[buf][off:off + cnt]
            self.dist.broadcast(t, src=self._global_rank(root), group=self.group)
            self.calls["broadcast"] += 1
            self.calls["bytes"] += 8 * cnt
        except BaseException as e:
            sys.stderr.write(f"[starneig-amd] collective failed: {e!r}\n")
            sys.stderr.flush()
            os._exit(3)


def hessenberg_sharded(tA, tQ, n=None, panel_width=-1, group=None, sample_every=0):
    """Reduces the matrix held (identically) by every rank in tA; on return every rank holds
    the full Hessenberg form in tA and the full Q in tQ.  Returns (rc, stats).
    sample_every = k > 0: every k-th gemv launch of this rank's shard (with the all-reduce behind it)
    and every per-panel collective is timed with HIP events on the reduction's stream (bench.py)."""
    import torch
    import torch.distributed as dist
    L = lib.load()
    n = tA.shape[0] if n is None else n
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    pw = panel_width if panel_width > 0 else L.starneig_amd_default_panel_width(n)
    ld = L.starneig_amd_hessenberg_panel_ld(n, pw)
    dev = tA.device
    tY = torch.zeros(ld, dtype=torch.float64, device=dev)
    tP = torch.zeros(ld * pw, dtype=torch.float64, device=dev)
    tW = torch.zeros(n * pw, dtype=torch.float64, device=dev)
    coll = Collectives({0: tY, 1: tP, 2: tW, 3: tA.view(-1),
                        4: tQ.view(-1) if tQ is not None else None}, group)
    native = native_rccl(group)
    st = (C.c_double * 32)()
    st[7] = float(max(0, sample_every))
    rc = L.starneig_amd_hessenberg_sharded_device(
        n, pw, tA.data_ptr(), tA.shape[1], tQ.data_ptr() if tQ is not None else None,
        tQ.shape[1] if tQ is not None else 0, tY.data_ptr(), tP.data_ptr(), tW.data_ptr(),
        tW.numel(), rank, world,
        None if native else C.cast(coll.allreduce_cb, C.c_void_p),
        None if native else C.cast(coll.broadcast_cb, C.c_void_p),
        None, torch.cuda.current_stream().cuda_stream, st)
    stats = {"total_ms": st[0], "gemv_bytes": st[1], "gemm_flops": st[2],
             "gemv_launches": int(st[5]), "collectives": "RCCL, called from the library" if native else dict(coll.calls)}
    if sample_every > 0:
        kinds = ("allreduce_y", "broadcast_panel", "allreduce_w", "assembly")
        stats.update({"gemv_sampled_ms": st[8], "gemv_sampled_bytes": st[9], "gemv_sampled_launches": int(st[10]),
                      "allreduce_y_calls": int(st[11]), "rccl_ranks": int(st[24]) if native else 0,
                      "comm": {k: {"ms": st[12 + 3 * i], "bytes": st[13 + 3 * i], "calls": int(st[14 + 3 * i])}
                               for i, k in enumerate(kinds)}})
    return rc, stats


def owned_h_columns(n, world, device=None):
    """Columns of the Schur form each rank is the owner of after the sharded reduction: the
    128-column tiles T with T % world == rank (starneig_amd_schur_sharded_device)."""
    import torch
    col = torch.arange(n, device=device)
    return [col[(col // 128) % world == k] for k in range(world)]


def assemble_h_tiles(tH, n, group=None):
    """Every rank's tH (tH[c] = column c of H) gets the columns of the other ranks' tiles from
    their owners: one all-gather of the packed tiles, each byte travels once."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    cols = owned_h_columns(n, world, tH.device)
    per = max(len(c) for c in cols)
    send = torch.zeros((per, tH.shape[1]), dtype=tH.dtype, device=tH.device)
    send[:len(cols[rank])] = tH[cols[rank]]
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    for k in range(world):
        if k != rank and len(cols[k]):
            tH[cols[k]] = recv[k][:len(cols[k])]


def schur_sharded(tH, tQ, n=None, conf=None, group=None):
    """Schur reduction of the Hessenberg matrix every rank holds (identically) in tH: each rank
    reduces its replica of H -- the reduction is deterministic, the replicas stay bit-identical,
    no communication -- but accumulates only its row block of Q (half of the update flops of
    the leg) and updates only its own 128-column tiles of the deflated part of H; Q and H are assembled
    at the end by one all-gather each.
    Returns (rc, real, imag, stats)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    L = lib.load()
    n = tH.shape[0] if n is None else n
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    r0, r1 = owned_q_rows(n, world, rank)
    if world > 1:
        # the size of the AED window follows the `cores` of starneig_node_init: replicas only stay
        # bit-identical if every rank was initialised with the same value
        cores = torch.tensor([float(L.starneig_node_get_cores())], dtype=torch.float64, device=tH.device)
        lo, hi = cores.clone(), cores.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if lo.item() != hi.item():
            raise RuntimeError("starneig_amd: the ranks of a sharded Schur reduction must pass the same "
                               f"`cores` to starneig_node_init (got {int(lo.item())} .. {int(hi.item())})")
    real, imag = np.zeros(n), np.zeros(n)
    st = (C.c_double * 8)()
    rc = L.starneig_amd_schur_sharded_device(
        n, tH.data_ptr(), tH.shape[1], tQ.data_ptr() + 8 * r0, tQ.shape[1], r1 - r0, rank, world,
        real.ctypes.data, imag.ctypes.data, C.byref(conf) if conf is not None else None,
        torch.cuda.current_stream().cuda_stream, st)
    # tQ has shape (columns, ld): dimension 1 runs over the rows of the column-major matrix
    if world > 1:
        # the row blocks of Q only fit together if every replica of H was reduced identically
        # (the reduction is deterministic: ordered norm, no atomics on the Schur path): compare a
        # checksum of the eigenvalues and of diag(H) across the ranks before assembling Q
        dg = torch.diagonal(tH[:, :n])
        chk = torch.stack([dg.sum(), (dg * dg).sum(),
                           torch.tensor(float(real.sum()), dtype=torch.float64, device=tH.device),
                           torch.tensor(float(np.abs(imag).sum()), dtype=torch.float64, device=tH.device),
                           torch.tensor(float(rc), dtype=torch.float64, device=tH.device)])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if not torch.equal(lo, hi):
            raise RuntimeError("starneig_amd: the replicas of H diverged in the sharded Schur leg "
                               f"(checksums {lo.tolist()} .. {hi.tolist()})")
        # all-gather of the row blocks (every row travels once, from its owner) -- not "zero the
        # rest and all-reduce the whole matrix", which moves twice the bytes through every rank
        chunk = owned_q_rows(n, world, 0)[1]
        send = torch.zeros((n, chunk), dtype=torch.float64, device=tQ.device)
        send[:, :r1 - r0] = tQ[:n, r0:r1]
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(recv, send, group=group)
        for k in range(world):
            a, b = owned_q_rows(n, world, k)
            if k != rank and b > a:
                tQ[:n, a:b] = recv[k][:, :b - a]
        del send, recv
        # the deflated column tiles of H were kept up to date by their owners only (tile T by rank
        # T % world, include/starneig_amd.h): one all-gather of the packed tiles assembles the Schur form
        assemble_h_tiles(tH, n, group)
    stats = {"total_ms": st[0], "sweeps": int(st[1]), "aeds": int(st[2]),
             "small_solves": int(st[3]), "chase_launches": int(st[4]), "gemm_flops": st[5],
             "aed_host_s": st[6], "gpu_wait_s": st[7], "q_rows": (r0, r1)}
    return rc, real, imag, stats
