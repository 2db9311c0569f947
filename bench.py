#!/usr/bin/env python3
"""Benchmark of the MI355X Hessenberg(+Schur) hot path -- driver contract.

    python bench.py --gpus N --steps K --warmup W

One "step" = Hessenberg reduction + Schur reduction (Q accumulated through both) of one
synthetic n x n matrix (the reference test driver's LCG input, seed 2019, generated
directly in HBM; Q = I), inputs resident in HBM when the timed region starts.
Prints ONE JSON line (rank 0).  `--workload qz` measures BASELINE config 5 instead (QZ of
a 12000 x 12000 Hessenberg-triangular pencil); it is not the headline metric.

At N > 1 the N GPUs reduce ONE matrix together (strong scaling): the Hessenberg leg is
sharded by block column (per-column all-reduce of the partial y = A v, panel broadcast,
starneig_amd/distributed.py); in the Schur leg every rank reduces its replica of H (the
latency-bound chain of window steps and host AEDs does not shard) but accumulates only its
row block of Q -- 45 % of the update flops -- and Q is assembled by one all-reduce.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_traffic_ratio():
    """HBM traffic / algorithmic bytes of the panel gemv from the committed PMC passes
    (profiles/r1_gemv_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_gemv_pmc_traffic.json")) as f:
            return json.load(f)["traffic_over_algorithmic"]
    except Exception:
        return None


def hess_flops(n):
    return 16.0 / 3.0 * n ** 3      # 10/3 n^3 (A) + 2 n^3 (Q)


def schur_flops(n):
    return 25.0 * n ** 3            # Golub & Van Loan convention


def qz_flops(n):
    return 66.0 * n ** 3            # Golub & Van Loan QZ with Q and Z


def bench_qz(args):
    """BASELINE config 5 (not the headline metric): generalized Schur (QZ) reduction of the
    reference test driver's random Hessenberg-triangular pencil, n = 12000 by default, Q = Z = I,
    device resident.  Prints one JSON line of the same shape."""
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    import starneig_amd as S
    S.node_init(1, 1, S.NO_MESSAGES)
    n = args.n if args.n != 20000 else 12000
    tH0, tR0 = S.device_matrix(n), S.device_matrix(n)
    assert S.lcg_pencil_device(tH0, tR0, n, seed=2019) == 0
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    times, st = [], None
    for it in range(args.warmup + args.steps):
        tH, tR = tH0.clone(), tR0.clone()
        S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n)
        torch.cuda.synchronize()
        assert rc == 0, f"gep schur rc={rc}"
        if it >= args.warmup:
            times.append(time.perf_counter() - t0)
    _, ca = S.check_pencil_device(tQ, tH, tZ, tH0, n=n)
    _, cb = S.check_pencil_device(tQ, tR, tZ, tR0, n=n)
    total = sum(times)
    print(json.dumps({
        "metric": "GFLOP/s generalized Schur (QZ), n=12000 Hessenberg-triangular pencil, 1 MI355X",
        "value": args.steps * qz_flops(n) / total / 1e9, "unit": "GFLOP/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": total / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"QZ, n={n}, Q and Z accumulated, LCG pencil seed 2019 (BASELINE config 5); "
                               f"value = 66 n^3 flop / time",
                   "n": n, "residual_a_u": ca["residual_u"], "residual_b_u": cb["residual_u"],
                   "orthogonality_q_u": ca["orthogonality_q_u"], "orthogonality_z_u": ca["orthogonality_z_u"],
                   "below_subdiagonal_nonzeros": ca["below_subdiagonal"],
                   "qz_sweeps": st["sweeps"], "aeds": st["aeds"],
                   "executed_gemm_tflop_per_step": st["gemm_flops"] / 1e12,
                   "aed_host_s": st["aed_host_s"]},
    }), flush=True)
    S.node_finalize()


def cpu_baseline(n_sample):
    """The CPU oracle (kind "port": the restatement of the reference algorithm) timed on
    this host's cores on a bounded sample of the same workload (smaller n, same input
    generator, same default panel width rule).  The Hessenberg leg runs OpenMP over all
    cores; the Schur leg (double-shift QR restatement) is a scalar single-thread port."""
    import oracle as O
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(min(cores, 64)))
    A = O.random_fullpos(n_sample)
    Q = O.identity(n_sample)
    t0 = time.perf_counter()
    O.hessenberg(A, Q)
    t1 = time.perf_counter()
    O.schur(A, Q)
    t2 = time.perf_counter()
    flops = hess_flops(n_sample) + schur_flops(n_sample)
    return {
        "value": flops / (t2 - t0) / 1e9, "unit": "GFLOP/s",
        "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
        "sample": f"oracle Hessenberg+Schur of the LCG matrix at n={n_sample}: Hessenberg "
                  f"{t1 - t0:.1f} s (OpenMP, {os.environ['OMP_NUM_THREADS']} threads), Schur "
                  f"{t2 - t1:.1f} s (1 thread); same flop conventions (16/3+25) n^3",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1,
                    help="untimed steps (the first call allocates the cached workspaces and creates ~10^4 events)")
    ap.add_argument("--size", "--n", dest="n", type=int, default=20000)
    ap.add_argument("--cpu-n", type=int, default=1500, help="size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the sharded Hessenberg path even at N=1 (exercises the collectives)")
    ap.add_argument("--workload", choices=["sep", "qz"], default="sep",
                    help="sep = Hessenberg + Schur (the headline metric); qz = BASELINE config 5")
    ap.add_argument("--sample-every", type=int, default=16,
                    help="time every k-th panel-gemv launch with HIP events")
    args = ap.parse_args()
    if args.workload == "qz":
        return bench_qz(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        # SN_BENCH_BACKEND=gloo with SN_BENCH_ONE_GPU=1 runs the N > 1 code path with all ranks on
        # cuda:0 (testing on a single-GPU box); the real runs use RCCL, one GPU per rank
        backend = os.environ.get("SN_BENCH_BACKEND", "nccl")
        if os.environ.get("SN_BENCH_ONE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")

    import starneig_amd as S
    from starneig_amd import distributed as D
    S.node_init(1, 1, S.NO_MESSAGES)

    n = args.n
    tA0 = S.device_matrix(n)
    assert S.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0       # the same matrix on every rank
    tA = torch.empty_like(tA0)
    tQ = S.device_matrix(n)

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(sample_every):
        tA.copy_(tA0)
        S.set_matrix_device(tQ, n, n, 0.0, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if sharded:
            rc, st = D.hessenberg_sharded(tA, tQ, n=n)
            st.update({"gemv_sampled_ms": 0.0, "gemv_sampled_bytes": 0.0, "gemv_sampled_launches": 0})
        else:
            rc, st = S.hessenberg_device(tA, tQ, n=n, stats=True, sample_every=sample_every)
        torch.cuda.synchronize()
        assert rc == 0
        t1 = time.perf_counter()
        if sharded:
            rc, real, imag, sst = D.schur_sharded(tA, tQ, n=n)
        else:
            rc, real, imag, sst = S.schur_device(tA, tQ, n=n)
        torch.cuda.synchronize()
        assert rc == 0, f"schur rc={rc}"
        t2 = time.perf_counter()
        st["hessenberg_s"] = t1 - t0
        st["schur_s"] = t2 - t1
        st["schur"] = sst
        return t2 - t0, st

    for _ in range(args.warmup):
        one_step(0)

    barrier()
    step_s, stats = [], []
    for _ in range(args.steps):
        dt, st = one_step(args.sample_every)
        step_s.append(dt)
        stats.append(st)
    barrier()
    total = sum(step_s)         # input reset (copy + identity) is outside the timed region
    t = torch.tensor([total], dtype=torch.float64, device="cuda")
    if sharded:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    total = t.item()

    # correctness of the last step, on the GPU (reference acceptance checks)
    rc, chk = S.check_device(tQ, tA, tA0, n=n)
    assert rc == 0

    out = None
    if rank == 0:
        ms_per_step = total / args.steps * 1e3
        value = args.steps * (hess_flops(n) + schur_flops(n)) / total / 1e9   # ONE job on all GPUs
        sm = sum(s["gemv_sampled_ms"] for s in stats)
        sb = sum(s["gemv_sampled_bytes"] for s in stats)
        nl = sum(s["gemv_sampled_launches"] for s in stats)
        achieved = sb / (sm * 1e-3) / 1e9 if sm > 0 else None
        ratio = pmc_traffic_ratio()
        out = {
            "metric": "GFLOP/s Hessenberg+Schur, n=20000 real dense, 1/2/4/8 MI355X; residual",
            "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"Hessenberg + multi-shift QR Schur, n={n}, Q accumulated, LCG input "
                            f"seed 2019 (BASELINE config 3); value = (16/3 + 25) n^3 flop / time",
                "n": n, "panel_width": S.default_panel_width(n),
                "parallelism": "single GPU" if not sharded else
                               f"Hessenberg sharded by block column over {world} GPU(s) (RCCL: per-column "
                               f"all-reduce of y, panel broadcast); Schur: H replicated, accumulation of Q "
                               f"sharded by row block (one all-reduce at the end)",
                "collectives": stats[-1].get("collectives"),
                "residual_u": chk["residual_u"], "orthogonality_u": chk["orthogonality_u"],
                "below_subdiagonal_nonzeros": chk["below_subdiagonal"],
                "hessenberg_s": stats[-1]["hessenberg_s"], "schur_s": stats[-1]["schur_s"],
                "hessenberg_gflops": hess_flops(n) / stats[-1]["hessenberg_s"] / 1e9,
                "schur_sweeps": stats[-1]["schur"]["sweeps"], "schur_aeds": stats[-1]["schur"]["aeds"],
                "executed_gemm_tflop_per_step":
                    (stats[-1]["gemm_flops"] + stats[-1]["schur"]["gemm_flops"]) / 1e12,
            },
            "roofline": {
                "kernel": "hess_gemv_kernel (panel y = A v, rows H2 of SURVEY 8a)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": (ratio * sb / nl) if (ratio and nl) else None,
                "traffic_note": "avg algorithmic bytes per launch x PMC ratio "
                                "(2*FETCH_SIZE+WRITE_SIZE)/algorithmic measured on the 312 "
                                "first-panel launches at n=20000, profiles/r1_gemv_pmc_traffic.json",
                "launches_timed": nl,
                "avg_launch_us": (sm / nl * 1e3) if nl else None,
                "avg_launch_bytes": (sb / nl) if nl else None,
            },
        }
        if world == 1 and args.cpu_n > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n)

    S.node_finalize()
    if sharded:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
