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
row block of Q -- 45 % of the update flops -- and keeps only its own 128-column tiles of the
deflated part of H up to date; Q and H are assembled by one all-gather each.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F64_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix peak (spec)


def pmc_traffic_ratio():
    """HBM traffic / algorithmic bytes of the panel gemv from the committed PMC passes
    (profiles/r6_gemv_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if the file is missing."""
    for name in ("r6_gemv_pmc_traffic.json", "r5_gemv_pmc_traffic.json", "r4_gemv_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)["traffic_over_algorithmic"]
        except Exception:
            continue
    return None


def hess_flops(n):
    return 16.0 / 3.0 * n ** 3      # 10/3 n^3 (A) + 2 n^3 (Q)


def schur_flops(n):
    return 25.0 * n ** 3            # Golub & Van Loan convention


def qz_flops(n):
    return 66.0 * n ** 3            # Golub & Van Loan QZ with Q and Z


def qz_pencil(S, n, kind):
    """Device-resident Hessenberg-triangular pencil.  kind "lcg": the reference test driver's
    random pencil (test/common/init.c:122-175, seed 2019; BASELINE config 5 -- its triangular
    factor is very ill-conditioned and the reduction is almost entirely AED).  kind "wellcond":
    the same draws with R <- triu(R,1)/sqrt(n) + diag(1 + |r_ii|) (oracle.random_pencil_wellcond):
    finite, well separated eigenvalues, so the QZ *sweeps* carry the reduction."""
    import torch
    tH0, tR0 = S.device_matrix(n), S.device_matrix(n)
    assert S.lcg_pencil_device(tH0, tR0, n, seed=2019) == 0
    if kind == "wellcond":
        R = tR0[:, :n]                       # R[c, r] = R(r, c): the upper triangle is torch's lower one
        d = torch.diagonal(R).abs() + 1.0
        R.copy_(torch.tril(R, -1) / float(n) ** 0.5)
        torch.diagonal(R).copy_(d)
    return tH0, tR0


def run_qz(S, n, kind, steps, warmup):
    """Times `steps` generalized Schur (QZ) reductions of one device-resident pencil; returns the
    fields of a bench line (value by the 66 n^3 convention, executed GEMM flops beside it)."""
    import torch
    tH0, tR0 = qz_pencil(S, n, kind)
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    times, st = [], None
    for it in range(warmup + steps):
        tH, tR = tH0.clone(), tR0.clone()
        S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n)
        torch.cuda.synchronize()
        assert rc == 0, f"gep schur rc={rc}"
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    _, ca = S.check_pencil_device(tQ, tH, tZ, tH0, n=n)
    _, cb = S.check_pencil_device(tQ, tR, tZ, tR0, n=n)
    total = sum(times)
    out = {"pencil": kind, "n": n, "steps": steps, "seconds_per_step": total / steps,
           "value": steps * qz_flops(n) / total / 1e9, "unit": "GFLOP/s (66 n^3 convention)",
           "executed_gemm_tflop_per_step": st["gemm_flops"] / 1e12,
           "executed_tflops_per_s": st["gemm_flops"] / 1e12 / (total / steps),
           "residual_a_u": ca["residual_u"], "residual_b_u": cb["residual_u"],
           "orthogonality_q_u": ca["orthogonality_q_u"], "orthogonality_z_u": ca["orthogonality_z_u"],
           "below_subdiagonal_nonzeros": ca["below_subdiagonal"],
           "qz_sweeps": st["sweeps"], "aeds": st["aeds"], "aed_host_s": st["aed_host_s"],
           # SURVEY 8d for a latency-bound leg: no roofline, the share of the time on the dependent chain of host
           # window kernels (one AED window after the other; the GPU's sweeps and updates fill the rest)
           "bound": "latency: the host's chain of AED windows", "critical_path_frac": st["aed_host_s"] / (total / steps)}
    del tH0, tR0, tQ, tZ
    torch.cuda.empty_cache()
    return out


def run_gep_chain(S, n):
    """starneig_GEP_SM_Reduce's two steps on a GENERAL pencil (two LCG matrices), device resident: the
    Hessenberg-triangular reduction, then QZ of its output -- a pencil on which the QZ sweeps (not only
    the AED windows) carry the reduction.  One timed run after a small warm-up of the workspaces."""
    import torch
    out = None
    first = None
    for m in (1000, 1600, n, n):       # (warm-ups: the rotation path, the two-stage path from n = 1100, and the
                                       # size itself -- the first call at a size allocates the workspaces)
        tA, tB = S.device_matrix(m), S.device_matrix(m)
        S.lcg_fill_device(tA, m, m, seed=2019); S.lcg_fill_device(tB, m, m, seed=77)
        tA0, tB0 = tA.clone(), tB.clone()
        tQ, tZ = S.device_matrix(m), S.device_matrix(m)
        S.set_matrix_device(tQ, m, m, 0.0, 1.0); S.set_matrix_device(tZ, m, m, 0.0, 1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, st = S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=m)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        assert rc == 0, rc
        rc, ar, ai, be, st2 = S.gep_schur_device(tA, tB, tQ, tZ, n=m)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        assert rc == 0, rc
        _, ca = S.check_pencil_device(tQ, tA, tZ, tA0, n=m)
        _, cb = S.check_pencil_device(tQ, tB, tZ, tB0, n=m)
        out = {"pencil": "general (LCG seeds 2019 / 77): Hessenberg-triangular reduction, then QZ", "n": m,
               "hessenberg_triangular_s": t1 - t0, "qz_s": t2 - t1,
               "hessenberg_triangular_first_call_s": first if first is not None else t1 - t0,
               "hessenberg_triangular_path": "two-stage Householder" if st.get("two_stage") else "rotations",
               "ns_per_chain_rotation": None if st.get("two_stage") else st["rotation_ms"] * 1e6 / max(st["rotations"] / 2, 1),
               "qz_sweeps": st2["sweeps"], "aeds": st2["aeds"], "aed_host_s": st2["aed_host_s"],
               "qz_bound": "latency: the host's chain of AED windows", "qz_critical_path_frac": st2["aed_host_s"] / max(t2 - t1, 1e-9),
               "hessenberg_triangular_stage1_s": st.get("stage1_ms", 0.0) / 1e3,
               "qz_executed_gemm_tflop": st2["gemm_flops"] / 1e12,
               "residual_a_u": ca["residual_u"], "residual_b_u": cb["residual_u"],
               "orthogonality_q_u": ca["orthogonality_q_u"], "orthogonality_z_u": ca["orthogonality_z_u"],
               "below_subdiagonal_nonzeros": ca["below_subdiagonal"]}
        if m == n and first is None:
            first = t1 - t0
        del tA, tB, tA0, tB0, tQ, tZ
        torch.cuda.empty_cache()
    return out


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
    r = run_qz(S, n, args.pencil, args.steps, args.warmup)
    print(json.dumps({
        "metric": "GFLOP/s generalized Schur (QZ), n=12000 Hessenberg-triangular pencil, 1 MI355X",
        "value": r["value"], "unit": "GFLOP/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["seconds_per_step"] * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": dict(r, workload=f"QZ, n={n}, Q and Z accumulated, {args.pencil} pencil seed 2019 "
                                   f"(BASELINE config 5); value = 66 n^3 flop / time"),
    }), flush=True)
    S.node_finalize()


def ht_flops(n):
    # 4/3 n^3 (QR of B) + 2 n^3 (Q0^T A) + 2 n^3 (Q Q0) + 14 n^3 (Moler-Stewart rotations with Q and Z,
    # Golub & Van Loan): the convention for the reduction the reference delegates to LAPACK
    return (16.0 / 3.0 + 14.0) * n ** 3


def lapack_ht_seconds(n):
    """dgeqrf + dormqr + dgghd3 (what wrappers/lapack.c calls) and the unblocked dgghrd, on the host
    cores, through scipy's bundled OpenBLAS; returns (dgghd3 chain s, dgghrd chain s) or None."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from make_golden_ht import lapack_ht, lcg_fullpos_pair
        import numpy as np
        rng = np.random.default_rng(2019)
        A, B = rng.random((n, n)), rng.random((n, n))
        out = []
        for blocked in (True, False):
            t0 = time.perf_counter()
            lapack_ht(A, B, blocked=blocked)
            out.append(time.perf_counter() - t0)
        return out
    except Exception:
        return None


def two_stage_roofline(n, st, traffic=None):
    """Stage 2 of the two-stage path (csrc/ht_twostage.hip) streams A and B through the reflector applications
    of its n^2 / 128 steps: rows p:p1 of A (columns right of the bulge) and B (columns p:n) from the left; columns
    p:p1 of A (rows top:p1+64) and B (rows top:p1) from the right, top = (j // 64) * 64 + 1 -- the rows above take
    the opposite reflectors of a whole group of 64 sweeps later, as compact-WY blocks (round 6; one pass over
    rows 0:top of the 127 columns of a block per group and position).  16 bytes an entry (read + write).
    Algorithmic bytes of those kernels over the whole of stage 2 (reflector generation, 75 us of latency a
    wavefront, included in the time) against the HBM peak.  traffic: PMC bytes of the same kernels when a counter
    pass is at hand (profiles/), else None."""
    import numpy as np
    r, gs = 64, 64
    j = np.arange(n - 2, dtype=np.int64)[:, None]
    t = np.arange((n - 3) // r + 1, dtype=np.int64)[None, :]
    p = j + 1 + r * t
    live = p <= n - 2
    p1 = np.minimum(p + r, n)
    ln = p1 - p
    c0 = np.where(t == 0, j, p - r)
    top = (j // gs) * gs + 1
    left = ln * ((n - c0 - 1) + (n - p))
    right = ln * ((np.minimum(p1 + r, n) - top) + (p1 - top))
    chase = 16.0 * float(((left + right) * live).sum())
    # the deferred rows: per group g and position t one pass over rows 0:top of <= 127 columns of A and B
    g = np.arange((n - 2 + gs - 1) // gs, dtype=np.int64)[:, None]
    col0 = g * gs + 1 + r * t
    k = np.minimum(gs, n - 1 - col0)
    m = np.minimum(k - 1 + r, n - col0)
    later = 16.0 * float((2 * (g * gs + 1) * m * (k > 0)).sum())
    nbytes = chase + later
    stage2_s = (st["rotation_ms"] - st["stage1_ms"]) / 1e3
    return {"bound": "hbm", "achieved": nbytes / stage2_s / 1e9, "peak": 8000.0, "unit": "GB/s",
            "frac": nbytes / stage2_s / 1e9 / 8000.0, "traffic": traffic,
            "kernel": "the three launches of a stage-2 wavefront -- ht2_m2_kernel (left application beside the second "
                      "half of the generation), ht2_near_kernel, ht2_m1_kernel (far part of the right application "
                      "beside the first half of the next generation) -- + the deferred rows in ht2_wy_right_kernel "
                      "(%.2f s, %.1f TB algorithmic of which %.2f TB deferred; rounds 1-5 applied everything at once: "
                      "%.1f TB); the 512-byte column pieces of the left application start on no 128-byte boundary: "
                      "a stand-alone copy of the pattern moves 3.6 TB/s where an in-place stream moves 4.7 "
                      "(profiles/r6_ht2_apply_patterns.txt)"
                      % (stage2_s, nbytes / 1e12, later / 1e12,
                         16.0 * float(((left + ln * (np.minimum(p1 + r, n) + p1)) * live).sum()) / 1e12)}


def bench_ht(args):
    """Hessenberg-triangular reduction (the step before BASELINE config 5): general
    pencil from the reference test driver's generator (two LCG matrices), n = 12000 by default,
    Q = Z = I, device resident.  Not the headline metric; one JSON line of the same shape."""
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    import starneig_amd as S
    S.node_init(1, 1, S.NO_MESSAGES)
    n = args.n if args.n != 20000 else 12000
    tA0, tB0 = S.device_matrix(n), S.device_matrix(n)
    S.lcg_fill_device(tA0, n, n, seed=2019); S.lcg_fill_device(tB0, n, n, seed=77)
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    times, st = [], None
    for it in range(args.warmup + args.steps):
        tA, tB = tA0.clone(), tB0.clone()
        S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc, st = S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
        torch.cuda.synchronize()
        assert rc == 0
        if it >= args.warmup:
            times.append(time.perf_counter() - t0)
    _, ca = S.check_pencil_device(tQ, tA, tZ, tA0, n=n)
    _, cb = S.check_pencil_device(tQ, tB, tZ, tB0, n=n)
    total = sum(times)
    cpu = lapack_ht_seconds(args.cpu_ht_n) if args.cpu_ht_n > 0 else None
    chain_steps = st["rotations"] / 2            # (0 on the two-stage path, the default from n = 2500: no rotation chain)
    two_stage = bool(st.get("two_stage"))
    print(json.dumps({
        "metric": "GFLOP/s Hessenberg-triangular reduction, n=12000 general pencil, 1 MI355X",
        "value": args.steps * ht_flops(n) / total / 1e9, "unit": "GFLOP/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": total / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"Hessenberg-triangular reduction, n={n}, Q and Z accumulated, LCG pencil "
                               f"(seeds 2019 / 77); value = (16/3 + 14) n^3 flop / time",
                   "n": n, "residual_a_u": ca["residual_u"], "residual_b_u": cb["residual_u"],
                   "orthogonality_q_u": ca["orthogonality_q_u"], "orthogonality_z_u": ca["orthogonality_z_u"],
                   "below_subdiagonal_nonzeros": ca["below_subdiagonal"],
                   "qr_step_s": st["qr_ms"] / 1e3, "rotation_step_s": st["rotation_ms"] / 1e3,
                   "path": "two-stage Householder (stage 1 %.2f s, stage 2 %.2f s)" % (st["stage1_ms"] / 1e3, (st["rotation_ms"] - st["stage1_ms"]) / 1e3)
                           if two_stage else "rotations (Moler-Stewart order, LDS-resident chain)",
                   "ns_per_chain_rotation": None if two_stage else st["rotation_ms"] * 1e6 / max(chain_steps, 1)},
        # the rotation path is bound by the dependent chain of n^2/2 column rotations, not by a roofline; stage 2 of
        # the two-stage path by the HBM traffic of its reflector applications:
        "roofline": two_stage_roofline(n, st) if two_stage else {
                     "bound": "latency", "achieved": st["rotation_ms"] * 1e6 / max(chain_steps, 1),
                     "peak": 78.0, "unit": "ns per dependent rotation (peak = the bare arithmetic of one "
                     "rotation on one wave, scratch/ht_micro.hip)", "frac": 78.0 / (st["rotation_ms"] * 1e6 / max(chain_steps, 1)),
                     "traffic": None},
        "cpu_baseline": None if cpu is None else {
            "value": ht_flops(args.cpu_ht_n) / cpu[0] / 1e9, "unit": "GFLOP/s", "cores": os.cpu_count(),
            "kind": "reference", "sample": f"LAPACK dgeqrf + dormqr + dgghd3 (the calls of wrappers/lapack.c) at "
            f"n={args.cpu_ht_n}: {cpu[0]:.1f} s; with the unblocked dgghrd: {cpu[1]:.1f} s (scipy's OpenBLAS, all cores)"},
    }), flush=True)
    S.node_finalize()


def cpu_baseline(n_lapack, n_port):
    """CPU baseline on this host's cores, on bounded samples of the same workload (smaller n, same LCG
    input, same flop conventions).  The reference's own StarPU build cannot be compiled in this image
    (DESIGN.md section 5), so there is no `kind: "reference"` number; what runs instead:
      * top level -- the STRONGEST CPU number the box offers, `kind: "lapack"`: LAPACK dgehrd + dorghr +
        dhseqr("S","V") through scipy's OpenBLAS on all the threads it uses: the comparator the reference's
        own test driver ships (`--solver lapack`, test/hessenberg/solvers.c:231-283,
        test/schur/solvers.c:120-169), the arithmetic the reference's sequential kernels call
        (schur/cpu_utils.c:2292) and -- dhseqr being a small-bulge multishift QR with aggressive early
        deflation -- the reference's algorithm class with multi-threaded BLAS-3 updates;
      * `port`: the repo's multi-threaded CPU restatement of the reference algorithm
        (oracle/hessenberg_oracle.c: the reference's panel / column / update order in plain loops under
        OpenMP; oracle/msqr_port.c: multishift QR with AED -- chains of packed bulges through diagonal
        windows, off-diagonal updates as threaded matrix products, window kernels from the product's
        host-only code).  A failure inside the port only drops this entry.
    Both are reported baselines, not targets; the reference's own published run is quoted in `sample`."""
    import numpy as np
    published = ("the reference itself, published (docs/_7_test_driver.md, BASELINE.md section 1): n=4000 on 6 workers "
                 "Hessenberg 13.1 s + Schur 9.5 s = 86 GFLOP/s by the same conventions")
    out = {"unit": "GFLOP/s", "kind": "lapack", "value": None, "cores": None, "sample": published}
    cores = os.cpu_count() or 1
    if n_lapack > 0:
        try:
            out.update(lapack_sample(n_lapack, cores))
            out["sample"] += "; " + published
        except Exception as e:              # the bench line must not die behind the GPU measurement
            out["lapack_error"] = repr(e)[:300]
    if n_port > 0:
        try:
            out["port"] = port_sample(n_port, cores)
        except Exception as e:
            out["port"] = {"error": repr(e)[:300]}
    if out["value"] is None and isinstance(out.get("port"), dict) and "value" in out["port"]:
        port = out.pop("port")
        out.update(port); out["kind"] = "port"; out["sample"] += "; " + published
    return out


def cpu_baseline_in_child(n_lapack):
    """The LAPACK sample in a child process with a clean threading environment: a launcher such as
    torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks, OpenBLAS sizes its thread pool from that when it
    is loaded, and raising the count afterwards (threadpoolctl) crashes scipy's OpenBLAS 0.3.29 -- found by
    running the N > 1 line, round 5.  The child never touches the GPU."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "GOTO_NUM_THREADS")}
    try:
        proc = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "cpu", "--cpu-n", str(n_lapack),
                               "--cpu-port-n", "0"], capture_output=True, text=True, timeout=600, env=env)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        if proc.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"unit": "GFLOP/s", "kind": "lapack", "value": None, "error": f"child rc {proc.returncode}", "stderr_tail": proc.stderr[-400:]}
    except Exception as e:
        return {"unit": "GFLOP/s", "kind": "lapack", "value": None, "error": repr(e)[:300]}


def port_sample(n_port, cores):
    import oracle as O
    import starneig_amd as S
    # (plain-loop kernels on 128-row windows: more threads than this only add overhead; set through the
    # OpenMP runtime, not the environment -- the runtime read that when torch loaded it)
    nthr = O.set_threads(min(cores, 32))
    hooks = S.lib.load_test_hooks()         # host-only window kernels of the product (csrc/schur_host.hip)
    A0 = O.random_fullpos(n_port)
    A = A0.copy(order="F")
    Q = O.identity(n_port)
    t0 = time.perf_counter()
    O.hessenberg(A, Q)
    t1 = time.perf_counter()
    rc, wr, wi, st = O.msqr_port(A, Q, hooks.sn_internal_aed_window, hooks.sn_internal_small_schur)
    t2 = time.perf_counter()
    if rc != 0 or O.check_schur_form(A) != 0:
        raise RuntimeError("the CPU port did not produce a Schur form")
    res = O.residual_u(Q, A, A0)
    if not res < 500.0:
        raise RuntimeError(f"CPU port residual {res} u")
    return {"value": (hess_flops(n_port) + schur_flops(n_port)) / (t2 - t0) / 1e9, "unit": "GFLOP/s",
            "cores": nthr, "kind": "port",
            "sample": f"CPU restatement of the reference algorithm on the LCG matrix at n={n_port}, "
                      f"{nthr} OpenMP threads: Hessenberg {t1 - t0:.1f} s "
                      f"(oracle/hessenberg_oracle.c: the reference's panel / column / update order), Schur "
                      f"{t2 - t1:.1f} s (oracle/msqr_port.c: multishift QR with AED, {st['sweeps']} sweeps, "
                      f"{st['aeds']} AED windows, one chain at a time -- the reference overlaps several; "
                      f"the window kernels are the product's own host code); residual {res:.0f} u; flop "
                      f"conventions (16/3+25) n^3"}


def lapack_sample(n, cores):
    import numpy as np
    import scipy.linalg as sl
    from scipy.linalg import lapack
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] + [1])
    except Exception:
        threads = cores
    import oracle as O
    A = np.asfortranarray(O.random_fullpos(n)[:n])
    lw1 = int(lapack.dgehrd_lwork(n)[0])        # (the default lwork selects the unblocked code)
    lw2 = int(lapack.dorghr_lwork(n)[0])
    t0 = time.perf_counter()
    ht, tau, info = lapack.dgehrd(A, lwork=lw1, overwrite_a=1)
    if info != 0:
        raise RuntimeError(f"dgehrd info {info}")
    H = np.asfortranarray(np.triu(ht, -1))
    Q, info = lapack.dorghr(ht, tau, lwork=lw2, overwrite_a=1)
    if info != 0:
        raise RuntimeError(f"dorghr info {info}")
    Q = np.asfortranarray(Q)
    t1 = time.perf_counter()
    if not lapack_dhseqr(H, Q):     # library symbol not found: real Schur form through dgees
        sl.schur(H, output="real")
    t2 = time.perf_counter()
    # sanity: the comparator really reduced the matrix (cheap check on a few columns)
    T = np.triu(H, -1)
    cols = np.arange(0, n, max(1, n // 16))
    A0 = O.random_fullpos(n)[:n]
    err = np.linalg.norm(Q @ (T @ Q.T[:, cols]) - A0[:, cols]) / np.linalg.norm(A0[:, cols])
    if not err < 1e-10:
        raise RuntimeError(f"LAPACK comparator residual {err}")
    flops = hess_flops(n) + schur_flops(n)
    return {"value": flops / (t2 - t0) / 1e9, "unit": "GFLOP/s", "cores": int(threads), "kind": "lapack",
            "sample": f"LAPACK (scipy OpenBLAS, {threads} threads of {cores} logical CPUs) on the LCG matrix "
                      f"at n={n}: dgehrd+dorghr {t1 - t0:.1f} s, dhseqr {t2 - t1:.1f} s; flop conventions "
                      f"(16/3+25) n^3; at this rate n=20000 would take {(t2 - t0) * (20000.0 / n) ** 3:.0f} s"}


def lapack_dhseqr(H, Z):
    """LAPACK dhseqr("S", "V") in place on Fortran-ordered H, Z -- scipy wraps no dhseqr, so
    the routine is called in scipy's bundled OpenBLAS (LP64, symbols prefixed scipy_)."""
    import ctypes as C
    import glob
    import numpy as np
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    if not libs:
        return False
    try:
        f = C.CDLL(libs[0]).scipy_dhseqr_
    except (OSError, AttributeError):
        return False
    n = H.shape[0]
    wr, wi = np.zeros(n), np.zeros(n)
    ci, vp = C.c_int, C.c_void_p

    def call(work, lwork):
        info = ci(0)
        f(C.c_char_p(b"S"), C.c_char_p(b"V"), C.byref(ci(n)), C.byref(ci(1)), C.byref(ci(n)),
          H.ctypes.data_as(vp), C.byref(ci(H.shape[0])), wr.ctypes.data_as(vp), wi.ctypes.data_as(vp),
          Z.ctypes.data_as(vp), C.byref(ci(Z.shape[0])), work.ctypes.data_as(vp), C.byref(ci(lwork)),
          C.byref(info), C.c_size_t(1), C.c_size_t(1))
        return info.value
    q = np.zeros(1)
    assert call(q, -1) == 0
    work = np.zeros(int(q[0]) + 1)
    assert call(work, work.size) == 0
    return True


def host_api_call(S, n):
    """ONE call of starneig_SEP_SM_Hessenberg + starneig_SEP_SM_Schur with HOST arrays (what the
    reference's caller sees: H2D + compute + D2H, test/common/hook_experiment.c:1817-1825).
    Pageable numpy arrays, ld = n rounded up to 8 (test/common/common.c:99).  Never `value`."""
    import numpy as np
    ld = (n + 7) // 8 * 8
    A = np.zeros((ld, n), order="F")
    # the LCG matrix, generated on the device and copied out (bit-identical to the oracle's)
    import torch
    t = S.device_matrix(n, ld=ld)
    S.lcg_fill_device(t, n, n, seed=2019, mode=0)
    A[:, :] = t.cpu().numpy().T
    del t
    torch.cuda.empty_cache()
    Q = np.zeros((ld, n), order="F")
    Q[np.arange(n), np.arange(n)] = 1.0
    real, imag = np.zeros(n), np.zeros(n)
    t0 = time.perf_counter()
    rc = S.SEP_SM_Hessenberg(n, A, ld, Q, Q.shape[0])
    t1 = time.perf_counter()
    assert rc == 0, rc
    rc = S.SEP_SM_Schur(n, A, ld, Q, Q.shape[0], real, imag)
    t2 = time.perf_counter()
    assert rc == 0, rc
    return {"hessenberg_s": t1 - t0, "schur_s": t2 - t1, "total_s": t2 - t0,
            "gflops": (hess_flops(n) + schur_flops(n)) / (t2 - t0) / 1e9}


def secondary_in_child():
    """The secondary workloads run in a child process of their own (started here, relayed, never exec'ed):
    the stream set-up the Hessenberg-triangular reduction is tuned for is the one of a fresh process -- after
    the Schur and QZ legs have created their ~20 priority streams the same reduction runs at half the rate
    (the runtime multiplexes streams onto a few hardware queues).  The caller starts it before it creates its own
    HIP context, for the same reason."""
    import subprocess
    proc = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "secondary"],
                          capture_output=True, text=True, timeout=900)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("[")]
    if proc.returncode != 0 or not lines:
        return {"error": f"secondary child failed (rc {proc.returncode})", "stderr_tail": proc.stderr[-600:]}
    return json.loads(lines[-1])


def bench_secondary(args):
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    import starneig_amd as S
    S.node_init(S.USE_ALL, 1, S.NO_MESSAGES)
    out = [run_gep_chain(S, 8000), run_qz(S, 12000, "lcg", 1, 1), run_qz(S, 12000, "wellcond", 1, 1)]
    S.node_finalize()
    print(json.dumps(out), flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as a CHILD
    process (torch.distributed.run, one rank per GPU) before anything here touches the GPU, relay
    its output and exit with its code."""
    import subprocess
    port = os.environ.get("MASTER_PORT", "29531")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1,
                    help="untimed steps (the first call allocates the cached workspaces and creates ~10^4 events)")
    ap.add_argument("--size", "--n", dest="n", type=int, default=20000)
    ap.add_argument("--cpu-n", type=int, default=4000,
                    help="size of the LAPACK CPU-baseline sample (0 = skip; n=4000 is ~30 s on the GPU box's host, "
                         "n=8000 ~4 min)")
    ap.add_argument("--cpu-port-n", type=int, default=3000,
                    help="size of the oracle-port CPU sample (0 = skip)")
    ap.add_argument("--host-api", type=int, default=1,
                    help="1: also time ONE call of the host-array API (PCIe-inclusive), N=1 only")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the sharded Hessenberg path even at N=1 (exercises the collectives)")
    ap.add_argument("--workload", choices=["sep", "qz", "ht", "secondary", "cpu"], default="sep",
                    help="sep = Hessenberg + Schur (the headline metric); qz = BASELINE config 5; "
                         "ht = Hessenberg-triangular reduction (the step before config 5)")
    ap.add_argument("--cpu-ht-n", type=int, default=1500,
                    help="size of the LAPACK sample of the ht workload (0 = skip)")
    ap.add_argument("--pencil", choices=["lcg", "wellcond"], default="lcg",
                    help="qz workload: the test driver's pencil (config 5) or its well-conditioned variant")
    ap.add_argument("--secondary", type=int, default=1,
                    help="1: append the QZ legs (config 5 and the well-conditioned pencil, n=12000, one timed "
                         "step each) and the generalized chain (Hessenberg-triangular + QZ, n=8000) to the "
                         "default line as `secondary`, N=1 only")
    ap.add_argument("--sample-every", type=int, default=16,
                    help="time every k-th panel-gemv launch with HIP events")
    args = ap.parse_args()
    if args.workload == "qz":
        return bench_qz(args)
    if args.workload == "ht":
        return bench_ht(args)
    if args.workload == "secondary":
        return bench_secondary(args)
    if args.workload == "cpu":      # the CPU baseline alone (no GPU): what cpu_baseline_in_child runs
        print(json.dumps(cpu_baseline(args.cpu_n, args.cpu_port_n)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} rank(s)"
    sharded = world > 1 or args.force_sharded
    # The secondary workloads (not part of `value`) run in a child process BEFORE this process touches the GPU: the
    # streams a live process holds are hardware queues the child's streams are multiplexed with -- the launch-bound
    # Hessenberg-triangular reduction measured 6.05 s at n = 8000 beside this process's idle streams, 5.45 s alone.
    secondary = secondary_in_child() if (world == 1 and args.secondary and not sharded) else None

    import torch
    import torch.distributed as dist
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
    S.node_init(S.USE_ALL, 1, S.NO_MESSAGES)     # cores: the staging threads of the host-array API use up to 8

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
            rc, st = D.hessenberg_sharded(tA, tQ, n=n, sample_every=sample_every)
            for k in ("gemv_sampled_ms", "gemv_sampled_bytes", "gemv_sampled_launches"):
                st.setdefault(k, 0)
            st.update({"gemm_main_ms": 0.0, "gemm_main_flops": 0.0, "gemm_side_ms": 0.0,
                       "gemm_fused_ms": 0.0, "gemm_fused_flops": 0.0})
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

    # N > 1: every rank sampled ITS shard of the gemv (1/N of the columns) and the collectives on its own
    # stream; the per-rank figures travel to rank 0 in one all-gather
    per_rank = None
    if sharded:
        kinds = ("allreduce_y", "broadcast_panel", "allreduce_w", "assembly")
        mine = [sum(s_["gemv_sampled_ms"] for s_ in stats), sum(s_["gemv_sampled_bytes"] for s_ in stats),
                float(sum(s_["gemv_sampled_launches"] for s_ in stats))]
        for k in kinds:
            c = [s_.get("comm", {}).get(k, {"ms": 0.0, "bytes": 0.0, "calls": 0}) for s_ in stats]
            mine += [sum(x["ms"] for x in c), sum(x["bytes"] for x in c), float(sum(x["calls"] for x in c))]
        mine += [float(sum(s_.get("allreduce_y_calls", 0) for s_ in stats)), float(stats[-1].get("rccl_ranks", 0)),
                 sum(s_["hessenberg_s"] for s_ in stats), sum(s_["schur_s"] for s_ in stats)]
        tm_ = torch.tensor(mine, dtype=torch.float64, device="cuda")
        gathered = [torch.empty_like(tm_) for _ in range(world)]
        dist.all_gather(gathered, tm_)
        per_rank = [g.cpu().tolist() for g in gathered]

    out = None
    if rank == 0:
        ms_per_step = total / args.steps * 1e3
        value = args.steps * (hess_flops(n) + schur_flops(n)) / total / 1e9   # ONE job on all GPUs
        sm = sum(s["gemv_sampled_ms"] for s in stats)
        sb = sum(s["gemv_sampled_bytes"] for s in stats)
        nl = sum(s["gemv_sampled_launches"] for s in stats)
        achieved = sb / (sm * 1e-3) / 1e9 if sm > 0 else None
        peak = HBM_PEAK_GBS
        shard_note = None
        if per_rank is not None:
            # whole-job rate of the sharded kernel = the sum of the ranks' rates on their shards (they stream
            # at the same time), against world x the HBM peak
            rates = [(r[1] / (r[0] * 1e-3) / 1e9) if r[0] > 0 else 0.0 for r in per_rank]
            achieved = sum(rates) if all(x > 0 for x in rates) else None
            peak = HBM_PEAK_GBS * world
            sb = sum(r[1] for r in per_rank); nl = int(sum(r[2] for r in per_rank)); sm = sum(r[0] for r in per_rank)
            shard_note = {"per_rank_GBps": rates, "per_rank_avg_launch_us": [(r[0] / r[2] * 1e3) if r[2] else None for r in per_rank]}
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
                               f"all-reduce of y, panel broadcast); Schur: band of H replicated, accumulation of Q "
                               f"sharded by row block, left updates of the deflated columns of H by "
                               f"128-column tile (one all-gather each at the end)",
                "collectives": stats[-1].get("collectives"),
                "residual_u": chk["residual_u"], "orthogonality_u": chk["orthogonality_u"],
                "below_subdiagonal_nonzeros": chk["below_subdiagonal"],
                "hessenberg_s": stats[-1]["hessenberg_s"], "schur_s": stats[-1]["schur_s"],
                "hessenberg_gflops": hess_flops(n) / stats[-1]["hessenberg_s"] / 1e9,
                "schur_sweeps": stats[-1]["schur"]["sweeps"], "schur_aeds": stats[-1]["schur"]["aeds"],
                "executed_gemm_tflop_per_step":
                    (stats[-1]["gemm_flops"] + stats[-1]["schur"]["gemm_flops"]) / 1e12,
                # the executed GEMM work over the step time, beside the convention-based `value`
                "executed_tflops_per_s":
                    sum(s_["gemm_flops"] + s_["schur"]["gemm_flops"] for s_ in stats) / 1e12 / total,
            },
            "roofline": {
                "kernel": "hess_gemv_kernel (panel y = A v, rows H2 of SURVEY 8a)",
                "bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                "frac": (achieved / peak) if achieved else None,
                "traffic": (ratio * sb / nl) if (ratio and nl) else None,
                "traffic_note": "avg algorithmic bytes per launch x PMC ratio "
                                "(2*FETCH_SIZE+WRITE_SIZE)/algorithmic measured on the 624 launches of "
                                "the first two panels at n=20000, profiles/r6_gemv_pmc_traffic.json",
                "launches_timed": nl,
                "avg_launch_us": (sm / nl * 1e3) if nl else None,
                "avg_launch_bytes": (sb / nl) if nl else None,
            },
        }
        if shard_note is not None:
            out["roofline"]["kernel"] = ("hess_gemv_kernel<16,true,true,SHARD> (each rank streams its block columns of the "
                                         "panel y = A v and folds its column splits in the launch; rows H2 of SURVEY 8a)")
            out["roofline"]["sharded"] = shard_note
            out["roofline"]["peak_note"] = f"{world} x {HBM_PEAK_GBS:.0f} GB/s; achieved = sum of the ranks' rates on their shards"
            # SURVEY 8d, scaling report: bytes per panel and achieved GB/s per collective kind.  Payload bytes / event
            # time on the reduction's stream (slowest rank); bus factor of a ring all-reduce 2 (N-1) / N, of a broadcast 1
            npanels = max(1, -(-(n - 1) // S.default_panel_width(n)))
            comm = {}
            base = 3
            for i, k in enumerate(("allreduce_y", "broadcast_panel", "allreduce_w", "assembly")):
                ms = max(r[base + 3 * i] for r in per_rank); by = per_rank[0][base + 3 * i + 1]; calls = per_rank[0][base + 3 * i + 2]
                bus = 2.0 * (world - 1) / world if k.startswith("allreduce") else 1.0
                comm[k] = {"timed_calls_per_step": calls / args.steps, "payload_bytes_per_call": (by / calls) if calls else None,
                           "avg_us_per_call": (ms / calls * 1e3) if calls else None,
                           "payload_GBps": (by / (ms * 1e-3) / 1e9) if ms > 0 else None,
                           "bus_GBps": (bus * by / (ms * 1e-3) / 1e9) if ms > 0 else None}
            ary = per_rank[0][base + 12] / args.steps
            comm["allreduce_y"]["calls_per_step"] = ary
            comm["allreduce_y"]["projected_s_per_step"] = (comm["allreduce_y"]["avg_us_per_call"] or 0.0) * ary * 1e-6
            bytes_per_panel = sum((comm[k]["payload_bytes_per_call"] or 0.0) * (comm[k]["timed_calls_per_step"] or 0.0)
                                  for k in ("broadcast_panel", "allreduce_w")) / npanels
            bytes_per_panel += (comm["allreduce_y"]["payload_bytes_per_call"] or 0.0) * ary / npanels
            out["config"]["rccl_ranks"] = int(per_rank[0][base + 13])
            out["config"]["collective_transport"] = ("RCCL over xGMI, called from the library (ncclCommCount in rccl_ranks)"
                                                     if per_rank[0][base + 13] > 0 else "torch.distributed callbacks (not RCCL)")
            out["config"]["collectives_by_kind"] = comm
            out["config"]["collective_payload_bytes_per_panel"] = bytes_per_panel
            out["config"]["per_rank_hessenberg_s"] = [r[base + 14] / args.steps for r in per_rank]
            out["config"]["per_rank_schur_s"] = [r[base + 15] / args.steps for r in per_rank]
        # second roofline entry: the compact-WY trailing update (rows H4-H6, fused into one
        # k = 2 nb MFMA GEMM + the W product), executed flops / event-timed duration on the
        # critical stream, in situ (the delayed Q updates and the next panel run beside it)
        gm = sum(s["gemm_main_ms"] for s in stats)
        gf = sum(s["gemm_main_flops"] for s in stats)
        gs = sum(s["gemm_side_ms"] for s in stats)
        gfs = sum(s["gemm_flops"] - s["gemm_main_flops"] for s in stats)
        fm = sum(s["gemm_fused_ms"] for s in stats)
        ff = sum(s["gemm_fused_flops"] for s in stats)
        if gm > 0 and fm > 0:
            tf = ff / (fm * 1e-3) / 1e12
            tfc = gf / (gm * 1e-3) / 1e12
            out["roofline_mfma"] = {
                "kernel": "dgemm_kernel<128,128,16,N,T>: the fused trailing update A -= [Y V][V' W]^T, k = 2 nb (rows H4, H6)",
                "bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / MFMA_F64_PEAK_TFLOPS, "traffic": None,
                "flops_per_step": ff / args.steps, "ms_per_step": fm / args.steps,
                # the whole critical update: + the split-K W product (row H5), S = Y^T VT, W -= V' S, their memsets and gaps
                "critical_update_tflops": tfc, "critical_update_frac": tfc / MFMA_F64_PEAK_TFLOPS,
                "critical_update_ms_per_step": gm / args.steps,
                "side_stream_tflops": (gfs / (gs * 1e-3) / 1e12) if gs > 0 else None,
                "note": "in situ, HIP events on the critical stream around every panel's launch(es); the same kernel "
                        "alone with PMC MFMA-busy counters: profiles/r4_dgemm_mfma_utilisation.json, in situ over the first four panels: profiles/r6_pmc_mfma_in_situ.json (0.72 MFMA busy at the library's panel width, k = 384; 0.76 at the reference's, k = 624; the kernel is unchanged since round 4); why in situ reads "
                        "lower than alone (clock state after the HBM-bound panel): profiles/r4_gemm_phase_experiment.txt",
            }
        if world == 1 and args.host_api:
            out["config"]["host_api_s"] = host_api_call(S, n)
        if world == 1 and (args.cpu_n > 0 or args.cpu_port_n > 0):
            out["cpu_baseline"] = cpu_baseline(args.cpu_n, args.cpu_port_n)

        if world == 1 and args.secondary and not sharded:
            # driver-timed secondary workloads (not part of `value`): BASELINE config 5 and the same
            # size on a well-conditioned pencil, where the QZ sweeps -- not the host AED -- do the work
            out["secondary"] = secondary

    if sharded:
        from starneig_amd import distributed as _D
        torch.cuda.synchronize()
        _D.shutdown()               # the library's own RCCL communicator, before torch's
    S.node_finalize()
    if sharded:
        dist.destroy_process_group()
    if rank == 0 and world > 1 and args.cpu_n > 0:
        # N > 1: the same baseline on rank 0's host cores -- after the process group is gone (no rank waits
        # inside a collective for it), the LAPACK sample only and a size smaller; the N = 1 line carries
        # the port as well
        out["cpu_baseline"] = cpu_baseline_in_child(min(args.cpu_n, 3000))
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
