"""The headline leg: BASELINE config 3, the 100k-pose SE(3) Lambda solve (numeric factor + two substitutions, inputs in HBM) -- its
algorithmic counts, per-kernel rooflines timed with HIP events, the CPU baseline (the compiled reference's CHOLMOD),
K value sets on one device (replicas), marginal covariances and the Lambda assembly beside it."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import _KEEP, _DevPtr  # noqa: F401

__all__ = ['algorithmic_counts_c3', 'per_kernel_bytes_sparse', 'cpu_baseline_c3', 'replicas_leg', 'marginals_leg_c3', 'assembly_leg', 'run_c3']


def algorithmic_counts_c3(lam, stats):
    if lam.n_bcols == 100_000 and lam.n_blocks == 201_998:
        c = dict(C3_REF)
        c["source"] = "reference CHOLMOD/AMD counters (fl, lnz) on this instance"
    else:  # other sizes: our own ordering's counts (upper bound on the reference's)
        c = {"n": lam.n_scalars, "nnz_triu": stats["nnz_upper"], "lnz": stats["l_nnz"],
             "fl": stats["factor_flops"], "source": "own ordering"}
    c["flops"] = c["fl"] + 4.0 * c["lnz"]
    c["factor_bytes"] = 8.0 * (c["nnz_triu"] + c["lnz"])
    c["solve_bytes"] = 16.0 * c["lnz"] + 32.0 * c["n"]
    return c
def per_kernel_bytes_sparse(plan, n_bottom_stages=1):
    """Algorithmic bytes each kernel of the sparse path moves per step, from the plan (our own
    factor structure): factor = 8 (nnz of the Lambda blocks read + nnz of the L columns written),
    substitution = 8 nnz(L columns) + vectors, split into the bottom-stage launch and the rest."""
    dim = plan["dim"].astype(np.int64)
    lptr, lrow, asrc = plan["lptr"], plan["lrow"].astype(np.int64), plan["asrc"]
    n = len(dim)
    col_of = np.repeat(np.arange(n), np.diff(lptr))
    is_diag = np.zeros(len(lrow), dtype=bool)
    is_diag[lptr[:-1]] = True
    dj, di = dim[col_of], dim[lrow]
    blk_nnz = np.where(is_diag, dj * (dj + 1) // 2, di * dj)
    l_col = np.bincount(col_of, weights=blk_nnz, minlength=n)
    a_col = np.bincount(col_of, weights=np.where(asrc >= 0, blk_nnz, 0), minlength=n)
    def cols_of(s0, s1):
        m = np.zeros(n, dtype=bool)
        t0, t1 = plan["stage_ptr"][s0], plan["stage_ptr"][s1]
        m[plan["task_cols"][plan["task_ptr"][t0]:plan["task_ptr"][t1]]] = True
        return m
    n_stages = len(plan["stage_ptr"]) - 1
    leaves = cols_of(0, 1)                                         # stage 0: the leaf subtrees (lane-per-task kernel)
    wide = cols_of(1, n_bottom_stages) if n_bottom_stages > 1 else np.zeros(n, dtype=bool)   # wave-per-task kernel
    upper = cols_of(n_bottom_stages, n_stages) if n_bottom_stages < n_stages else np.zeros(n, dtype=bool)
    fac = 8.0 * (l_col + a_col) + 16.0 * dim     # + the fused forward substitution's vector traffic
    sub = 8.0 * l_col + 16.0 * dim               # a substitution reads the L column, reads + writes the vector
    return {"factor_leaves": float(fac[leaves].sum()), "factor_wide": float(fac[wide].sum()),
            "factor_upper": float(fac[upper].sum()), "forward": float(sub.sum()), "backward": float(sub.sum()),
            "leaf_cols": int(leaves.sum())}
def cpu_baseline_c3(lam, counts, x_gpu, budget_reps=12):
    """The compiled reference on this box's host, on the same system: (a) what its nonlinear solver pays per iteration
    with CLinearSolver_CholMod -- Solve_PosDef, whose tag is "basic": conversion, ordering and symbolic analysis re-run on
    every call (LinearSolver_CholMod.h:86-94) -- is `value`; (b) the numeric phases alone (cholmod_factorize +
    cholmod_solve, the like-for-like of the GPU's warm step) and (c) its fastest solver with a cached analysis, the native
    block Cholesky (CLinearSolver_UberBlock::Solve_PosDef_Blocky, second call) are reported beside it.  Also returns the
    rel-inf distance of the GPU's solution from the reference's."""
    from oracle import oracle_lib as O
    with tempfile.TemporaryDirectory() as td:
        if O.have_reference():
            path, xp = os.path.join(td, "c3.bin"), os.path.join(td, "x.bin")
            lam.save(path)
            t0 = time.perf_counter()
            r = O.reference_solve(path, "cholmod_auto", xp, reps=budget_reps)
            wall = time.perf_counter() - t0
            ms = float(np.median(r["times_ms"]))
            x_ref = np.fromfile(xp, dtype=np.float64)
            out = {"value": counts["flops"] / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "reference",
                   "ms_per_solve": ms, "sample": f"{budget_reps} x CLinearSolver_CholMod(CHOLMOD_AUTO, AMD)::Solve_PosDef "
                   f"on the same 100k-pose system (median; {wall:.1f} s of CPU incl. load); serial in the reference; built -O3 -march=x86-64-v3 "
                   "(the reference's own flag is -march=native; the binary has to run on another host)",
                   "x_gpu_vs_reference_rel_inf": float(np.abs(x_gpu - x_ref).max() / np.abs(x_ref).max())}
            try:
                ph = subprocess.run([O.REF_HARNESS, "cholmod_phases", path, "auto", "3"], capture_output=True, text=True, timeout=300,
                                    env=O.reference_env())
                reps = json.loads([l for l in ph.stdout.splitlines() if l.startswith("{")][-1])["reps"]
                med = {k: float(np.median([q[k] for q in reps])) for k in ("convert_ms", "analyze_ms", "factorize_ms", "solve_ms")}
                out["cholmod_phases_ms"] = med
                out["numeric_only_ms"] = med["factorize_ms"] + med["solve_ms"]
                xu = os.path.join(td, "x_ub.bin")
                ub = O.reference_solve(path, "uberblock", xu, reps=3)
                out["native_block_solver_ms"] = {"first_call": float(ub["times_ms"][0]), "warm": float(min(ub["times_ms"][1:]))}
                # "the acceptance report must print cond-proxy + inter-oracle spread beside our error"
                xs = [x_ref, np.fromfile(xu, dtype=np.float64)]
                xc = os.path.join(td, "x_cs.bin")
                if O.reference_solve(path, "csparse", xc, reps=1)["ok"]:
                    xs.append(np.fromfile(xc, dtype=np.float64))
                out["inter_oracle_spread"] = max(float(np.abs(a - b).max() / np.abs(b).max()) for a in xs for b in xs if a is not b)
                out["inter_oracle_solvers"] = "CHOLMOD (auto), native block Cholesky, CSparse" if len(xs) == 3 else "CHOLMOD (auto), native block Cholesky"
                out["cond_proxy"] = O.solve_sparse(lam)[2].get("cond_proxy")   # (max / min diagonal of R)^2 <= cond_2, natural order, CPU restatement
            except Exception as e:      # the headline baseline stands without the split
                out["phases_error"] = str(e)[:200]
            return out
    t0 = time.perf_counter()
    ok, _, _ = O.solve_sparse(lam)
    ms = (time.perf_counter() - t0) * 1e3
    return {"value": counts["flops"] / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "ms_per_solve": ms, "sample": "1 x oracle/slampp_oracle.c up-looking block Cholesky, natural order"}
def replicas_leg(lam, counts, dev, local_rank, torch, ks=(1, 2, 4, 8), steps=10):
    """SURVEY.md section 8(e), second row: pose graphs do not shard -- "replicas only (multiple independent problems /
    damping values per GPU)".  K independent solves of the SAME structure on ONE device, each on a handle (and HIP stream)
    of its own: what an LM loop trying K damping values at once, or K robots' graphs, would enqueue
    (/root/reference/include/slam/NonlinearSolver_Lambda_LM.h:967-1001, 1660-1676: the reference re-damps and re-solves one after
    the other).  A single C3 solve is a chain of 12 dependent launches that fills a fraction of the chip; K chains side by
    side is the throughput the device has for this workload.  Reported: aggregate GFLOP/s and the whole-step HBM fraction
    on SURVEY 8d's algorithmic bytes, per K."""
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
    k_max = max(ks)
    solvers = [CLinearSolver_HIP(device=local_rank) for _ in range(k_max)]
    for s_ in solvers:
        s_.SymbolicDecomposition_Blocky(lam)
    vals = [torch.from_numpy(lam.values).to(dev) for _ in range(k_max)]     # K value sets (damped copies, in the LM reading)
    torch.cuda.synchronize()
    for k_, v_ in enumerate(vals):
        solvers[k_].apply_damping_device_async(v_.data_ptr(), 1e-3 * k_, 0, lam.n_bcols)
        solvers[k_].sync()
    rhs0 = torch.from_numpy(lam.rhs).to(dev)
    out = {"workload": f"K concurrent solves of the C3 structure on one device, a handle and a stream each; K value sets (damping 1e-3 k)", "by_k": {}}
    bytes_step = counts["factor_bytes"] + counts["solve_bytes"]
    for K in ks:
        bufs = [[rhs0.clone() for _ in range(steps + 1)] for _ in range(K)]
        torch.cuda.synchronize()
        for k_ in range(K):
            solvers[k_].factor_solve_device_async(vals[k_].data_ptr(), bufs[k_][0].data_ptr())
        assert all(solvers[k_].sync() for k_ in range(K))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            for k_ in range(K):
                solvers[k_].factor_solve_device_async(vals[k_].data_ptr(), bufs[k_][i].data_ptr())
        ok = all(solvers[k_].sync() for k_ in range(K))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        x = bufs[K - 1][-1].cpu().numpy()
        out["by_k"][str(K)] = {"ok": bool(ok), "ms_per_round": dt * 1e3, "solves_per_s": K / dt,
                                "GFLOP/s": K * counts["flops"] / dt / 1e9, "hbm_frac_whole_step": K * bytes_step / dt / 1e9 / HBM_PEAK_GBS,
                                "finite": bool(np.isfinite(x).all())}
    out["speedup_k8_vs_k1"] = out["by_k"][str(k_max)]["solves_per_s"] / out["by_k"]["1"]["solves_per_s"] if "1" in out["by_k"] else None
    # ... and the same K value sets through ONE handle in ONE pass of launches (slampp_hip_factor_solve_batch_device_async):
    # the chain of dependent launches as long as for one system, every launch K times as wide
    del solvers[1:]
    solver = solvers[0]
    n_v, n_s = lam.values.shape[0] + lam.values.shape[0] % 2, lam.n_scalars + lam.n_scalars % 2
    vb = torch.zeros(k_max * n_v, dtype=torch.float64, device=dev)
    for k_ in range(k_max):
        vb[k_ * n_v:k_ * n_v + lam.values.shape[0]] = vals[k_]
    del vals
    out["batched"] = {}
    for K in ks:
        rb = [torch.zeros(K * n_s, dtype=torch.float64, device=dev) for _ in range(steps + 1)]
        for r_ in rb:
            for k_ in range(K):
                r_[k_ * n_s:k_ * n_s + lam.n_scalars] = rhs0
        torch.cuda.synchronize()
        solver.factor_solve_batch_device_async(K, vb.data_ptr(), n_v, rb[0].data_ptr(), n_s)
        ok = all(solver.sync_batch(K))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            solver.factor_solve_batch_device_async(K, vb.data_ptr(), n_v, rb[i].data_ptr(), n_s)
        ok = ok and all(solver.sync_batch(K))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # parity guard: the last member's solution against the residual of ITS system (damped by 1e-3 (K - 1))
        x = rb[-1][(K - 1) * n_s:(K - 1) * n_s + lam.n_scalars].cpu().numpy()
        resid = float(np.abs(lam.to_scipy() @ x + 1e-3 * (K - 1) * x - lam.rhs).max() / np.abs(lam.rhs).max())
        out["batched"][str(K)] = {"ok": bool(ok), "ms_per_round": dt * 1e3, "solves_per_s": K / dt, "GFLOP/s": K * counts["flops"] / dt / 1e9,
                                  "hbm_frac_whole_step": K * bytes_step / dt / 1e9 / HBM_PEAK_GBS, "last_member_resid": resid}
    return out
def marginals_leg_c3(args, solver, lam, vals, dev, torch):
    """Outside the timed region: the block diagonal of the covariance Lambda^-1 of the same pose graph (numeric
    factorization + sparse inverse subset on the factor's pattern + extraction), next to the reference's recipe for it
    (ordering, CholeskyOf_FBS, CMarginals::Calculate_DenseMarginals_Recurrent_FBS) on the host."""
    n, d = lam.n_bcols, int(lam.cumsum[1])
    cov = torch.empty(n * d * d, dtype=torch.float64, device=dev)
    lib, h = solver._lib, solver._h
    solver._check(lib.slampp_hip_marginals_device_async(h, vals.data_ptr(), cov.data_ptr()))
    if not solver.sync():
        return None
    solver.profile(reset=True)
    reps = 10
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        solver._check(lib.slampp_hip_marginals_device_async(h, vals.data_ptr(), cov.data_ptr()))
    ok = solver.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    prof = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items() if v[0]}
    solver.profile(reset=True)
    c_np = cov.cpu().numpy().reshape(n, d, d)
    err = 0.0
    for c in (n // 3, n - 1):   # column j of the covariance is the solution of Lambda x = e_j
        e = np.zeros(lam.n_scalars)
        e[d * c] = 1.0
        if not solver.Solve_PosDef_Blocky(lam, e):
            return None
        err = max(err, float(np.abs(e[d * c:d * c + d] - c_np[c][:, 0]).max() / np.abs(c_np[c][:, 0]).max()))
    out = {"workload": f"block diagonal of Lambda^-1: {n} blocks {d}x{d}", "ok": bool(ok), "ms_per_call": ms, "phases_ms": prof,
           "column_check_rel_inf": err}
    if not args.no_cpu_baseline:
        from oracle import oracle_lib as O
        if O.have_reference():
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "c3.bin")
                lam.save(path)
                t0 = time.perf_counter()
                r = subprocess.run([O.REF_HARNESS, "sparse_marginals", path, os.path.join(td, "m")], capture_output=True, text=True,
                                   timeout=900, env=O.reference_env())
                wall = time.perf_counter() - t0
            if '"ok": true' in r.stdout:
                out["cpu_baseline"] = {"value": wall * 1e3, "unit": "ms", "cores": 1, "kind": "reference",
                                       "sample": "block ordering, CholeskyOf_FBS and CMarginals::Calculate_DenseMarginals_Recurrent_FBS"
                                                 "(.., mpart_Diagonal) on the same system, incl. load"}
    return out
def assembly_leg(solver, lam, dev, reps=20, rd=None, column_vertex_first=False):
    """Outside the timed region: Lambda and eta of the same graph assembled on the device from synthetic per-edge
    Jacobians, written where the solver reads them.  HBM-bound: bytes in (J0, J1, Sigma^-1,
    error, weight per edge) + bytes out (Lambda values, eta) over the HIP-event time of the two kernels."""
    import torch
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLambdaAssembly_HIP
    col = np.repeat(np.arange(lam.n_bcols), np.diff(lam.bcol_ptr))
    off = lam.brow_idx != col
    v0, v1 = lam.brow_idx[off].astype(np.int64), col[off].astype(np.int64)
    if column_vertex_first:   # BA: vertex 0 of a projection edge is the landmark (EDGE_P2MC xyz_id cam_id), the later block column
        v0, v1 = v1, v0
    dims = np.diff(lam.cumsum)
    d = int(dims[0]) if rd is None else int(rd)
    es = synth.random_edge_set(dims, v0, v1, rd=d, seed=3)
    asm = CLambdaAssembly_HIP(solver, lam, v0, v1, d)
    bufs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (es.J0, es.J1, es.sigma_inv, es.err, es.weight)]
    values = torch.empty(lam.values.shape[0], dtype=torch.float64, device=dev)
    eta = torch.empty(lam.n_scalars, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    args = [t.data_ptr() for t in bufs] + [values.data_ptr(), eta.data_ptr(), es.unary_vertex, es.unary_factor, es.unary_error]
    asm.Refresh_Lambda_device(*args)
    solver.sync()
    solver.set_option("profile", 1)        # (level 3 keeps only the solve's own kernels)
    solver.profile(reset=True)
    for _ in range(reps):
        asm.Refresh_Lambda_device(*args)
    solver.sync()
    cnt, ms = solver.profile().get("assemble", (0, 0.0))
    ok = solver.factor_solve_device(values.data_ptr(), eta.data_ptr())     # the assembled system, solved where it lies
    n_bytes = 8 * (sum(int(np.prod(t.shape)) for t in bufs) + values.numel() + eta.numel())
    us = ms / max(cnt, 1) * 1e3
    return {"n_edges": int(v0.shape[0]), "us_per_assembly": us, "algorithmic_bytes": n_bytes,
            "achieved_GBs": n_bytes / (us * 1e-6) / 1e9 if us > 0 else None, "peak_GBs": HBM_PEAK_GBS,
            "assembled_system_solved": bool(ok)}
def run_c3(args, rank, world, local_rank, dist):
    import torch
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP

    dev = torch.device(f"cuda:{local_rank}")
    lam = synth.pose_chain(n=args.poses)
    solver = CLinearSolver_HIP(device=local_rank)
    analyze_ms = timed_cold_analysis(solver, lam)
    stats = solver.stats()
    counts = algorithmic_counts_c3(lam, stats)
    vals = torch.from_numpy(lam.values).to(dev)
    rhs0 = torch.from_numpy(lam.rhs).to(dev)
    bufs = [rhs0.clone() for _ in range(args.steps + args.warmup)]
    torch.cuda.synchronize()
    for k in range(args.warmup):
        solver.factor_solve_device_async(vals.data_ptr(), bufs[k].data_ptr())
    if not solver.sync():
        raise SystemExit("warm-up solve failed: not positive definite")
    solver.set_option("profile", 3)      # one event pair in the timed region: around the leaf kernel, the roofline's
    solver.profile(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        solver.factor_solve_device_async(vals.data_ptr(), bufs[k].data_ptr())
    ok = solver.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not ok:
        raise SystemExit("solve failed: not positive definite")
    if rank != 0:
        return None
    # parity guard on the last timed solution: ||Lambda x - eta||_inf / ||eta||_inf
    x = bufs[-1].cpu().numpy()
    resid = float(np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max())
    ms_per_step = dt / args.steps * 1e3
    prof = solver.profile()          # timed region: one event pair, around the leaf kernel (option profile = 3)
    # the split of the table below comes from a few extra, untimed steps: every event pair between two kernels costs
    # microseconds of stream time, and the timed region carries only the one the roofline needs
    solver.set_option("profile", 2)
    solver.profile(reset=True)
    extra = [rhs0.clone() for _ in range(5)]
    for t_ in extra:
        solver.factor_solve_device_async(vals.data_ptr(), t_.data_ptr())
    solver.sync()
    prof_fine = solver.profile()
    solver.set_option("profile", 3)
    solver.profile(reset=True)
    prof = dict(prof_fine, **{k_: v_ for k_, v_ in prof.items() if k_ in ("factor_leaves",)})
    n_stages, n_bottom = stats["n_stages"], stats["n_bottom_stages"]
    kb = per_kernel_bytes_sparse(solver.plan(), n_bottom)
    # the separator stages are one launch each -- the tasks as panels in LDS, next to them the updates the NEXT stage's blocks
    # receive from further down (one half-workgroup per factor block) -- plus one launch of those updates for the first of them
    # (the first panel stage has an update launch of its own only when it sits above wide one-wave-per-column stages; right
    # above the lane-per-task leaves its tasks bring in their updates themselves)
    launches = {"factor_leaves": 1, "factor_wide": max(n_bottom - 1, 1), "factor_upper": max(n_stages - n_bottom, 1) + (1 if n_bottom > 1 else 0),
                "forward": n_stages, "backward": n_stages}
    names = {"factor_leaves": "factor_simt_kernel", "factor_wide": "factor_stage_kernel<D, 1, 8, 32, 48>",
             "factor_upper": "factor_panel_kernel (slices of the elimination tree, one launch per stage)", "forward": "forward_stage_kernel",
             "backward": "backward_stage_kernel (+ backward_simt_kernel for the leaf subtrees where there are many)"}
    needles = {"factor_leaves": "factor_simt_kernel", "factor_wide": ", 1, 8, 32, 48>", "factor_upper": "factor_panel_kernel",
               "forward": "forward_stage_kernel", "backward": "::backward_s"}   # as rocprofv3 spells the kernels (backward_stage_ / backward_simt_)
    traffic, traffic_file = load_traffic("c3")
    kernels = []
    for ph, (cnt, tot_ms) in prof.items():
        if ph not in kb or cnt == 0:
            continue
        per_step_ms = tot_ms / cnt
        kernels.append({"kernel": names[ph], "launches_per_step": launches[ph], "ms_per_step": per_step_ms,
                        "avg_launch_us": per_step_ms / launches[ph] * 1e3,
                        "algorithmic_bytes_per_launch": kb[ph] / launches[ph],
                        "achieved_GBs": kb[ph] / (per_step_ms * 1e-3) / 1e9,
                        "timed_in": "timed region" if ph == "factor_leaves" else "5 extra steps after it",
                        "hbm_traffic_bytes_per_launch": kernel_traffic_mean(traffic, needles[ph]) if isinstance(needles[ph], str) else
                        (lambda parts: (sum(parts) / len(parts)) if all(p is not None for p in parts) else None)(
                            [kernel_traffic(traffic, n_) for n_ in needles[ph]])})
    kernels.sort(key=lambda k: -k["ms_per_step"])
    gpu_ms = sum(k["ms_per_step"] for k in kernels)
    for k in kernels:
        k["share_of_step_time"] = k["ms_per_step"] / gpu_ms if gpu_ms > 0 else None
    # two roofline objects: `roofline` is for the kernel the step spends most of its time in (since round 3 the panel kernel of
    # the separator slices: a chain of launches of 0.1-4 MB each, bound by dependent latency, and priced as what it is);
    # `roofline_leaf_kernel` for the one that moves the step's bytes -- the leaf kernel reads nearly all of Lambda and writes
    # nearly all of L in ONE launch.  `roofline_whole_step` prices the step as a whole.
    def roofline_of(k):
        return {"bound": "hbm", "kernel": k["kernel"], "achieved": k["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": k["achieved_GBs"] / HBM_PEAK_GBS,
                "traffic": k["hbm_traffic_bytes_per_launch"], "traffic_source": traffic_file,
                "traffic_measured_in": "builder's rocprofv3 --pmc run of this command (replayed from the committed file, not measured in this run)",
                "avg_launch_us": k["avg_launch_us"], "launches_per_step": k["launches_per_step"],
                "algorithmic_bytes_per_launch": k["algorithmic_bytes_per_launch"], "share_of_step_time": k["share_of_step_time"],
                "timed_in": k["timed_in"]}
    dom = kernels[0]
    leaf = max(kernels, key=lambda k: k["algorithmic_bytes_per_launch"])
    out = {
        "metric": "Lambda solve GFLOP/s (algorithmic factor+solve flops / wall-clock), 100k-pose SE(3)",
        "value": counts["flops"] * world / (dt / args.steps) / 1e9, "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C3: synthetic {lam.n_bcols}-pose SE(3) chain + loop closures, 6x6 blocks, "
                               f"{lam.n_blocks} upper blocks, n={lam.n_scalars}; numeric factor + 2 substitutions per step "
                               "(symbolic analysis cached, inputs resident in HBM)",
                   "parallelism": "1 GPU" if world == 1 else f"{world} independent replicas (path does not shard)"},
        "solve_residual_rel_inf": resid,
        "algorithmic": {"flops_per_step": counts["flops"], "factor_bytes": counts["factor_bytes"],
                        "solve_bytes": counts["solve_bytes"], "source": counts["source"]},
        "own_ordering": {"l_nnz": stats["l_nnz"], "factor_flops": stats["factor_flops"], "n_stages": n_stages,
                         "n_tasks": stats["n_tasks"], "analyze_ms_cold": analyze_ms},
        "roofline": roofline_of(dom), "roofline_leaf_kernel": roofline_of(leaf),
        "kernels": kernels,
        "roofline_whole_step": {"bound": "hbm", "achieved": (counts["factor_bytes"] + counts["solve_bytes"]) / (dt / args.steps) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": (counts["factor_bytes"] + counts["solve_bytes"]) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                                "launches_per_step": int(sum(k["launches_per_step"] for k in kernels))},
    }
    if args.c3_solve_only:     # (the profiling passes: per-kernel averages of the solve alone)
        return out
    out["assembly"] = assembly_leg(solver, lam, dev)
    if world == 1:
        out["replicas_one_gpu"] = replicas_leg(lam, counts, dev, local_rank, torch)
    if world == 1:
        out["marginals"] = marginals_leg_c3(args, solver, lam, vals, dev, torch)
    if world == 1:
        out["host_path"] = host_path_leg(lambda: CLinearSolver_HIP(device=local_rank), lam)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c3(lam, counts, x)
        out["solve_x_vs_reference_rel_inf"] = out["cpu_baseline"].get("x_gpu_vs_reference_rel_inf")
        out["inter_oracle_spread"] = out["cpu_baseline"].get("inter_oracle_spread")
        out["cond_proxy"] = out["cpu_baseline"].get("cond_proxy")
        out["dropin_cpp"] = dropin_leg(lam)
    return out
