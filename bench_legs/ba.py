"""The BA legs: BASELINE config 4 (Venice-like, band and uniform visibility) and config 5 / the north star's 1k x 1M system, one GPU or
landmark-sharded over the ranks (one all-reduce of the reduced camera system per step)."""
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
from .c3 import *  # noqa: F401,F403
from .small import *  # noqa: F401,F403

__all__ = ['cpu_baseline_ba', 'incremental_leg', 'marginals_leg', 'run_ba']


def cpu_baseline_ba(lam, flops, x_gpu):
    from oracle import oracle_lib as O
    if not O.have_reference():
        return None
    with tempfile.TemporaryDirectory() as td:
        path, xp = os.path.join(td, "ba.bin"), os.path.join(td, "x.bin")
        lam.save(path)
        t0 = time.perf_counter()
        r = O.reference_solve(path, "schur", xp, reps=2, timeout=900)
        wall = time.perf_counter() - t0
        x_ref = np.fromfile(xp, dtype=np.float64)
    ms = float(r["times_ms"][-1])
    return {"value": flops / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "cores": host_cores(), "kind": "reference",
            "ms_per_solve": ms, "ms_first_call": float(r["times_ms"][0]),
            "x_gpu_vs_reference_rel_inf": float(np.abs(x_gpu - x_ref).max() / np.abs(x_ref).max()),
            "sample": f"2 x CLinearSolver_Schur<CholMod>::Solve_PosDef[_Blocky] on the same system (second call, ordering reused; "
                      f"{wall:.1f} s of CPU incl. load); OpenMP only in the block-diagonal inverse and one SpMV, dense LLT serial"}
def incremental_leg(lam, dev, local_rank, torch, share=0.01, reps=5, always=False):
    """Outside the timed region: option schur_incremental.  After a relinearization that moved `share` of the landmarks the
    reduced camera system is updated from the previous one (the reference's dog-leg solver does that from Omega = delta
    Lambda, NonlinearSolver_Lambda_DL.h:2301-) instead of rebuilt: the same solve both ways, same values."""
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    rng = np.random.default_rng(3)
    nc, n_pts = lam.n_matrix_cut, lam.n_bcols - lam.n_matrix_cut
    points = np.sort(rng.choice(n_pts, size=max(int(share * n_pts), 1), replace=False))
    off = lam.block_value_offsets()
    vals2 = lam.values.copy()
    for p_ in points:                                             # the moved landmarks: more curvature, scaled projections
        k0, k1 = int(lam.bcol_ptr[nc + p_]), int(lam.bcol_ptr[nc + p_ + 1])
        vals2[off[k0]:off[k1 - 1]] *= 0.9
        vals2[off[k1 - 1]:off[k1]] += 0.5 * np.eye(3).ravel()
    solver = CLinearSolver_Schur_HIP(device=local_rank, schur_incremental=2 if always else 1)   # 2: use the list however long it is
    solver.SymbolicDecomposition_Blocky(lam)
    v1, v2 = torch.from_numpy(lam.values).to(dev), torch.from_numpy(vals2).to(dev)
    rhs = torch.from_numpy(lam.rhs).to(dev)
    out = {}
    for name, use_list in (("full_rebuild_ms", False), ("update_ms", True)):
        ms, xs = [], None
        for _ in range(reps):
            b1, b2 = rhs.clone(), rhs.clone()
            solver.factor_solve_device(v1.data_ptr(), b1.data_ptr())           # the system before the relinearization
            if use_list:
                solver.Set_Changed_Landmarks(points)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ok = solver.factor_solve_device(v2.data_ptr(), b2.data_ptr())
            ms.append((time.perf_counter() - t0) * 1e3)
            xs = b2.cpu().numpy()
        out[name] = float(np.median(ms))
        out[name.replace("_ms", "_x")] = xs
    x_full, x_upd = out.pop("full_rebuild_x"), out.pop("update_x")
    out["update_vs_full_rel_inf"] = float(np.abs(x_upd - x_full).max() / np.abs(x_full).max())
    out["changed_landmarks"] = int(len(points))
    out["ok"] = bool(ok)
    out["note"] = ("option schur_incremental = 1: the list is used when it is the shorter way (up to 1/32 of the landmarks "
                   "with the landmark-major assembly); a longer list is answered with the full rebuild")
    return out
def marginals_leg(args, solver, lam, vals, dev, torch):
    """Block diagonal of the covariance on the bench's BA system, after the timed solves:
    the reduced system assembled and factored as for a solve, the blocks of its inverse the landmarks need taken from
    a sparse inverse subset on the factor's pattern -- or, `dense_inverse`, S inverted on the matrix cores (2 n^3 / 3
    flops) -- then gathered per landmark.  The reference's CSchurComplement_Marginals is run beside it on a bounded
    sample (--no-cpu-baseline skips it)."""
    nc, n_pts = lam.n_matrix_cut, lam.n_bcols - lam.n_matrix_cut
    cams = torch.empty(nc * 36, dtype=torch.float64, device=dev)
    pts = torch.empty(n_pts * 9, dtype=torch.float64, device=dev)
    def timed():
        solver.schur_marginals_device_async(vals.data_ptr(), cams.data_ptr(), pts.data_ptr())
        if not solver.sync():
            return None
        solver.profile(reset=True)
        reps = 3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            solver.schur_marginals_device_async(vals.data_ptr(), cams.data_ptr(), pts.data_ptr())
        ok_ = solver.sync()
        ms_ = (time.perf_counter() - t0) / reps * 1e3
        prof_ = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items() if k_.startswith("marginals")}
        solver.profile(reset=True)
        return ok_, ms_, prof_

    # first the way the library picks (the sparse inverse subset when the solves factor S by the sparse block path), then
    # with S inverted densely on the matrix cores: that one has the MFMA roofline
    n = 6.0 * nc
    solver.set_option("marginals_dense", 1)
    r_dense = timed()
    solver.set_option("marginals_dense", 0)
    r = timed()
    if r is None or r_dense is None:
        return None
    ok, ms, prof = r
    tf = 2.0 * n ** 3 / 3.0 / (r_dense[2]["marginals_inverse"] * 1e-3) / 1e12
    # a sampled check against the definition: column j of the covariance is the solution of Lambda x = e_j
    c_np, p_np = cams.cpu().numpy().reshape(nc, 6, 6), pts.cpu().numpy().reshape(n_pts, 3, 3)
    err = 0.0
    for (idx, blk, d, base) in ((nc // 3, c_np, 6, 0), (n_pts // 2, p_np, 3, 6 * nc)):
        e = np.zeros(lam.n_scalars)
        e[base + d * idx] = 1.0
        if not solver.Solve_PosDef_Blocky(lam, e):
            return None
        ref = e[base + d * idx: base + d * idx + d]
        err = max(err, float(np.abs(ref - blk[idx][:, 0]).max() / np.abs(ref).max()))
    out = {"workload": f"block diagonal of Lambda^-1: {nc} camera blocks 6x6 + {n_pts} landmark blocks 3x3", "ok": bool(ok),
           "ms_per_call": ms, "phases_ms": prof, "column_check_rel_inf": err,
           "dense_inverse": {"ok": bool(r_dense[0]), "ms_per_call": r_dense[1], "phases_ms": r_dense[2],
                             "roofline": {"bound": "mfma",
                                          "kernel": "inverse_level_kernel + inverse_lauum_kernel (inverse of S from its factor)",
                                          "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": tf / F64_MFMA_PEAK_TFLOPS, "traffic": None, "flops": 2.0 * n ** 3 / 3.0,
                                          "ms": r_dense[2]["marginals_inverse"]}}}
    if not args.no_cpu_baseline:
        from oracle import oracle_lib as O
        if O.have_reference():
            import subprocess
            sample = dataclasses_replace_points(lam, min(n_pts, 100_000))
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "ba.bin")
                sample.save(path)
                t0 = time.perf_counter()
                r = subprocess.run([O.REF_HARNESS, "schur_marginals", path, os.path.join(td, "m")], capture_output=True, text=True,
                                   timeout=900, env=O.reference_env())
                wall = time.perf_counter() - t0
            if '"ok": true' in r.stdout:
                out["cpu_baseline"] = {"value": wall * 1e3, "unit": "ms", "cores": host_cores(), "kind": "reference",
                                       "sample": f"CSchurComplement_Marginals::Schur_Marginals on the first "
                                                 f"{sample.n_bcols - nc} landmarks of the same system (all {nc} cameras), with the "
                                                 f"Schur complement and its Cholesky factor it needs, incl. load; OpenMP"}
    return out
def run_ba(args, rank, world, local_rank, dist, schur_sparse=-1, mode="band", extras=True, cams=None, points=None, label=None):
    """One BA system of `cams` cameras x `points` landmarks in total (default: C4, --ba-cams x --ba-points), solved
    through the Schur complement.  N > 1: the SAME system cut into N landmark shards, one per rank (strong scaling), the
    partial reduced camera systems summed by one RCCL all-reduce per step.
    schur_sparse: -1 = the library decides how to factor the reduced camera system (sparse block path when under 15 %
    of its blocks are nonzero), 0 = force the dense MFMA factorization."""
    import torch
    from slam_plus_plus_amd import synth, sharding
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP

    dev = torch.device(f"cuda:{local_rank}")
    k = 4
    n_cams, n_points_total = cams or args.ba_cams, points or args.ba_points
    lam_full = shared_system(f"ba_{n_cams}x{n_points_total}_{mode}",
                             lambda: synth.ba(n_cams, n_points_total, k=k, mode=mode, seed=777), rank, world, dist)
    if world > 1 and rank == 0 and label == "C5":
        _KEEP["C5"] = lam_full      # the device group leg solves the same system after the ranks are done
    if world > 1:
        lam, own = sharding.landmark_shard(lam_full, rank, world)   # A and eta_x as 1 / world on every rank: the sum is the system
    else:
        lam, own = lam_full, slice(int(lam_full.cumsum[lam_full.n_matrix_cut]), lam_full.n_scalars)
    solver = CLinearSolver_Schur_HIP(device=local_rank, schur_sparse=schur_sparse)
    analyze_ms = timed_cold_analysis(solver, lam)
    if dist is not None:
        solver.set_option("shard_rank", rank)
        solver.set_option("shard_world", world)
        solver.set_allreduce(make_allreduce(dist, torch, dev))
    st = solver.stats()
    vals = torch.from_numpy(lam.values).to(dev)
    steps, warmup = args.ba_steps, 1
    bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(steps + warmup)]
    torch.cuda.synchronize()
    for i in range(warmup):
        solver.factor_solve_device_async(vals.data_ptr(), bufs[i].data_ptr())
    if not solver.sync():
        raise SystemExit("BA warm-up solve failed")
    solver.set_option("profile", 3)      # event pairs only around the kernels the rooflines are about
    solver.profile(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        solver.factor_solve_device_async(vals.data_ptr(), bufs[i].data_ptr())
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
        raise SystemExit("BA solve failed: not positive definite")
    prof = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items()}   # timed region: the roofline kernels' phases
    # the other phases from three extra, untimed steps with every phase bracketed (an event pair costs microseconds);
    # every rank runs them: the solve holds a collective
    solver.set_option("profile", 1)
    solver.profile(reset=True)
    extra_rhs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(3)]
    torch.cuda.synchronize()
    for t_ in extra_rhs:
        solver.factor_solve_device_async(vals.data_ptr(), t_.data_ptr())
    solver.sync()
    del extra_rhs
    prof = dict({k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items()}, **prof)
    # ... and the same three steps under the names the reference prints with __SCHUR_PROFILING (LinearSolver_Schur.h:1895-1912)
    prof_ref_names = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile_reference_names().items()}
    solver.profile(reset=True)
    red = solver.reduced_stats()
    x_local = bufs[-1].cpu().numpy()
    totals = np.array([st["n_points"], st["n_observations"], st["n_update_pairs"]], dtype=np.float64)
    phases_by_rank, parity = [prof], None
    if world > 1:
        # the whole system's counts, every rank's phases, and the parity guard of the sharded solution: the residual of the
        # FULL system, assembled from every rank's landmarks (rank 0 holds the full matrix: it built it)
        t = torch.from_numpy(totals).to(dev)
        dist.all_reduce(t)
        totals = t.cpu().numpy()
        phases_by_rank = [None] * world
        dist.all_gather_object(phases_by_rank, prof)
        pieces = [None] * world
        dist.all_gather_object(pieces, (own.start, own.stop, x_local[int(lam_full.cumsum[lam_full.n_matrix_cut]):]))
        if rank == 0:
            x_full = np.empty(lam_full.n_scalars)
            x_full[:int(lam_full.cumsum[lam_full.n_matrix_cut])] = x_local[:int(lam_full.cumsum[lam_full.n_matrix_cut])]
            for a_, b_, piece in pieces:
                x_full[a_:b_] = piece
            parity = float(np.abs(lam_full.to_scipy() @ x_full - lam_full.rhs).max() / np.abs(lam_full.rhs).max())
    if rank != 0:
        return None
    ms = dt / steps * 1e3
    n_values_total, n_scalars_total = int(lam_full.values.shape[0]), int(lam_full.n_scalars)
    dc_ = int(lam_full.cumsum[1] - lam_full.cumsum[0])
    n_pts, n_obs, n_pairs, N = int(totals[0]), int(totals[1]), int(totals[2]), st["schur_dim"]
    # per point with k observations 58 + 108 k + 216 k (k + 1) / 2 flops for the Schur
    # products, 2 flops per stored scalar of U for each of the 3 SpMV passes, n^3/3 + ... for the dense factor
    schur_flops = n_pts * 58.0 + 108.0 * n_obs + 216.0 * n_pairs + 3 * 2.0 * 18 * n_obs   # the whole system's (all shards)
    dense_flops = st["factor_flops"] + st["solve_flops"]
    b_dense = "dense_chol" in prof
    # the factorization of the reduced system is redundant on every rank: counted once -- n^3/3 for the dense one, the inner
    # plan's own count (sum of squared column counts under our ordering + 4 nnz(L)) for the sparse one
    reduced_flops = dense_flops if b_dense else red["factor_flops"] + red["solve_flops"]
    flops = schur_flops + reduced_flops
    out = {
        "workload": f"{label or ('C4' if (n_cams, n_points_total) == (1000, 500_000) else 'BA')}: BA {n_cams} cams x {n_points_total} points"
                    f"{'' if world == 1 else f' as {world} landmark shards (one fixed system: strong scaling)'}, "
                    f"{'2..30 (mean 5.3)' if mode == 'venice' else k} obs/point, {mode} visibility; Schur complement + "
                    f"{'dense (MFMA)' if b_dense else 'sparse block'} factorization of the reduced system, per step",
        "reduced_system": "dense" if b_dense else "sparse",
        "ms_per_step": ms, "points_per_s": n_pts / (dt / steps), "GFLOP/s": flops / (dt / steps) / 1e9,
        "n_gpus": world, "steps": steps, "schur_dim": N, "n_observations": n_obs, "analyze_ms_cold": analyze_ms,
        "phases_ms": prof, "phases_ms_reference_names": prof_ref_names,
        "n_camera_pair_blocks": st["l_blocks"], "n_contributions": n_pairs,
        "algorithmic_flops": {"schur": schur_flops, "reduced_system": reduced_flops},
        # what a caller's host arrays hold and what the ranks exchange (scaling_model): packed values, scalars, and the
        # nonzero camera-pair blocks of S + the reduced right-hand side (the whole lower triangle when S is dense)
        "n_values": n_values_total, "n_scalars": n_scalars_total,
        "n_exchange_doubles": (N * (N + 1) // 2 if b_dense else int(st["l_blocks"]) * dc_ * dc_) + N,
    }
    if world > 1:
        out["phases_ms_by_rank"] = phases_by_rank
        out["solve_residual_rel_inf"] = parity
        out["exchange"] = "torch.distributed all_reduce (RCCL) of the packed blocks of S + the reduced right-hand side, on the solver's stream"
    # the committed counter passes are of the C4 legs on one GPU: other sizes and shards have no measured traffic
    b_c4_single = world == 1 and (cams or args.ba_cams) == 1000 and (points or args.ba_points) == 500_000
    traffic, traffic_file = load_traffic("ba_" + mode) if b_c4_single else ({}, None)
    if b_c4_single and not traffic and mode == "band":
        traffic, traffic_file = load_traffic("ba")   # (rounds 1 and 2 named the band leg's file so)
    if "dense_chol" in prof:
        tf = st["factor_flops"] / (prof["dense_chol"] * 1e-3) / 1e12
        n_panels = (N + 1 + 63) // 64
        tr = None
        if traffic and all(any(k_ in n for n in traffic) for k_ in ("potrf_diag_kernel", "trsm_kernel")):
            # HBM bytes of the whole factorization = sum over its kernels of launches x bytes per launch (the updates ride in
            # the potrf_diag launches; syrk_kernel launches of its own only in the files of rounds 1 and 2)
            present = [k_ for k_ in ("potrf_diag_kernel", "trsm_kernel", "syrk_kernel") if any(k_ in n for n in traffic)]
            tr = sum((kernel_traffic(traffic, k_) or 0.0) * traffic[[n for n in traffic if k_ in n][0]]["launches"]
                     for k_ in present) / \
                max(traffic[[n for n in traffic if "potrf_diag_kernel" in n][0]]["launches"] / n_panels, 1)
        out["roofline_dense_cholesky"] = {"bound": "mfma", "kernel": "dense_cholesky (potrf_diag_kernel with the updates riding + trsm_kernel, one factorization)",
                           "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TFLOPS,
                           "sustained_matrix_rate_measured": F64_MFMA_SUSTAINED_TFLOPS, "traffic": tr, "traffic_source": traffic_file, "flops_per_factorization": st["factor_flops"],
                           "ms_per_factorization": prof["dense_chol"]}
    if "schur_tiles" in prof:
        # landmark-major assembly (schur_tiles.hip): every landmark's column of Lambda is read once -- 144 B per observation,
        # 72 B of C and 24 B of l per landmark --, C^-1 written (72 B), every block of S read and written once
        # (per launch = this rank's shard)
        nbytes = 144.0 * st["n_observations"] + (72.0 + 24.0 + 72.0) * st["n_points"] + 2 * 8.0 * 36 * st["l_blocks"]
        gb = nbytes / (prof["schur_tiles"] * 1e-3) / 1e9
        tr = None
        if traffic:
            parts = [kernel_traffic_sum(traffic, k_) for k_ in ("schur_run_kernel", "schur_tile_kernel", "schur_tile_reduce_kernel")]
            tr = sum(p for p in parts if p) or None
        out["roofline_schur_assembly"] = {
            "bound": "hbm", "kernel": "schur_run_kernel (+ schur_tile_kernel for landmarks outside runs, + schur_tile_reduce_kernel): "
                                      "S and r assembled landmark by landmark, one launch of each per step",
            "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS, "traffic": tr,
            "traffic_source": traffic_file,
            "traffic_measured_in": "builder's rocprofv3 --pmc run (replayed from the committed file)" if traffic_file else None,
            "algorithmic_bytes_per_step": nbytes, "ms_per_step": prof["schur_tiles"]}
        # ... and over EVERY assembly kernel of the step (run / tile / reduce kernels + what the contribution lists of the
        # landmarks in no run or tile cost: C^-1, W, gather, right-hand side), on the same algorithmic bytes: what the
        # reference's steps a13-a18 (LinearSolver_Schur.h:1743-1767) cost here as a whole
        asm_ms = sum(prof.get(k_, 0.0) for k_ in ("schur_tiles", "schur_gather", "schur_points", "schur_rhs", "schur_init"))
        tr_all = None
        if traffic:
            parts = [kernel_traffic_sum(traffic, k_) for k_ in ("schur_run_kernel", "schur_tile_kernel", "schur_tile_reduce_kernel", "schur_gather_S_kernel",
                                                                "schur_obs_W_kernel", "schur_rhs_kernel", "schur_point_inverse_kernel", "schur_scatter_A_kernel",
                                                                "schur_wide_kernel")]
            tr_all = sum(p for p in parts if p) or None
        out["roofline_schur_assembly_all"] = {
            "bound": "hbm", "kernel": "every kernel of the assembly of S and r (landmark-major runs / tiles / reduction + contribution lists)",
            "achieved": nbytes / (asm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (asm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": tr_all, "traffic_source": traffic_file, "algorithmic_bytes_per_step": nbytes, "ms_per_step": asm_ms}
    if "schur_gather" in prof and prof["schur_gather"] > 0.02 and "schur_tiles" not in prof:
        gb = (288.0 * st["n_update_pairs"] + 2 * 8.0 * 36 * st["l_blocks"]) / (prof["schur_gather"] * 1e-3) / 1e9
        out["roofline_schur_gather"] = {"bound": "hbm", "kernel": "schur_gather_S_kernel", "achieved": gb, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                                        "traffic": kernel_traffic(traffic, "3, 8>" if mode != "uniform" else "3, 1>"),
                                        "traffic_source": traffic_file,
                                        "traffic_measured_in": "builder's rocprofv3 --pmc run (replayed from the committed file)",
                                        "ms_per_launch": prof["schur_gather"]}
    if "reduced_sparse" in prof and red["l_nnz"] > 0:
        # the reduced camera system through the sparse block path (inner plan: its own nested dissection, a dense top on the
        # matrix cores where it has one).  Priced both ways: SURVEY 8d bytes (8 (nnz + lnz) for the factor, 16 lnz + 32 n for
        # the substitutions) against HBM, and the inner plan's flops against the fp64 MFMA peak -- with a dense top the
        # flops are what the phase is made of, without one it is a chain of small launches and neither roof is near
        r_bytes = 8.0 * (red["nnz_upper"] + red["l_nnz"]) + 16.0 * red["l_nnz"] + 32.0 * red["n_scalars"]
        r_flops = red["factor_flops"] + red["solve_flops"]
        t_s = prof["reduced_sparse"] * 1e-3
        b_top = red["schur_dim"] > 0
        out["roofline_reduced_sparse"] = {
            "bound": "mfma" if b_top else "hbm", "kernel": "reduced camera system: sparse block Cholesky + substitutions "
            f"({red['n_stages']} stages{', dense top of dimension %d on the matrix cores' % red['schur_dim'] if b_top else ''})",
            "achieved": (r_flops / t_s / 1e12) if b_top else (r_bytes / t_s / 1e9),
            "peak": F64_MFMA_PEAK_TFLOPS if b_top else HBM_PEAK_GBS, "unit": "TFLOP/s" if b_top else "GB/s",
            "frac": (r_flops / t_s / 1e12 / F64_MFMA_PEAK_TFLOPS) if b_top else (r_bytes / t_s / 1e9 / HBM_PEAK_GBS),
            "traffic": None, "algorithmic_flops": r_flops, "algorithmic_bytes": r_bytes, "ms_per_step": prof["reduced_sparse"],
            "l_nnz": red["l_nnz"], "dense_top_dim": red["schur_dim"]}
    # the leg's "roofline" is the one of the phase the step spends most of its time in (each stays under its own name as well)
    phase_ms = {"roofline_dense_cholesky": prof.get("dense_chol", 0.0),
                "roofline_schur_assembly_all": sum(prof.get(k_, 0.0) for k_ in ("schur_tiles", "schur_gather", "schur_points", "schur_rhs", "schur_init")),
                "roofline_schur_gather": prof.get("schur_gather", 0.0),
                "roofline_reduced_sparse": prof.get("reduced_sparse", 0.0)}
    present = [k_ for k_ in phase_ms if k_ in out]
    if present:
        k_dom = max(present, key=lambda k_: phase_ms[k_])
        out["roofline"] = dict(out[k_dom], name=k_dom, phase_ms=phase_ms[k_dom], share_of_step=phase_ms[k_dom] / ms,
                               other_phases_ms={k_: phase_ms[k_] for k_ in present if k_ != k_dom})
    if world == 1:
        x = bufs[-1].cpu().numpy()
        out["solve_residual_rel_inf"] = float(np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max())
        if not args.no_cpu_baseline and schur_sparse != 0 and extras:
            # the reference factors S densely whatever its structure: n^3/3 for it even where the GPU path counts none
            n3 = float(N) ** 3 / 3.0
            out["cpu_baseline"] = cpu_baseline_ba(lam, schur_flops + (dense_flops if b_dense else n3), x)
            out["solve_x_vs_reference_rel_inf"] = out["cpu_baseline"]["x_gpu_vs_reference_rel_inf"] if out["cpu_baseline"] else None
        if extras and schur_sparse != 0:
            out["host_path"] = host_path_leg(lambda: CLinearSolver_Schur_HIP(device=local_rank), lam, reps=3)
            if not args.no_cpu_baseline:
                out["dropin_cpp"] = dropin_leg(lam, reps=3)
        if extras and schur_sparse != 0:
            out["marginals"] = marginals_leg(args, solver, lam, vals, dev, torch)
            out["incremental_schur_update"] = incremental_leg(lam, dev, local_rank, torch, share=0.01)       # the list is used
            out["incremental_schur_update_10pct"] = incremental_leg(lam, dev, local_rank, torch, share=0.1)  # answered with a rebuild
            # Lambda of the same structure assembled on the device from 2-d projection residuals (one edge per observation)
            out["assembly"] = assembly_leg(solver, lam, dev, reps=5, rd=2, column_vertex_first=True)
    return out
