"""BASELINE configs 1 and 2 (Manhattan3500 and Sphere2500 look-alikes): one solve each, against the reference's native block solver."""
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

__all__ = ['run_small_configs']


def run_small_configs(args, local_rank):
    """BASELINE.json configs[0] and configs[1] (parity-test cases, not the benchmark workload): the Manhattan3500 SE(2)
    and Sphere2500 SE(3) look-alikes, warm numeric factor + solve on the GPU next to the reference's CHOLMOD on the
    host.  Reported as extra objects of the JSON line; N = 1 only."""
    import torch
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
    from oracle import oracle_lib as O
    dev = torch.device(f"cuda:{local_rank}")
    out = {}
    which = [k_ for k_ in args.small_configs.split(",") if k_]
    for key, name, make in (("C1", "Manhattan3500 SE(2) look-alike, 3x3 blocks", lambda: synth.manhattan(3500)),
                            ("C2", "Sphere2500 SE(3) look-alike, 6x6 blocks", lambda: synth.sphere(50, 50))):
        if key not in which:
            continue
        lam = make()
        solver = CLinearSolver_HIP(device=local_rank)
        analyze_ms = timed_cold_analysis(solver, lam)
        vals = torch.from_numpy(lam.values).to(dev)
        reps = 20
        bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(reps + 1)]
        torch.cuda.synchronize()
        if not solver.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr()):
            raise SystemExit(f"{key}: not positive definite")
        t0 = time.perf_counter()
        for b in bufs[1:]:
            solver.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        solver.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        x = bufs[-1].cpu().numpy()
        st = solver.stats()
        rec = {"workload": f"{name}, n={lam.n_scalars}", "ms_per_solve": ms, "analyze_ms_cold": analyze_ms,
               "dense_top_dim": st["schur_dim"], "n_stages": st["n_stages"],
               "solve_residual_rel_inf": float(np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max())}
        # the split of the step (five extra solves with every phase bracketed by events) and its roofline: the big separators
        # of a 2-D-like graph are factored as one dense matrix on the matrix cores (the "dense top": the flops of its columns
        # under our ordering, sum of squared column counts, against the fp64 MFMA peak); the block-by-block part below it is
        # a chain of small launches and is priced against HBM on its SURVEY 8d bytes
        solver.set_option("profile", 2)
        solver.profile(reset=True)
        extra = [torch.from_numpy(lam.rhs).to(dev) for _ in range(5)]
        for t_ in extra:
            solver.factor_solve_device_async(vals.data_ptr(), t_.data_ptr())
        solver.sync()
        prof = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items()}
        rec["phases_ms"] = prof
        plan = solver.plan()
        dim = plan["dim"].astype(np.int64)
        lptr, lrow, dpos = plan["lptr"], plan["lrow"].astype(np.int64), plan["dense_pos"]
        col_of = np.repeat(np.arange(len(dim)), np.diff(lptr))
        below = np.bincount(col_of, weights=dim[lrow], minlength=len(dim)) - dim      # scalar rows below the diagonal block
        t_ = np.arange(1, dim.max() + 1)
        col_flops = np.array([np.sum((below[j] + t_[:dim[j]]) ** 2) for j in range(len(dim))], dtype=np.float64)
        col_lnz = dim * (dim + 1) // 2 + dim * below
        top = dpos >= 0
        if "dense_chol" in prof and top.any():
            tf = float(col_flops[top].sum()) / (prof["dense_chol"] * 1e-3) / 1e12
            rec["roofline"] = {"bound": "mfma", "kernel": "dense top: tile-scheduled Cholesky on the matrix cores (tile_potrf / tile_trsm / tile_update, "
                               "or potrf_diag / trsm / syrk)", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / F64_MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_flops": float(col_flops[top].sum()),
                               "ms": prof["dense_chol"], "dense_top_dim": int(st["schur_dim"])}
        sparse_ms = sum(prof.get(k_, 0.0) for k_ in ("factor_leaves", "factor_wide", "factor_upper", "factor_rest"))
        if sparse_ms > 0:
            nbytes = 8.0 * (float(st["nnz_upper"]) + float(col_lnz[~top].sum()))
            rec["roofline_block_part"] = {"bound": "hbm", "kernel": "block-by-block elimination below the dense top (leaf subtrees + separator panels)",
                                          "achieved": nbytes / (sparse_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": nbytes / (sparse_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                          "algorithmic_bytes": nbytes, "ms": sparse_ms}
        if not args.no_cpu_baseline and O.have_reference():
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "p.bin")
                lam.save(path)
                r = O.reference_solve(path, "cholmod_auto", "-", reps=5)
            rec["reference_cholmod_ms"] = float(np.median(r["times_ms"]))
            # ... and the reference's fastest solver with its analysis cached (CLinearSolver_UberBlock::Solve_PosDef_Blocky,
            # calls after the first): the like-for-like of the GPU's warm step
            try:
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, "p.bin")
                    lam.save(path)
                    ub = O.reference_solve(path, "uberblock", "-", reps=5)
                rec["reference_native_block_solver_ms"] = {"first_call": float(ub["times_ms"][0]), "warm": float(min(ub["times_ms"][1:]))}
                rec["speedup_vs_reference_native_warm"] = float(min(ub["times_ms"][1:])) / ms
            except Exception as e:
                rec["reference_native_error"] = str(e)[:200]
        out[key] = rec
    return out
