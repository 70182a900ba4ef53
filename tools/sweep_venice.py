"""Dev helper: the reduced camera system of the Venice-like leg (sparse block path = a dense top under a tile schedule)
against the ordering's knobs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
lam = synth.ba(1000, 500_000, mode="venice", seed=777)
vals = torch.from_numpy(lam.values).cuda()
for opts in ({}, {"leaf_size": 8}, {"leaf_size": 10}, {"leaf_size": 16}, {"leaf_size": 32}, {"nd_balance": 30}, {"nd_balance": 45},
             {"leaf_size": 10, "nd_balance": 40}, {"dense_top_tiles": 0}, {"natural_order": 1}, {"dense_top_nb": 0}, {"dense_top_nb": 64},
             {"dense_top_nb": 100}):
    s = CLinearSolver_Schur_HIP(schur_sparse=1, **opts)
    s.SymbolicDecomposition_Blocky(lam)
    bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(7)]
    torch.cuda.synchronize()
    ok = s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
    s.set_option("profile", 1); s.profile(reset=True)
    t0 = time.perf_counter()
    for b in bufs[1:]:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync()
    dt = (time.perf_counter() - t0) / 6 * 1e3
    red = s.reduced_stats()
    pr = {k: round(ms / max(c, 1), 3) for k, (c, ms) in s.profile().items()}
    print(opts, "ok", ok, f"step {dt:.3f} ms reduced_sparse {pr.get('reduced_sparse')} stages {red['n_stages']} dense_dim {red['schur_dim']} l_nnz {red['l_nnz']:.3g} flops {red['factor_flops']:.3g}", flush=True)
