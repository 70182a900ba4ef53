"""Dev helper: how the Venice-like leg's Schur complement is assembled (option schur_tiles)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
lam = synth.ba(1000, 500_000, mode="venice", seed=777)
vals = torch.from_numpy(lam.values).cuda()
for opts in ({}, {"schur_tiles": 3}, {"schur_tiles": 1}, {"schur_tiles": 0}):
    s = CLinearSolver_Schur_HIP(**opts)
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
    x = bufs[-1].cpu().numpy()
    pr = {k: round(ms / max(c, 1), 3) for k, (c, ms) in s.profile().items()}
    print(opts, "ok", ok, f"step {dt:.3f} ms", pr, flush=True)
