"""Dev helper: where the analysis of the Venice-like BA leg spends its time (SLAMPP_HIP_PLAN_TIMING=1), and its first solve."""
import sys, os, time
os.environ["SLAMPP_HIP_PLAN_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
lam = synth.ba(1000, 500_000, mode="venice", seed=777)
s = CLinearSolver_Schur_HIP()
s.SymbolicDecomposition_Blocky(synth.ba(20, 600, seed=3))
for i in range(2):
    s.Clear_SymbolicDecomposition()
    t0 = time.perf_counter(); s.SymbolicDecomposition_Blocky(lam); print("analyze total %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
    vals = torch.from_numpy(lam.values).cuda(); rhs = torch.from_numpy(lam.rhs).cuda(); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.factor_solve_device(vals.data_ptr(), rhs.data_ptr()); print("first solve %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
