"""Dev helper: leaf stage through panels (few leaf tasks) against the lane-per-task kernel, by problem size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
for n in (2000, 5000, 10000, 15000, 30000):
    lam = synth.pose_chain(n=n)
    vals = torch.from_numpy(lam.values).cuda()
    for opts in ({}, {"simt": 1}, {"simt": 0}, {"panel": 0}):
        s = CLinearSolver_HIP(**opts)
        s.SymbolicDecomposition_Blocky(lam)
        bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(21)]
        torch.cuda.synchronize()
        s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
        t0 = time.perf_counter()
        for b in bufs[1:]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync(); dt = (time.perf_counter() - t0) / 20
        print(f"n={n} {opts}: {dt * 1e3:.3f} ms", flush=True)
