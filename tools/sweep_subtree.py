"""Dev helper: subtree size on pose chains of several lengths."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
for n in (2000, 10000, 30000, 100000):
    lam = synth.pose_chain(n=n)
    vals = torch.from_numpy(lam.values).cuda()
    for sub in (2, 4, 8, 16, 32):
        s = CLinearSolver_HIP(subtree_size=sub)
        s.SymbolicDecomposition_Blocky(lam)
        bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(21)]
        torch.cuda.synchronize()
        s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
        t0 = time.perf_counter()
        for b in bufs[1:]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync()
        print(f"n={n} subtree={sub}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
