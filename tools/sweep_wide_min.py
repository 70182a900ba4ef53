import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
for n in (100000, 1000000, 30000):
    lam = synth.pose_chain(n=n)
    vals = torch.from_numpy(lam.values).cuda()
    for wm in (256, 512, 1024, 2048, 4096):
        s = CLinearSolver_HIP(wide_min_tasks=wm)
        s.SymbolicDecomposition_Blocky(lam)
        bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(21)]
        torch.cuda.synchronize()
        s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
        t0 = time.perf_counter()
        for b in bufs[1:]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync()
        print(f"n={n} wide_min_tasks={wm}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
