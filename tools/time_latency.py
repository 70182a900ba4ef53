"""Dev helper: one solve at a time (enqueue + wait) against the back-to-back rate, C3 and C4."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
for name, lam, cls in (("C3", synth.pose_chain(n=100000), CLinearSolver_HIP), ("C4", synth.ba(1000, 500000, k=4, mode="band"), CLinearSolver_Schur_HIP)):
    s = cls()
    s.SymbolicDecomposition_Blocky(lam)
    vals = torch.from_numpy(lam.values).cuda()
    rhs = torch.from_numpy(lam.rhs).cuda()
    work = rhs.clone()
    torch.cuda.synchronize()
    s.factor_solve_device(vals.data_ptr(), work.data_ptr())
    ts = []
    for i in range(30):
        work.copy_(rhs); torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.factor_solve_device_async(vals.data_ptr(), work.data_ptr())
        t1 = time.perf_counter()
        s.sync()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts = np.array(ts[5:]) * 1e3
    bufs = [rhs.clone() for _ in range(20)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in bufs:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync(); rate = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name}: enqueue {np.median(ts[:, 0]):.3f} ms, enqueue + wait {np.median(ts[:, 1]):.3f} ms, back to back {rate:.3f} ms per solve", flush=True)
