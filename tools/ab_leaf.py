"""Dev helper (round 4): C3 with / without the lane-per-task backward kernel of the leaf subtrees (option simt_backward)."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lam = synth.pose_chain(n=n)
s = CLinearSolver_HIP()
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).to(dev)
reps = 20
for sb in (0, 1, 0, 1):
    s.set_option("simt_backward", sb)
    s.set_option("profile", 0)
    bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(2 * reps + 1)]
    torch.cuda.synchronize()
    assert s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
    t0 = time.perf_counter()
    for b in bufs[1:reps + 1]:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync()
    dt = (time.perf_counter() - t0) / reps
    s.set_option("profile", 2)
    s.profile(reset=True)
    for b in bufs[reps + 1:]:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync()
    x = bufs[-1].cpu().numpy()
    res = np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max()
    print(f"simt_backward={sb}  {dt*1e3:.3f} ms  resid {res:.1e}  " + "  ".join(f"{k}={ms/max(c,1)*1e3:.0f}" for k, (c, ms) in s.profile().items()), flush=True)
# another right-hand side with the kept factor (the forward kernel reads the inverses the factorization no longer stored)
e = lam.rhs.copy() * 2.0
assert s.Solve_PosDef_Blocky(lam, lam.rhs.copy())
assert s.Solve_Again(e)
print("solve_again resid", np.abs(lam.to_scipy() @ e - 2.0 * lam.rhs).max() / np.abs(lam.rhs).max())
