"""Dev helper: phase times of the C3 solve (100k-pose SE(3)) with HIP-event profiling on."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lam = synth.pose_chain(n=n)
dev = torch.device("cuda:0")
s = CLinearSolver_HIP(**{a.split("=")[0]: int(a.split("=")[1]) for a in os.environ.get("SOLVER_OPTS", "").split(",") if a})  # e.g. SOLVER_OPTS=simt_width=16
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).to(dev)
reps = 20
bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(2 * reps + 1)]
torch.cuda.synchronize()
assert s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
t0 = time.perf_counter()
for b in bufs[1:reps + 1]:
    s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
s.sync(); dt = (time.perf_counter() - t0) / reps
s.set_option("profile", 1); s.profile(reset=True)
for b in bufs[reps + 1:]:
    s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
s.sync()
x = bufs[-1].cpu().numpy()
res = np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max()
print(f"factor+solve {dt*1e3:.3f} ms  resid {res:.2e}  " + "  ".join(f"{k}={ms/max(c,1)*1e3:.0f}us" for k, (c, ms) in s.profile().items()))
