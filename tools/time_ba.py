"""Dev helper: phase timing of the BA (Schur) path."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
mode = sys.argv[3] if len(sys.argv) > 3 else "band"
t0 = time.perf_counter(); lam = synth.ba(nc, npts, k=4, mode=mode); print("gen %.1fs" % (time.perf_counter() - t0), flush=True)
dev = torch.device("cuda:0")
extra = {a.split("=")[0]: int(a.split("=")[1]) for a in os.environ.get("SOLVER_OPTS", "").split(",") if a}  # e.g. SOLVER_OPTS=subtree_size=16
s = CLinearSolver_Schur_HIP(schur_sparse=int(os.environ.get("SCHUR_SPARSE", "-1")), schur_tiles=int(os.environ.get("SCHUR_TILES", "-1")), **extra)  # 0 forces the dense reduced system
t0 = time.perf_counter(); s.SymbolicDecomposition_Blocky(lam); print("analyze %.1f ms" % ((time.perf_counter() - t0) * 1e3), s.stats(), flush=True)
vals = torch.from_numpy(lam.values).to(dev)
reps = 5
bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(reps + 1)]
torch.cuda.synchronize()
print("first", s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr()), flush=True)
s.set_option("profile", 1); s.profile(reset=True)
t0 = time.perf_counter()
for r in bufs[1:]:
    s.factor_solve_device_async(vals.data_ptr(), r.data_ptr())
ok = s.sync()
dt = (time.perf_counter() - t0) / reps
print("ok", ok, "ms/solve %.3f" % (dt * 1e3))
for k, (c, ms) in s.profile().items():
    print("  %-14s %8.3f ms" % (k, ms / max(c, 1)))
x = bufs[-1].cpu().numpy()
A = lam.to_scipy()
print("resid", np.abs(A @ x - lam.rhs).max() / np.abs(lam.rhs).max())
