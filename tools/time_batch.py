"""Dev helper: K value sets of the C3 structure in one pass of launches (slampp_hip_factor_solve_batch_device_async);
under `rocprofv3 --kernel-trace --stats` this is the per-kernel picture of a batch (profiles/r05_c3_batch8_kernel_stats.csv)."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lam = synth.pose_chain(n=int(os.environ.get("POSES", "100000")))
s = CLinearSolver_HIP()
s.SymbolicDecomposition_Blocky(lam)
n_v, n_s = lam.values.shape[0] + lam.values.shape[0] % 2, lam.n_scalars + lam.n_scalars % 2
vals = torch.zeros(K * n_v, dtype=torch.float64, device="cuda")
for k in range(K):
    vals[k * n_v:k * n_v + lam.values.shape[0]] = torch.from_numpy(lam.values).cuda()
rhs = [torch.zeros(K * n_s, dtype=torch.float64, device="cuda") for _ in range(steps + 1)]
for r in rhs:
    for k in range(K):
        r[k * n_s:k * n_s + lam.n_scalars] = torch.from_numpy(lam.rhs).cuda()
torch.cuda.synchronize()
s.factor_solve_batch_device_async(K, vals.data_ptr(), n_v, rhs[0].data_ptr(), n_s)
print("warm-up", s.sync_batch(K))
t0 = time.perf_counter()
for i in range(1, steps + 1):
    s.factor_solve_batch_device_async(K, vals.data_ptr(), n_v, rhs[i].data_ptr(), n_s)
ok = s.sync_batch(K)
dt = (time.perf_counter() - t0) / steps
x = rhs[-1][:lam.n_scalars].cpu().numpy()
print(f"K={K}: {dt * 1e3:.3f} ms per round, {dt * 1e3 / K:.3f} ms per system, ok {all(ok)}, "
      f"resid {np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max():.2e}")
