"""Dev helper: block-diagonal covariances of the C3 pose graph (factorization + sparse inverse subset + extraction)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lam = synth.pose_chain(n=n)
s = CLinearSolver_HIP(profile=1)
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).cuda()
out = torch.empty(n * 36, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
s._check(s._lib.slampp_hip_marginals_device_async(s._h, vals.data_ptr(), out.data_ptr())); s.sync()
print("first call (lists built) %.1f ms" % ((time.perf_counter() - t0) * 1e3))
s.profile(reset=True)
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    s._check(s._lib.slampp_hip_marginals_device_async(s._h, vals.data_ptr(), out.data_ptr()))
s.sync()
print("ms per call %.3f" % ((time.perf_counter() - t0) / reps * 1e3), {k: round(v[1] / max(v[0], 1), 3) for k, v in s.profile().items()})
