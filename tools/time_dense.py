"""Dev helper: BA with a dense reduced system (uniform visibility): phase times, dense factorization TFLOP/s."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
from oracle import oracle_lib as O
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
mode = sys.argv[3] if len(sys.argv) > 3 else "uniform"
lam = synth.ba(nc, npts, k=4, mode=mode)
dev = torch.device("cuda:0")
s = CLinearSolver_Schur_HIP(schur_sparse=0)
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).to(dev)
reps = 5
bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(reps + 1)]
torch.cuda.synchronize()
assert s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
s.set_option("profile", 1); s.profile(reset=True)
t0 = time.perf_counter()
for b in bufs[1:]:
    s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
ok = s.sync(); dt = (time.perf_counter() - t0) / reps
x = bufs[-1].cpu().numpy()
st = s.stats()
prof = {k: ms / max(c, 1) for k, (c, ms) in s.profile().items()}
tf = st["factor_flops"] / (prof["dense_chol"] * 1e-3) / 1e12 if "dense_chol" in prof else 0
res = np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max() if npts <= 200000 else -1
print(f"{nc} cams x {npts} pts {mode}: ok={ok} solve {dt*1e3:.3f} ms resid {res:.1e} dense_chol {prof.get('dense_chol', 0):.3f} ms = {tf:.1f} TFLOP/s  " +
      "  ".join(f"{k}={v*1e3:.0f}us" for k, v in prof.items()), flush=True)
