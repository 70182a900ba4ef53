import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
for d in (3, 7):
    lam = synth.pose_chain(n=300000, d=d)
    s = CLinearSolver_HIP()
    x = lam.rhs.copy()
    t0 = time.perf_counter(); ok = s.Solve_PosDef(lam, x); t = time.perf_counter() - t0
    x2 = lam.rhs.copy(); t0 = time.perf_counter(); s.Solve_PosDef_Blocky(lam, x2); t2 = time.perf_counter() - t0
    print("chain d=%d ok=%s cold %.1f ms warm(host path) %.1f ms resid %.2e" % (d, ok, t * 1e3, t2 * 1e3,
          np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max()), flush=True)
lam = synth.ba(2000, 2000000, k=4, mode="band")
s = CLinearSolver_Schur_HIP(profile=1)
x = lam.rhs.copy()
t0 = time.perf_counter(); ok = s.Solve_PosDef(lam, x); t = time.perf_counter() - t0
print("C5-size solve ok", ok, "cold %.1f ms" % (t * 1e3), "resid %.2e" % (np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max()), flush=True)
t0 = time.perf_counter(); cams, pts = s.Schur_Marginals(lam); t = time.perf_counter() - t0
print("C5-size marginals %.1f ms (host path)" % (t * 1e3), {k: round(v[1] / max(v[0], 1), 3) for k, v in s.profile().items() if k.startswith("marg")}, flush=True)
e = np.zeros(lam.n_scalars); e[12000 + 3 * 777] = 1.0
assert s.Solve_PosDef_Blocky(lam, e)
print("col check", np.abs(e[12000 + 3 * 777:12000 + 3 * 777 + 3] - pts[777][:, 0]).max() / np.abs(pts[777][:, 0]).max(), "device MB", s.stats()["device_bytes"] / 1e6)
