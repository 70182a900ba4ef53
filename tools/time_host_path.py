"""Dev helper: PCIe-inclusive timing of the host-array entry point (what the C++ header wrapper calls)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
for name, lam, cls in [("C3", synth.pose_chain(), CLinearSolver_HIP), ("C4", synth.ba(1000, 500000), CLinearSolver_Schur_HIP)]:
    s = cls()
    eta = lam.rhs.copy()
    t0 = time.perf_counter(); ok = s.Solve_PosDef(lam, eta); cold = (time.perf_counter() - t0) * 1e3
    t_cold = s.times.as_dict()
    warm = []
    for _ in range(5):
        eta = lam.rhs.copy()
        t0 = time.perf_counter(); s.Solve_PosDef_Blocky(lam, eta); warm.append((time.perf_counter() - t0) * 1e3)
    print(name, "cold %.1f ms" % cold, "warm median %.2f ms" % np.median(warm), {k: round(v, 3) for k, v in s.times.as_dict().items() if v}, "cold phases", {k: round(v, 2) for k, v in t_cold.items() if v})
