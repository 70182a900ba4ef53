"""Dev helper: C1 / C2 over the dense-top threshold and the dissection's balance (the plan's own choice first)."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
dev = torch.device("cuda:0")


def run(lam, reps=20, **opts):
    s = CLinearSolver_HIP(**opts)
    s.SymbolicDecomposition_Blocky(lam)
    vals = torch.from_numpy(lam.values).to(dev)
    bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(reps + 1)]
    torch.cuda.synchronize()
    assert s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
    t0 = time.perf_counter()
    for b in bufs[1:]:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync()
    st = s.stats()
    return (time.perf_counter() - t0) / reps * 1e3, st.get("n_stages"), st.get("dense_dim", None)


for name, lam in (("C1", synth.manhattan(3500)), ("C2", synth.sphere(50, 50))):
    print(name, "plan's choice: %.3f ms" % run(lam)[0], flush=True)
    grid = [(nb, bal) for nb in (12, 16, 24, 36, 48) for bal in (15, 25, 35, 45)]
    if os.environ.get("SWEEP_FINE"):
        grid = [(nb, bal) for nb in (20, 24, 30, 36, 42) for bal in (40, 43, 45, 47, 49)]
    for nb, bal in grid:
        if True:
            t, n_st, dd = run(lam, dense_top_nb=nb, nd_balance=bal)
            print(f"  {name} dense_top_nb={nb:3d} nd_balance={bal:2d}: {t:.3f} ms  stages {n_st}", flush=True)
