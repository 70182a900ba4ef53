"""Dev helper (round 4): an option of the panel tasks on (1) and off (0) -- AB_OPTION=panel_rows (the level walk as rows; the
default) or panel_handup (tasks hand their contributions up; takes effect at analysis: a solver per setting) -- on C3, C2,
C1 and the reduced camera systems of the BA legs, same process, alternating."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
dev = torch.device("cuda:0")


OPTION = os.environ.get("AB_OPTION", "panel_rows")
VALUES = [int(v) for v in os.environ.get("AB_VALUES", "0,1,0,1").split(",")]  # e.g. AB_OPTION=task_height AB_VALUES=4,5,6,4,5,6


def run(name, lam, cls, reps=20):
    vals = torch.from_numpy(lam.values).to(dev)
    out = {}
    solvers = {}
    for rows in VALUES:
        if rows not in solvers:
            solvers[rows] = cls(**{OPTION: rows})
            solvers[rows].SymbolicDecomposition_Blocky(lam)
        s = solvers[rows]
        s.set_option("profile", 0)
        bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(2 * reps + 1)]
        torch.cuda.synchronize()
        assert s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
        t0 = time.perf_counter()
        for b in bufs[1:reps + 1]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync()
        dt = (time.perf_counter() - t0) / reps
        s.set_option("profile", 2 if cls is CLinearSolver_HIP else 1)
        s.profile(reset=True)
        for b in bufs[reps + 1:]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync()
        x = bufs[-1].cpu().numpy()
        res = np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max()
        print(f"{name:14s} {OPTION}={rows}  {dt*1e3:.3f} ms  resid {res:.1e}  " +
              "  ".join(f"{k}={ms/max(c,1)*1e3:.0f}" for k, (c, ms) in s.profile().items()), flush=True)


which = sys.argv[1:] or ["c3", "c2", "c1", "band", "c5", "venice"]
if "c3" in which:
    run("C3", synth.pose_chain(n=100000), CLinearSolver_HIP)
if "c2" in which:
    run("C2", synth.sphere(50, 50), CLinearSolver_HIP)
if "c1" in which:
    run("C1", synth.manhattan(3500), CLinearSolver_HIP)
if "band" in which:
    run("BA band 1kx500k", synth.ba(1000, 500000, k=4, mode="band", seed=777), CLinearSolver_Schur_HIP, reps=5)
if "c5" in which:
    run("BA C5 2kx2M", synth.ba(2000, 2000000, k=4, mode="band", seed=777), CLinearSolver_Schur_HIP, reps=5)
if "venice" in which:
    run("BA venice", synth.ba(1000, 500000, k=4, mode="venice", seed=777), CLinearSolver_Schur_HIP, reps=5)
