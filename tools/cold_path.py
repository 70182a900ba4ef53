"""Dev helper (bench_legs/cold.py is the form the bench itself uses): what the analysis (set_structure + analyze, what the first Solve_PosDef_Blocky pays before any arithmetic) costs
on the bench's workloads, a fresh handle every time; with SLAMPP_HIP_PLAN_TIMING=1 the phases on stderr.
usage: cold_path.py [c1 c2 c3 venice band c5 uniform 1kx1m]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP

CASES = {
    "c1": (CLinearSolver_HIP, lambda: synth.manhattan(3500)),
    "c2": (CLinearSolver_HIP, lambda: synth.sphere(50, 50)),
    "c3": (CLinearSolver_HIP, lambda: synth.pose_chain(n=100000)),
    "venice": (CLinearSolver_Schur_HIP, lambda: synth.ba(1000, 500000, mode="venice", seed=777)),
    "band": (CLinearSolver_Schur_HIP, lambda: synth.ba(1000, 500000, mode="band", seed=777)),
    "uniform": (CLinearSolver_Schur_HIP, lambda: synth.ba(1000, 500000, mode="uniform", seed=777)),
    "c5": (CLinearSolver_Schur_HIP, lambda: synth.ba(2000, 2000000, mode="band", seed=777)),
    "1kx1m": (CLinearSolver_Schur_HIP, lambda: synth.ba(1000, 1000000, mode="band", seed=777)),
}
reps = int(os.environ.get("REPS", "3"))
warm = CLinearSolver_HIP()      # the process's first handle pays the runtime's start-up, not the analysis
warm.SymbolicDecomposition_Blocky(synth.pose_chain(n=64))
for name in (sys.argv[1:] or ["c1", "c2", "c3", "venice", "c5", "uniform"]):
    cls, make = CASES[name]
    lam = make()
    times = []
    for rep in range(reps):
        s = cls()
        print(f"--- {name} rep {rep}", file=sys.stderr, flush=True)
        if os.environ.get("SETTLE_MS"):      # let the driver finish releasing the previous handle's memory first
            time.sleep(float(os.environ["SETTLE_MS"]) * 1e-3)
        t0 = time.perf_counter()
        s.SymbolicDecomposition_Blocky(lam)
        times.append((time.perf_counter() - t0) * 1e3)
        del s
    print(f"{name}: analyze_ms_cold " + " ".join(f"{t:.1f}" for t in times), flush=True)
