"""Dev helper: what an analysis costs on small systems in the caller's order (what FastL's Factorize_PosDef_Blocky pays per
call when the part of R it hands over changed): SLAMPP_HIP_PLAN_TIMING=1 prints the phases to stderr."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP

for n in (20, 100, 600, 2000):
    s = CLinearSolver_HIP(natural_order=1)
    for rep in range(4):
        lam = synth.pose_chain(n=n + rep, d=6, loop_every=7, loop_min=3, loop_max=6, seed=rep)   # a new structure every call
        t0 = time.perf_counter()
        s.SymbolicDecomposition_Blocky(lam)
        t1 = time.perf_counter()
        f = s.factorize(lam)
        t2 = time.perf_counter()
        print(f"n={n + rep}: analyze {1e3 * (t1 - t0):.2f} ms, factorize {1e3 * (t2 - t1):.2f} ms", flush=True)
        print(f"--- n={n + rep}", file=sys.stderr, flush=True)
