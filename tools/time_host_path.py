"""Dev helper: what a caller pays at C3 -- host arrays through slampp_hip_factor_solve (bench.host_path_leg), and, if the
compiled reference travelled, a CUberBlockMatrix through LinearSolver_HIP.h (bench.dropin_leg)."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP

lam = synth.pose_chain(n=100000)
r = bench.host_path_leg(lambda: CLinearSolver_HIP(device=0), lam, reps=15)
print("host arrays:", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "last_call_ms"}),
      {k: round(v, 3) for k, v in r["last_call_ms"].items()})
d = bench.dropin_leg(lam, reps=15)
if d:
    print("drop-in:", {k: d[k] for k in d if k.startswith("hip_")})
