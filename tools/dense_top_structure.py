"""Dev helper: tile structure / level count of the dense top (prints the [setup] lines of the library)."""
import sys, os
os.environ["SLAMPP_HIP_PLAN_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
for name, lam in [("C1", synth.manhattan(3500)), ("C2", synth.sphere(50, 50)), ("grid100x100", synth.sphere(100, 100))]:
    print(name, file=sys.stderr, flush=True)
    s = CLinearSolver_HIP(dense_top_tiles=1)
    s.SymbolicDecomposition_Blocky(lam)
