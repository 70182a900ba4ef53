"""Dev helper: phase times of the cold path (ordering, symbolic, records, uploads) on the GPU box's host."""
import sys, os, time
os.environ["SLAMPP_HIP_PLAN_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
lam = synth.pose_chain(n=100000)
s = CLinearSolver_HIP()
s.SymbolicDecomposition_Blocky(synth.pose_chain(n=1000))   # library and device warm
for i in range(2):
    s.Clear_SymbolicDecomposition()
    t0 = time.perf_counter(); s.SymbolicDecomposition_Blocky(lam); print("analyze total %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
