"""Dev helper: what the FIRST use of the library in a process costs beyond the work (code object load, first launches):
a tiny pose chain -- its analysis and solve are microseconds of work -- solved by a first and by a second handle."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()   # the runtime itself is up before the clock starts
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP

def once(tag, make, lam):
    t0 = time.perf_counter(); s = make(); t1 = time.perf_counter()
    s.SymbolicDecomposition_Blocky(lam); t2 = time.perf_counter()
    eta = lam.rhs.copy(); ok = s.Solve_PosDef_Blocky(lam, eta); t3 = time.perf_counter()
    eta = lam.rhs.copy(); ok = s.Solve_PosDef_Blocky(lam, eta) and ok; t4 = time.perf_counter()
    print("%-28s create %7.2f ms, analyze %7.2f, first solve %7.2f, second solve %7.2f  ok=%s" % (tag, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, ok), flush=True)

lam = synth.pose_chain(200, seed=1)
once("pose chain, first handle", CLinearSolver_HIP, lam)
once("pose chain, second handle", CLinearSolver_HIP, lam)
ba = synth.ba(20, 400, k=4, mode="band", seed=1)
once("BA, first handle", CLinearSolver_Schur_HIP, ba)
once("BA, second handle", CLinearSolver_Schur_HIP, ba)
