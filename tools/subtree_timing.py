"""Dev helper: in-kernel clock samples of the bottom-stage kernel (SLAMPP_HIP_STAGE_TIMING=1): a workgroup in the
middle of the grid samples after its column records, block records, Lambda blocks, then after the diagonal and the
sub-diagonal blocks of every column."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, os
os.environ["SLAMPP_HIP_STAGE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lam = synth.pose_chain(n=n)
s = CLinearSolver_HIP()
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).cuda()
for i in range(2):
    rhs = torch.from_numpy(lam.rhs).cuda()
    torch.cuda.synchronize()
    s.factor_solve_device_async(vals.data_ptr(), rhs.data_ptr())
    s.sync()
