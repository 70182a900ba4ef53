"""Dev helper: in-kernel clock samples of the upper-stage factor kernel (set SLAMPP_HIP_STAGE_TIMING=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
lam = synth.pose_chain(n=100000)
s = CLinearSolver_HIP()
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).cuda()
for i in range(3):
    rhs = torch.from_numpy(lam.rhs).cuda()
    torch.cuda.synchronize()
    s.factor_solve_device_async(vals.data_ptr(), rhs.data_ptr())
    if i < 2:
        os.environ.pop("X", None)
    s.sync()
