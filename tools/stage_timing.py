"""Dev helper: in-kernel clock samples of the panel / upper-stage factor kernels (workgroup 0 of every launch) at C3.
usage: SLAMPP_HIP_STAGE_TIMING=1 python tools/stage_timing.py [option=value ...]"""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, os
os.environ.setdefault("SLAMPP_HIP_STAGE_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
lam = synth.pose_chain(n=100000)
s = CLinearSolver_HIP(**{a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:]})
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).cuda()
for i in range(3):
    rhs = torch.from_numpy(lam.rhs).cuda()
    torch.cuda.synchronize()
    s.factor_solve_device_async(vals.data_ptr(), rhs.data_ptr())
    s.sync()
