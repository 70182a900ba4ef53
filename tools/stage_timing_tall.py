import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
lam = synth.pose_chain(n=100000)
s = CLinearSolver_HIP(task_height=3, wide_min_tasks=1 << 30)
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).cuda()
bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(3)]
for b in bufs:
    s.factor_solve_device(vals.data_ptr(), b.data_ptr())
