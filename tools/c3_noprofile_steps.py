"""Dev tool: ten back-to-back C3 solves with the phase events off (option profile = 0), for a kernel trace of a step without them."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
lam = synth.pose_chain()
dev = torch.device("cuda:0")
solver = CLinearSolver_HIP(device=0)
solver.SymbolicDecomposition_Blocky(lam)
solver.set_option("profile", 0)
v = torch.from_numpy(lam.values).to(dev)
bs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(10)]
torch.cuda.synchronize()
for b in bs:
    solver.factor_solve_device_async(v.data_ptr(), b.data_ptr())
print("ok", solver.sync())
