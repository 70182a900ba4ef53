"""Dev helper: options of the inner (sparse reduced system) solver on the C4 workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
lam = synth.ba(1000, 500000, k=4, mode="band")
vals = torch.from_numpy(lam.values).cuda()
for opts in [{}, {"simt": 0}, {"simt": 0, "subtree_size": 8}, {"simt_width": 16}, {"simt_width": 64}, {"subtree_size": 2}, {"subtree_size": 2, "simt": 0},
             {"leaf_size": 2}, {"leaf_size": 8}, {"wide_min_tasks": 64}, {"wide_min_tasks": 64, "simt": 0}]:
    s = CLinearSolver_Schur_HIP(**opts)
    s.SymbolicDecomposition_Blocky(lam)
    bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(6)]
    torch.cuda.synchronize()
    s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
    s.set_option("profile", 1); s.profile(reset=True)
    for b in bufs[1:]:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync()
    pr = s.profile()
    print(f"{opts}: reduced_sparse {pr['reduced_sparse'][1] / pr['reduced_sparse'][0]:.3f} ms", flush=True)
