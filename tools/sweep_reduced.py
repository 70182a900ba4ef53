"""Dev helper: leaf / subtree size of the inner (sparse reduced system) solver on the C4 workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
lam = synth.ba(1000, 500000, k=4, mode="band")
vals = torch.from_numpy(lam.values).cuda()
for leaf, sub in [(4, 16), (2, 8), (8, 32), (4, 64), (16, 16), (1, 4), (4, 4), (32, 64)]:
    s = CLinearSolver_Schur_HIP(leaf_size=leaf, subtree_size=sub)
    s.SymbolicDecomposition_Blocky(lam)
    bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(6)]
    torch.cuda.synchronize()
    s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
    s.set_option("profile", 1); s.profile(reset=True)
    for b in bufs[1:]:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync()
    pr = s.profile()
    print(f"leaf={leaf} sub={sub}: reduced_sparse {pr['reduced_sparse'][1] / pr['reduced_sparse'][0]:.3f} ms", flush=True)
