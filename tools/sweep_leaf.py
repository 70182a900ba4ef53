"""Dev helper: leaf / subtree sizes on C3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
lam = synth.pose_chain(n=100000)
vals = torch.from_numpy(lam.values).cuda()
for leaf in (2, 4, 8):
    for sub in (4, 8, 16):
        for bal in (10, 15):
            s = CLinearSolver_HIP(leaf_size=leaf, subtree_size=sub, nd_balance=bal)
            s.SymbolicDecomposition_Blocky(lam)
            st = s.stats()
            bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(21)]
            torch.cuda.synchronize()
            s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
            t0 = time.perf_counter()
            for b in bufs[1:]:
                s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
            s.sync()
            print(f"leaf={leaf} sub={sub} bal={bal}: stages={st['n_stages']} l_nnz={st['l_nnz']} warm={(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
