"""Dev helper: ND balance constraint on pose chains of several sizes / seeds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
for n, seed in [(100000, 1), (100000, 2), (30000, 3), (300000, 4)]:
    lam = synth.pose_chain(n=n, seed=seed)
    vals = torch.from_numpy(lam.values).cuda()
    for pct in (10, 15, 20, 25):
        s = CLinearSolver_HIP(nd_balance=pct)
        s.SymbolicDecomposition_Blocky(lam)
        st = s.stats()
        bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(21)]
        torch.cuda.synchronize()
        s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
        t0 = time.perf_counter()
        for b in bufs[1:]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync()
        print(f"n={n} seed={seed} balance>={pct}%: stages={st['n_stages']} l_nnz={st['l_nnz']} warm={(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
