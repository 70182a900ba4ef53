"""Dev helper: balance constraint of the nested-dissection separators."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
for name, lam in [("C3", synth.pose_chain(n=100000)), ("10k", synth.pose_chain(n=10000)), ("C1", synth.manhattan(3500)), ("C2", synth.sphere(50, 50))]:
    vals = torch.from_numpy(lam.values).cuda()
    for pct in (10, 15, 20, 25, 30):
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
        print(f"{name} balance>={pct}%: stages={st['n_stages']} l_nnz={st['l_nnz']} height={st['etree_height']} warm={(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
