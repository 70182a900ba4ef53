"""Dev helper: where the line between block-by-block elimination and the dense top is drawn (option dense_top_nb, fixed
balance) against what the plan's own choice gives, on the C1 / C2 look-alikes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
for name, lam in (("C2", synth.sphere(50, 50)), ("C1", synth.manhattan(3500))):
    vals = torch.from_numpy(lam.values).cuda()
    for opts in ({},) + tuple({"nd_balance": pct, "dense_top_nb": nb} for pct in (25, 35, 45) for nb in (8, 12, 16, 24, 36, 54)):
        s = CLinearSolver_HIP(**opts)
        s.SymbolicDecomposition_Blocky(lam)
        bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(21)]
        torch.cuda.synchronize()
        ok = s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
        t0 = time.perf_counter()
        for b in bufs[1:]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        st = s.stats()
        print(f"{name} {opts or 'auto'}: step {dt:.3f} ms dense_dim {st['schur_dim']} stages {st['n_stages']} ok {ok}", flush=True)
