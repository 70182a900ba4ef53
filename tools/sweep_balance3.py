"""Dev helper: balance of the nested dissection where the plan has a dense top (tile levels are what its chain is made of)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP, CLinearSolver_HIP
ven = synth.ba(1000, 500_000, mode="venice", seed=777)
for name, lam, cls in (("venice", ven, CLinearSolver_Schur_HIP), ("C2", synth.sphere(50, 50), CLinearSolver_HIP), ("C1", synth.manhattan(3500), CLinearSolver_HIP),
                       ("grid100", None, None)):
    if lam is None:
        continue
    vals = torch.from_numpy(lam.values).cuda()
    for pct in (15, 25, 30, 35, 40, 45, 49):
        for nb in (None, 16, 36):
            opts = {"nd_balance": pct}
            if nb:
                opts["dense_top_nb"] = nb
            s = cls(**opts)
            t0 = time.perf_counter(); s.SymbolicDecomposition_Blocky(lam); ana = (time.perf_counter() - t0) * 1e3
            bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(11)]
            torch.cuda.synchronize()
            ok = s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
            t0 = time.perf_counter()
            for b in bufs[1:]:
                s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
            s.sync()
            dt = (time.perf_counter() - t0) / 10 * 1e3
            st = s.reduced_stats() if name == "venice" else s.stats()
            print(f"{name} balance {pct} nb {nb}: step {dt:.3f} ms analyze {ana:.0f} ms dense_dim {st['schur_dim']} stages {st['n_stages']} ok {ok}", flush=True)
