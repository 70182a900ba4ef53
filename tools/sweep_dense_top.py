"""Dev helper: sweep the dense-top threshold on the C1 / C2 look-alikes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
dev = torch.device("cuda:0")
for name, lam in [("C1", synth.manhattan(3500)), ("C2", synth.sphere(50, 50)), ("grid100x100", synth.sphere(100, 100))]:
    for nb, tiles in ((12, 1), (16, 1), (24, 0), (24, 1), (32, 1), (48, 1)):
        s = CLinearSolver_HIP(dense_top_nb=nb, dense_top_tiles=tiles)
        s.SymbolicDecomposition_Blocky(lam)
        st = s.stats()
        vals = torch.from_numpy(lam.values).to(dev)
        bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(11)]
        torch.cuda.synchronize()
        s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
        s.set_option("profile", 1); s.profile(reset=True)
        t0 = time.perf_counter()
        for b in bufs[1:]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync(); dt = (time.perf_counter() - t0) / 10 * 1e3
        pr = {k: round(v[1] / max(v[0], 1), 3) for k, v in s.profile().items()}
        print(f"{name} nb>={nb:3d} tiles={tiles}: dense_dim={st['schur_dim']:5d} stages={st['n_stages']:3d} warm={dt:7.3f} ms  {pr}", flush=True)
