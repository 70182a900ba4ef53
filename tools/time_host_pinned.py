"""Dev helper: the host-array entry point (values uploaded on every call) with pageable against pinned host memory."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP, CLinearSolver_HIP
for name, lam, cls in (("C4", synth.ba(1000, 500000, k=4, mode="band"), CLinearSolver_Schur_HIP), ("C3", synth.pose_chain(), CLinearSolver_HIP)):
    s = cls()
    x = lam.rhs.copy(); assert s.Solve_PosDef(lam, x)
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            tv = torch.empty(lam.values.shape[0], dtype=torch.float64).pin_memory(); tv.numpy()[:] = lam.values
            tr = torch.empty(lam.rhs.shape[0], dtype=torch.float64).pin_memory()
            lam2 = synth.BlockSystem(lam.cumsum, lam.bcol_ptr, lam.brow_idx, tv.numpy(), lam.rhs, lam.n_matrix_cut)
            xbuf = tr.numpy()
        else:
            lam2, xbuf = lam, np.empty_like(lam.rhs)
        ts = []
        for i in range(6):
            xbuf[:] = lam.rhs
            t0 = time.perf_counter(); ok = s.Solve_PosDef_Blocky(lam2, xbuf); ts.append(time.perf_counter() - t0)
        print(name, kind, "ms per call", np.round(np.array(ts[1:]) * 1e3, 2), s.times.as_dict())
