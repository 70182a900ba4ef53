"""Dev tool: phases of a Schur solve that updates the reduced system from a list of changed landmarks, against the rebuild
(option schur_incremental), on the bench's C4 legs.  Usage: python tools/incremental_phases.py [venice|band|uniform] [share]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP

mode = sys.argv[1] if len(sys.argv) > 1 else "venice"
share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
lam = synth.ba(1000, 500_000, mode=mode)
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
nc, n_pts = lam.n_matrix_cut, lam.n_bcols - lam.n_matrix_cut
points = np.sort(rng.choice(n_pts, size=max(int(share * n_pts), 1), replace=False))
off = lam.block_value_offsets()
vals2 = lam.values.copy()
for p_ in points:
    k0, k1 = int(lam.bcol_ptr[nc + p_]), int(lam.bcol_ptr[nc + p_ + 1])
    vals2[off[k0]:off[k1 - 1]] *= 0.9
    vals2[off[k1 - 1]:off[k1]] += 0.5 * np.eye(3).ravel()
solver = CLinearSolver_Schur_HIP(device=0, schur_incremental=2)
solver.SymbolicDecomposition_Blocky(lam)
v1, v2 = torch.from_numpy(lam.values).to(dev), torch.from_numpy(vals2).to(dev)
rhs = torch.from_numpy(lam.rhs).to(dev)
for use_list in (False, True):
    solver.set_option("profile", 1)
    for _ in range(3):
        b1, b2 = rhs.clone(), rhs.clone()
        solver.set_option("profile", 0)
        solver.factor_solve_device(v1.data_ptr(), b1.data_ptr())
        if use_list:
            solver.Set_Changed_Landmarks(points)
        torch.cuda.synchronize()
        solver.set_option("profile", 1)
        t0 = time.perf_counter()
        solver.factor_solve_device(v2.data_ptr(), b2.data_ptr())
        ms = (time.perf_counter() - t0) * 1e3
    print("update" if use_list else "rebuild", "%.3f ms (last of 3 solves; phases summed over them)" % ms, {k: (c, round(v, 3)) for k, (c, v) in solver.profile(reset=True).items()})
