"""Dev helper: wall-clock of the sparse path on the 100k-pose SE(3) system for a few schedule knobs."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lam = synth.pose_chain(n=n)
A = lam.to_scipy()
dev = torch.device("cuda:0")
vals = torch.from_numpy(lam.values).to(dev)
for leaf, sub in [tuple(int(x) for x in a.split(",")) for a in (sys.argv[2:] or ["32,32", "16,16", "8,8", "4,4", "2,2", "1,1", "64,64", "4,16", "8,32"])]:
    s = CLinearSolver_HIP(leaf_size=leaf, subtree_size=sub)
    t0 = time.perf_counter(); s.SymbolicDecomposition_Blocky(lam); t_an = time.perf_counter() - t0
    st = s.stats()
    rhs = torch.from_numpy(lam.rhs).to(dev)
    torch.cuda.synchronize()
    ok = s.factor_solve_device(vals.data_ptr(), rhs.data_ptr())
    x = rhs.cpu().numpy()
    res = np.abs(A @ x - lam.rhs).max() / np.abs(lam.rhs).max()
    reps = 20
    rhs_list = [torch.from_numpy(lam.rhs).to(dev) for _ in range(reps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in rhs_list:
        s.factor_solve_device_async(vals.data_ptr(), r.data_ptr())
    s.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"leaf={leaf} sub={sub} analyze={t_an*1e3:.1f}ms ok={ok} resid={res:.2e} stages={st['n_stages']} tasks={st['n_tasks']} "
          f"lnz={st['l_nnz']} flops={st['factor_flops']:.3g} factor+solve={dt*1e3:.3f}ms", flush=True)
