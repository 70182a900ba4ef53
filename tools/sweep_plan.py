"""Dev helper: plan knobs (balance of the dissection, leaf / subtree size) against the C3 step time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
lam = synth.pose_chain(n=100000)
vals = torch.from_numpy(lam.values).cuda()
for opts in [{}, {"nd_balance": 25}, {"nd_balance": 35}, {"nd_balance": 45}, {"nd_balance": 10}, {"subtree_size": 16}, {"subtree_size": 32},
             {"nd_balance": 35, "subtree_size": 16}, {"leaf_size": 8}, {"leaf_size": 2}, {"wide_min_tasks": 4096}, {"wide_min_tasks": 256}]:
    s = CLinearSolver_HIP(**opts)
    s.SymbolicDecomposition_Blocky(lam)
    st = s.stats()
    bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(21)]
    torch.cuda.synchronize()
    s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
    t0 = time.perf_counter()
    for b in bufs[1:]:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync(); dt = (time.perf_counter() - t0) / 20
    print(f"{opts}: {dt * 1e3:.3f} ms  stages {st['n_stages']} bottom {st['n_bottom_stages']} tasks {st['n_tasks']} l_blocks {st['l_blocks']}", flush=True)
