"""Dev tool: the C3 step for the leaf kernel's tasks per wave (option simt_width) -- with SLAMPP_HIP_SIMT_PAIRS / SLAMPP_HIP_SIMT_LPT
read once per process, run once per setting."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
lam = synth.pose_chain()
dev = torch.device("cuda:0")
v = torch.from_numpy(lam.values).to(dev)
for width in [int(a) for a in (sys.argv[1:] or ["32", "16", "64"])]:
    s = CLinearSolver_HIP(device=0, simt_width=width)
    s.SymbolicDecomposition_Blocky(lam)
    bs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(30)]
    for b in bs[:5]:
        s.factor_solve_device_async(v.data_ptr(), b.data_ptr())
    s.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in bs[5:]:
        s.factor_solve_device_async(v.data_ptr(), b.data_ptr())
    ok = s.sync()
    dt = (time.perf_counter() - t0) / 25
    s.set_option("profile", 1)
    b = torch.from_numpy(lam.rhs).to(dev)
    s.factor_solve_device(v.data_ptr(), b.data_ptr())
    print("simt_width", width, "ok", ok, "step %.3f ms" % (dt * 1e3), {k: round(x[1], 3) for k, x in s.profile().items()}, flush=True)
