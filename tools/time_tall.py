"""Dev helper: separator tasks as slices of the elimination tree (option task_height) against one level per stage."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
which = sys.argv[1:] or ["C3"]
systems = {"C3": lambda: synth.pose_chain(n=100000), "10k": lambda: synth.pose_chain(n=10000), "1M": lambda: synth.pose_chain(n=1000000),
           "C1": lambda: synth.manhattan(3500), "C2": lambda: synth.sphere(50, 50)}
for name in which:
    lam = systems[name]()
    vals = torch.from_numpy(lam.values).cuda()
    x_ref = None
    for h, wide in ((1, 1024), (3, 1024), (3, 1 << 30), (2, 1 << 30)):
        s = CLinearSolver_HIP(task_height=h, wide_min_tasks=wide)
        s.SymbolicDecomposition_Blocky(lam)
        st = s.stats()
        bufs = [torch.from_numpy(lam.rhs).cuda() for _ in range(41)]
        torch.cuda.synchronize()
        assert s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
        t0 = time.perf_counter()
        for b in bufs[1:21]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        s.set_option("profile", 2); s.profile(reset=True)
        for b in bufs[21:]:
            s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        s.sync()
        x = bufs[-1].cpu().numpy()
        res = np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max()
        if x_ref is None:
            x_ref = x
        print(f"{name} task_height={h} wide_min={wide}: stages={st['n_stages']} tasks={st['n_tasks']} warm={dt:.3f} ms resid={res:.1e} "
              f"vs h=1 {np.abs(x - x_ref).max() / np.abs(x_ref).max():.1e}  " +
              "  ".join(f"{k}={ms / max(c, 1) * 1e3:.0f}us" for k, (c, ms) in s.profile().items()), flush=True)
