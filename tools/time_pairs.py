"""Dev helper: leaf kernel with one / two lanes per task (SLAMPP_HIP_SIMT_PAIRS is read once per process: run twice)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
for name, lam in (("C3", synth.pose_chain(n=100000)), ("1M", synth.pose_chain(n=1000000)), ("se2_100k", synth.pose_chain(n=100000, d=3))):
    vals = torch.from_numpy(lam.values).cuda()
    for width in (32, 16):
        s = CLinearSolver_HIP(simt_width=width)
        s.SymbolicDecomposition_Blocky(lam)
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
        print(f"pairs={os.environ.get('SLAMPP_HIP_SIMT_PAIRS', '1')} {name} width={width}: warm={dt:.3f} ms resid={res:.1e} " +
              "  ".join(f"{k}={ms / max(c, 1) * 1e3:.0f}us" for k, (c, ms) in s.profile().items()), flush=True)
