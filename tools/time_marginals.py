"""Dev helper: phase timing of the Schur marginals (block diagonal of the covariance) at bench sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
mode = sys.argv[3] if len(sys.argv) > 3 else "band"
lam = synth.ba(nc, npts, k=4, mode=mode)
dev = torch.device("cuda:0")
s = CLinearSolver_Schur_HIP()
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).to(dev)
cams = torch.empty(nc * 36, dtype=torch.float64, device=dev)
pts = torch.empty(npts * 9, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
s.schur_marginals_device_async(vals.data_ptr(), cams.data_ptr(), pts.data_ptr())
print("first", s.sync(), flush=True)
s.set_option("profile", 1); s.profile(reset=True)
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    s.schur_marginals_device_async(vals.data_ptr(), cams.data_ptr(), pts.data_ptr())
ok = s.sync()
print("ok", ok, "ms/call %.3f" % ((time.perf_counter() - t0) / reps * 1e3))
for k, (c, ms) in s.profile().items():
    print("  %-20s %8.3f ms" % (k, ms / max(c, 1)))
n = 6.0 * nc
print("inverse: %.1f TFLOP/s (2 n^3 / 3)" % (2 * n ** 3 / 3 / (s.profile()["marginals_inverse"][1] / reps * 1e-3) / 1e12))
# spot check against solves with unit vectors
c_np, p_np = cams.cpu().numpy().reshape(nc, 6, 6), pts.cpu().numpy().reshape(npts, 3, 3)
nx = 6 * nc
for (idx, blk, d, base) in ((7, c_np, 6, 0), (nc - 1, c_np, 6, 0), (12345 % npts, p_np, 3, nx), (npts - 1, p_np, 3, nx)):
    e = np.zeros(lam.n_scalars); e[base + d * idx] = 1.0
    assert s.Solve_PosDef_Blocky(lam, e)
    ref = e[base + d * idx: base + d * idx + d]
    print("block", idx, "rel err", np.abs(ref - blk[idx][:, 0]).max() / np.abs(ref).max())
