"""Dev helper: the covariances with the dense inverse of the reduced system forced (option marginals_dense), phase times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
lam = synth.ba(nc, npts, k=4, mode="band")
s = CLinearSolver_Schur_HIP(marginals_dense=1, profile=1)
s.SymbolicDecomposition_Blocky(lam)
vals = torch.from_numpy(lam.values).cuda()
cams = torch.empty(nc * 36, dtype=torch.float64, device="cuda"); pts = torch.empty(npts * 9, dtype=torch.float64, device="cuda")
for i in range(4):
    s.schur_marginals_device_async(vals.data_ptr(), cams.data_ptr(), pts.data_ptr()); s.sync()
    if i == 0: s.profile(reset=True)
p = {k: round(v[1] / max(v[0], 1), 3) for k, v in s.profile().items()}
print(p, "inverse TFLOP/s %.1f" % (2 * (6.0 * nc) ** 3 / 3 / (p["marginals_inverse"] * 1e-3) / 1e12))
