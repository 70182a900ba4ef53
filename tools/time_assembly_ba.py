"""Dev helper: device time of the Lambda / eta assembly of a BA system (C4 shape: 1000 cameras x 500k points, 4
observations per point, edges point -> camera with 2-d residuals), parity against the CPU oracle, then the Schur solve
on the assembled values without leaving the device."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP, CLambdaAssembly_HIP
from oracle import oracle_lib as O

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
k = 4
rng = np.random.default_rng(1)
c0 = rng.integers(0, nc, npts)
cams = (c0[:, None] + 7 * np.arange(k)[None, :]) % nc            # the band visibility of synth.ba
v0 = (nc + np.repeat(np.arange(npts), k)).astype(np.int64)       # vertex 0 = the point (EDGE_P2MC xyz_id cam_id)
v1 = cams.reshape(-1).astype(np.int64)
dims = np.concatenate([np.full(nc, 6), np.full(npts, 3)])
es = synth.random_edge_set(dims, v0, v1, rd=2, seed=2, anchor=0)
lam = synth.structure_from_edges(dims, v0, v1)
lam.n_matrix_cut = nc
t0 = time.perf_counter(); ref_v, ref_eta = O.assemble_lambda(lam, es); t_cpu = time.perf_counter() - t0
# the 2-d residuals leave the 3-d points' blocks rank-2 per observation: a little damping, as LM would add
solver = CLinearSolver_Schur_HIP(profile=1)
t0 = time.perf_counter(); asm = CLambdaAssembly_HIP(solver, lam, v0, v1, 2); t_setup = time.perf_counter() - t0
dv = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
J0, J1, S, E, W = dv(es.J0), dv(es.J1), dv(es.sigma_inv), dv(es.err), dv(es.weight)
values = torch.zeros(lam.values.shape[0], dtype=torch.float64, device="cuda")
eta = torch.zeros(lam.n_scalars, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
args = (J0.data_ptr(), J1.data_ptr(), S.data_ptr(), E.data_ptr(), W.data_ptr(), values.data_ptr(), eta.data_ptr(), 0,
        es.unary_factor, es.unary_error)
asm.Refresh_Lambda_device(*args); solver.sync()
err_v = np.abs(values.cpu().numpy() - ref_v).max() / np.abs(ref_v).max()
err_e = np.abs(eta.cpu().numpy() - ref_eta).max() / np.abs(ref_eta).max()
solver.profile(reset=True)
reps = 10
for _ in range(reps):
    asm.Refresh_Lambda_device(*args)
solver.sync()
cnt, ms = solver.profile()["assemble"]
ne = es.n_edges
in_bytes = ne * 8 * (2 * 3 + 2 * 6 + 4 + 2 + 1)
out_bytes = 8 * (lam.values.shape[0] + lam.n_scalars)
print(f"edges={ne} setup={t_setup*1e3:.1f}ms cpu_oracle={t_cpu*1e3:.1f}ms device={ms/cnt*1e3:.1f}us "
      f"err_values={err_v:.1e} err_eta={err_e:.1e} algorithmic={(in_bytes+out_bytes)/1e6:.1f}MB "
      f"-> {(in_bytes+out_bytes)/(ms/cnt*1e-3)/1e9:.0f} GB/s", flush=True)
x = eta.clone()
ok = solver.factor_solve_device(values.data_ptr(), x.data_ptr())
lam.values, lam.rhs = ref_v, ref_eta
xs = x.cpu().numpy()
print("assembled system Schur solve ok:", ok, "resid", np.abs(lam.to_scipy() @ xs - ref_eta).max() / np.abs(ref_eta).max())
