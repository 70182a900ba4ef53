"""Dev helper: device time of the Lambda / eta assembly on a C3-sized SE(3) graph (100k poses), against the
bytes it has to move, and the same edge set through the reference-like CPU oracle for scale."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLambdaAssembly_HIP
from oracle import oracle_lib as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = rd = 6
rng = np.random.default_rng(1)
n_loops = n // 50
a = rng.integers(60, n, n_loops); b = a - rng.integers(26, 60, n_loops)
v0 = np.concatenate([np.arange(n - 1), b]).astype(np.int64)
v1 = np.concatenate([np.arange(1, n), a]).astype(np.int64)
dims = np.full(n, d)
es = synth.random_edge_set(dims, v0, v1, rd=rd, seed=2)
lam = synth.structure_from_edges(dims, v0, v1)
t0 = time.perf_counter(); ref_v, ref_eta = O.assemble_lambda(lam, es); t_cpu = time.perf_counter() - t0
solver = CLinearSolver_HIP(profile=1)
t0 = time.perf_counter(); asm = CLambdaAssembly_HIP(solver, lam, v0, v1, rd); t_setup = time.perf_counter() - t0
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
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    asm.Refresh_Lambda_device(*args)
solver.sync()
wall = (time.perf_counter() - t0) / reps
cnt, ms = solver.profile()["assemble"]
ne = es.n_edges
in_bytes = ne * 8 * (rd * d * 2 + rd * rd + rd + 1)
out_bytes = 8 * (lam.values.shape[0] + lam.n_scalars)
print(f"edges={ne} setup={t_setup*1e3:.1f}ms cpu_oracle={t_cpu*1e3:.1f}ms device={ms/cnt*1e3:.1f}us wall={wall*1e6:.1f}us "
      f"err_values={err_v:.1e} err_eta={err_e:.1e} algorithmic={(in_bytes+out_bytes)/1e6:.1f}MB "
      f"-> {(in_bytes+out_bytes)/(ms/cnt*1e-3)/1e9:.0f} GB/s")
lam.values, lam.rhs = ref_v, ref_eta
ok = solver.factor_solve_device(values.data_ptr(), eta.data_ptr())
print("assembled system factor+solve ok:", ok, solver.times.as_dict()["total_ms"], "ms")
