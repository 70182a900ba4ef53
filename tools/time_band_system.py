"""Dev helper: the sparse block path on a matrix shaped like C4's reduced camera system (1000 block columns of 6, blocks at
circular distances 7, 14, 21): plan summary, per-kernel trace is left to rocprofv3."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 6
rng = np.random.default_rng(5)
pairs = set()
for c in range(n):
    for j in (7, 14, 21):
        a, b = sorted((c, (c + j) % n))
        if a != b:
            pairs.add((a, b))
cols = [[] for _ in range(n)]
for a, b in pairs:
    cols[b].append(a)
bcol_ptr = [0]; brow = []
for c in range(n):
    rows = sorted(cols[c]) + [c]
    brow += rows; bcol_ptr.append(len(brow))
nb = len(brow)
vals = rng.standard_normal((nb, d, d)) * 0.1
brow = np.array(brow, dtype=np.int32); bcol_ptr = np.array(bcol_ptr, dtype=np.int64)
col_of = np.repeat(np.arange(n), np.diff(bcol_ptr))
diag = brow == col_of
vals[diag] = np.eye(d) * 4.0 + 0.01 * (vals[diag] + vals[diag].transpose(0, 2, 1))
lam = synth.BlockSystem(np.arange(n + 1, dtype=np.int64) * d, bcol_ptr, brow, vals.transpose(0, 2, 1).reshape(-1).copy(), rng.standard_normal(n * d), 0)
opts = {}
for a in sys.argv[2:]:
    k, v = a.split("="); opts[k] = int(v)
s = CLinearSolver_HIP(**opts)
s.SymbolicDecomposition_Blocky(lam)
p = s.plan()
st = s.stats()
print({k: st[k] for k in ("n_stages", "n_tasks", "l_blocks", "n_update_pairs", "etree_height", "n_bottom_stages")})
sp = np.asarray(p["stage_ptr"]); tp = np.asarray(p["task_ptr"])
for i in range(len(sp) - 1):
    t0, t1 = sp[i], sp[i + 1]
    ncols = tp[t1] - tp[t0]
    print(f"stage {i}: {t1 - t0} tasks, {ncols} columns, longest task {int((tp[t0 + 1:t1 + 1] - tp[t0:t1]).max())} columns")
dev = torch.device("cuda:0")
v = torch.from_numpy(lam.values).to(dev)
bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(11)]
torch.cuda.synchronize()
assert s.factor_solve_device(v.data_ptr(), bufs[0].data_ptr())
s.set_option("profile", 2); s.profile(reset=True)
t0 = time.perf_counter()
for b in bufs[1:]:
    s.factor_solve_device_async(v.data_ptr(), b.data_ptr())
s.sync(); dt = (time.perf_counter() - t0) / 10
x = bufs[-1].cpu().numpy()
print(f"solve {dt * 1e3:.3f} ms resid {np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max():.1e}  " + "  ".join(f"{k}={ms / max(c, 1) * 1e3:.0f}us" for k, (c, ms) in s.profile().items()))
