"""Dev helper: one handle over several members (slampp_hip_create_multi) against the single handle, host entry points;
on a 1-GPU box the members share device 0 (no xGMI in these numbers: the bookkeeping, the events and the exchange code)."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
n_dev = torch.cuda.device_count()
for mode, opts in (("band", {}), ("uniform", {"schur_sparse": 0}), ("uniform", {"schur_sparse": 0, "schur_distributed": 1})):
    lam = synth.ba(1000, 500_000, k=4, mode=mode, seed=777)
    for members in (1, 2, 4):
        devices = [i % n_dev for i in range(members)]
        s = CLinearSolver_Schur_HIP(devices=devices, profile=1, **opts) if members > 1 else CLinearSolver_Schur_HIP(profile=1, **{k: v for k, v in opts.items() if k != "schur_distributed"})
        eta = lam.rhs.copy()
        t0 = time.perf_counter(); ok = s.Solve_PosDef(lam, eta); cold = (time.perf_counter() - t0) * 1e3
        warm = []
        for _ in range(4):
            eta = lam.rhs.copy()
            t0 = time.perf_counter(); ok = s.Solve_PosDef_Blocky(lam, eta) and ok; warm.append((time.perf_counter() - t0) * 1e3)
        res = np.abs(lam.to_scipy() @ eta - lam.rhs).max() / np.abs(lam.rhs).max()
        pr = {k: round(ms / max(c, 1), 3) for k, (c, ms) in s.profile().items()}
        print(f"{mode} {opts} members={members} devices={devices}: cold {cold:.1f} ms warm {min(warm):.2f} ms resid {res:.1e} {s.group_info().get('exchange')} phases(member 0) {pr}", flush=True)
