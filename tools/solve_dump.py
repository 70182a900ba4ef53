"""Solves a system matrix dumped by SLAM++ (`-dsm`: system.mtx + system.bla) on the GPU and reports the timing.

usage: python tools/solve_dump.py system.mtx system.bla [--cut N_CAMERAS] [--rhs rhs.txt] [--reps 10]

With --cut the Schur-complement solver is used (the first N block columns are cameras); the right-hand side defaults to
Lambda times a vector of ones (the dump does not contain eta), so that the error of the solution can be printed."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("mtx"); ap.add_argument("bla")
ap.add_argument("--cut", type=int, default=0)
ap.add_argument("--rhs", default=None)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
lam = synth.load_matrix_market(args.mtx, args.bla, n_matrix_cut=args.cut)
A = lam.to_scipy()
x_true = None
if args.rhs:
    lam.rhs = np.loadtxt(args.rhs).ravel()
else:
    x_true = np.ones(lam.n_scalars)
    lam.rhs = A @ x_true
solver = CLinearSolver_Schur_HIP() if args.cut else CLinearSolver_HIP()
eta = lam.rhs.copy()
t0 = time.perf_counter(); ok = solver.Solve_PosDef(lam, eta); cold = (time.perf_counter() - t0) * 1e3
print(f"{lam.n_bcols} block columns, {lam.n_blocks} upper blocks, n = {lam.n_scalars}; positive definite: {ok}")
if ok:
    warm = []
    for _ in range(args.reps):
        e = lam.rhs.copy()
        t0 = time.perf_counter(); solver.Solve_PosDef_Blocky(lam, e); warm.append((time.perf_counter() - t0) * 1e3)
    print(f"cold {cold:.2f} ms, warm (host arrays in and out) median {np.median(warm):.3f} ms, device phases {solver.times.as_dict()}")
    print(f"residual ||A x - b||_inf / ||b||_inf = {np.abs(A @ eta - lam.rhs).max() / np.abs(lam.rhs).max():.2e}")
    if x_true is not None:
        print(f"error vs the known solution: {np.abs(eta - x_true).max():.2e}")
