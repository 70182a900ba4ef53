"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE -- the CPU restatement of the
reference's Lambda-solve path, see oracle/slampp_oracle.c).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by slam_plus_plus_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_HARNESS = os.path.join(_HERE, "_ref", "ref_harness")
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_solve_sparse.restype = C.c_int
        _lib.oracle_solve_schur.restype = C.c_int
        _lib.oracle_exec_plan.restype = C.c_int
        _lib.oracle_assemble_lambda.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def solve_sparse(lam, perm=None):
    """(ok, x, stats) -- up-looking block Cholesky + substitutions on the CPU."""
    cs = np.ascontiguousarray(lam.cumsum, dtype=np.int64)
    bp = np.ascontiguousarray(lam.bcol_ptr, dtype=np.int64)
    br = np.ascontiguousarray(lam.brow_idx, dtype=np.int32)
    vals = np.ascontiguousarray(lam.values, dtype=np.float64)
    x = np.array(lam.rhs, dtype=np.float64, copy=True)
    pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
    stats = np.zeros(4)
    rc = lib().oracle_solve_sparse(C.c_int64(lam.n_bcols), _p(cs), _p(bp), _p(br), _p(vals), _p(x), _p(pm), _p(stats))
    if rc < 0:
        raise ValueError("oracle_solve_sparse: bad input")
    out = {"r_blocks": int(stats[0]), "r_values": int(stats[1])}
    if rc == 0 and stats[2] > 0:   # (max / min diagonal entry of R)^2 <= cond_2(Lambda): the conditioning proxy of the reports
        out["cond_proxy"] = float((stats[3] / stats[2]) ** 2)
    return rc == 0, x, out


def solve_schur(lam, n_cut=None, want_S=False):
    """(ok, x, S or None, rhs_reduced)."""
    n_cut = int(lam.n_matrix_cut if n_cut is None else n_cut)
    cs = np.ascontiguousarray(lam.cumsum, dtype=np.int64)
    bp = np.ascontiguousarray(lam.bcol_ptr, dtype=np.int64)
    br = np.ascontiguousarray(lam.brow_idx, dtype=np.int32)
    vals = np.ascontiguousarray(lam.values, dtype=np.float64)
    x = np.array(lam.rhs, dtype=np.float64, copy=True)
    N = int(cs[n_cut])
    S = np.zeros(N * N) if want_S else None
    rr = np.zeros(N)
    rc = lib().oracle_solve_schur(C.c_int64(lam.n_bcols), _p(cs), _p(bp), _p(br), _p(vals), _p(x),
                                  C.c_int64(n_cut), _p(S), _p(rr))
    if rc < 0:
        raise ValueError("oracle_solve_schur: bad structure")
    return rc == 0, x, (S.reshape(N, N).T.copy() if want_S else None), rr   # S returned row-major [row, col]


def exec_plan(lam, plan: dict):
    """Replays the product's elimination plan on the CPU: (status, x); status 0 ok, 1 not PD, 2 bad schedule."""
    vals = np.ascontiguousarray(lam.values, dtype=np.float64)
    x = np.array(lam.rhs, dtype=np.float64, copy=True)
    cs = np.ascontiguousarray(lam.cumsum, dtype=np.int64)
    a = {k: np.ascontiguousarray(v) for k, v in plan.items() if isinstance(v, np.ndarray)}
    rc = lib().oracle_exec_plan(
        C.c_int64(plan["n_bcols"]), _p(a["perm"]), _p(a["dim"]), _p(a["lptr"]), _p(a["lrow"]), _p(a["loff"]),
        _p(a["asrc"]), _p(a["atrans"]), _p(a["pptr"]), _p(a["pa"]), _p(a["pb"]), _p(a["rptr"]), _p(a["rblk"]),
        C.c_int64(plan["n_stages"]), _p(a["stage_ptr"]), _p(a["task_ptr"]), _p(a["task_cols"]),
        _p(cs), C.c_int64(plan["l_values"]), _p(vals), _p(x), _p(a["dense_pos"]), C.c_int64(plan["dense_dim"]))
    return rc, x


def assemble_lambda(lam, edges):
    """(values, eta) of Lambda for one edge set (slam_plus_plus_amd.synth.EdgeSet) on the structure of ``lam``."""
    cs = np.ascontiguousarray(lam.cumsum, dtype=np.int64)
    bp = np.ascontiguousarray(lam.bcol_ptr, dtype=np.int64)
    br = np.ascontiguousarray(lam.brow_idx, dtype=np.int32)
    values = np.zeros(lam.values.shape[0])
    eta = np.zeros(int(cs[-1]))
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    J0, J1, si, err, w = f(edges.J0), f(edges.J1), f(edges.sigma_inv), f(edges.err), f(edges.weight)
    uf, ue = f(edges.unary_factor), f(edges.unary_error)
    v0 = np.ascontiguousarray(edges.v0, dtype=np.int64)
    v1 = np.ascontiguousarray(edges.v1, dtype=np.int64)
    rc = lib().oracle_assemble_lambda(C.c_int64(lam.n_bcols), _p(cs), _p(bp), _p(br), C.c_int64(edges.n_edges),
                                      _p(v0), _p(v1), C.c_int64(edges.rd), _p(J0), _p(J1), _p(si), _p(err), _p(w),
                                      C.c_int64(edges.unary_vertex), _p(uf), _p(ue), _p(values), _p(eta))
    if rc != 0:
        raise ValueError("oracle_assemble_lambda: bad edge set")
    return values, eta


def schur_marginals(lam, n_cut=None):
    """(camera blocks [nc, dc, dc], landmark blocks [np, dp, dp]) of the covariance Lambda^-1 -- numpy restatement of
    CSchurComplement_Marginals::Schur_Marginals (/root/reference/include/slam/BAMarginals.h:579-806) as its callers
    feed it (NonlinearSolver_Lambda_DL.h:1590-1640): S = A - U Dinv U^T = R^T R; landmark i gets
    Dinv_ii + (R^-T U Dinv_i)^T (R^-T U Dinv_i) (BAMarginals.h:703-727), the cameras the diagonal blocks of S^-1
    (:765-772).  Dense, for test sizes only."""
    import scipy.linalg as sl
    n_cut = int(lam.n_matrix_cut if n_cut is None else n_cut)
    cs = np.asarray(lam.cumsum)
    nx = int(cs[n_cut])
    dense = lam.to_scipy().toarray()
    A, U, D = dense[:nx, :nx], dense[:nx, nx:], dense[nx:, nx:]
    dc, dp = int(cs[1] - cs[0]), int(cs[n_cut + 1] - cs[n_cut])
    n_pts = lam.n_bcols - n_cut
    Dinv = np.zeros_like(D)
    for i in range(n_pts):
        sl_ = slice(i * dp, (i + 1) * dp)
        Dinv[sl_, sl_] = np.linalg.inv(D[sl_, sl_])
    U_Dinv = U @ Dinv
    S = A - U_Dinv @ U.T
    R = np.linalg.cholesky(S).T                                    # upper, S = R^T R
    B = sl.solve_triangular(R, U_Dinv, trans="T", lower=False)     # R^-T U Dinv, all landmark columns at once
    pts = np.empty((n_pts, dp, dp))
    for i in range(n_pts):
        sl_ = slice(i * dp, (i + 1) * dp)
        pts[i] = Dinv[sl_, sl_] + B[:, sl_].T @ B[:, sl_]
    Rinv = sl.solve_triangular(R, np.eye(nx), lower=False)
    Sinv = Rinv @ Rinv.T
    cams = np.stack([Sinv[c * dc:(c + 1) * dc, c * dc:(c + 1) * dc] for c in range(n_cut)])
    return cams, pts


def have_reference() -> bool:
    return os.path.exists(REF_HARNESS) and os.access(REF_HARNESS, os.X_OK)


def host_cores() -> int:
    """CPUs this process may really use: the visible count, cut down to the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def reference_env() -> dict:
    """Environment for the compiled reference: its OpenMP loops get as many threads as the process may use (a box that
    shows 256 CPUs under a 16-CPU quota otherwise runs 256 spinning threads, and every parallel region takes ~1 s)."""
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(host_cores()))
    return env


def reference_solve(problem_path: str, solver: str, x_path: str = "-", reps: int = 1, timeout: int = 600):
    """Runs the compiled reference (oracle/_ref/ref_harness) on a SPPLAM01 file; returns its JSON."""
    import json
    out = subprocess.run([REF_HARNESS, "solve", problem_path, solver, x_path, str(reps)],
                         capture_output=True, text=True, timeout=timeout, env=reference_env())
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise RuntimeError(f"ref_harness failed: {out.stdout} {out.stderr}")
    return json.loads(line[-1])
