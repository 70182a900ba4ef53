"""A drop-in does not throw where the reference solves (LinearSolver_Schur.h:1635-1638: no landmark part -> the base solver;
:1721-1726: C not block diagonal -> InverseOf_Symmteric_FBS; any block size).  The library sends such systems through the
sparse block path -- block columns wider than 8 cut into pieces on the way -- and the solution is the reference's:
parity against the oracle / a direct sparse solve of the same Lambda, rel-inf 1e-10."""
import numpy as np
import pytest

from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel_inf(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max())


def direct_solve(lam):
    import scipy.sparse.linalg as spl
    return spl.spsolve(lam.to_scipy().tocsc(), lam.rhs)


def system_from_dense(M, dims, rhs, n_matrix_cut=0, name=""):
    """Upper block-CSC of a dense symmetric matrix: every block (r <= c) with a nonzero entry is stored."""
    cumsum = np.concatenate([[0], np.cumsum(dims)]).astype(np.int64)
    n = len(dims)
    bcol_ptr, brow, vals = [0], [], []
    for c in range(n):
        for r in range(c + 1):
            blk = M[cumsum[r]:cumsum[r + 1], cumsum[c]:cumsum[c + 1]]
            if r == c or np.any(blk != 0):
                brow.append(r)
                vals.append(blk.T.ravel())       # column-major
        bcol_ptr.append(len(brow))
    return synth.BlockSystem(cumsum, np.asarray(bcol_ptr, dtype=np.int64), np.asarray(brow, dtype=np.int32),
                             np.concatenate(vals), rhs, n_matrix_cut, name)


def test_no_landmark_part_is_solved_by_the_sparse_path():
    lam = synth.pose_chain(n=300, d=6)
    ok, x_ref, _ = O.solve_sparse(lam)
    eta = lam.rhs.copy()
    assert ok and CLinearSolver_Schur_HIP().Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_landmark_landmark_blocks(devices):
    """C is not block diagonal (a landmark-landmark constraint): the reference inverts C as a sparse symmetric matrix; here
    the whole system takes the sparse block path -- on a multi-device handle as well."""
    rng = np.random.default_rng(4)
    lam = synth.ba(12, 80, k=3, mode="uniform", seed=9)
    M = lam.to_scipy().toarray()
    nc, n_x = lam.n_matrix_cut, int(lam.cumsum[lam.n_matrix_cut])
    for p, q in ((3, 7), (10, 11), (20, 70), (41, 42)):
        a, b = n_x + 3 * p, n_x + 3 * q
        B = 0.05 * rng.standard_normal((3, 3))
        M[a:a + 3, b:b + 3] += B
        M[b:b + 3, a:a + 3] += B.T
    assert np.linalg.eigvalsh(M).min() > 0
    lam2 = system_from_dense(M, np.diff(lam.cumsum), lam.rhs, nc)
    assert lam2.n_blocks == lam.n_blocks + 4
    x_ref = np.linalg.solve(M, lam.rhs)
    eta = lam2.rhs.copy()
    solver = CLinearSolver_Schur_HIP(devices=devices) if devices else CLinearSolver_Schur_HIP()
    assert solver.Solve_PosDef(lam2, eta)
    assert rel_inf(eta, x_ref) < TOL
    eta = 2.0 * lam2.rhs        # the analysis is kept
    assert solver.Solve_PosDef_Blocky(lam2, eta) and rel_inf(eta, 2.0 * x_ref) < TOL
    with pytest.raises((NotImplementedError, ValueError)):
        CLinearSolver_Schur_HIP(schur_fallback=0).Solve_PosDef(lam2, lam2.rhs.copy())


@pytest.mark.parametrize("d", [9, 11, 12, 16, 17, 25])
def test_block_columns_wider_than_eight(d):
    lam = synth.pose_chain(n=60, d=d, loop_every=5, loop_min=2, loop_max=4)
    x_ref = direct_solve(lam)
    eta = lam.rhs.copy()
    solver = CLinearSolver_HIP()
    assert solver.Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL
    eta = lam.rhs.copy()      # again with the cached analysis, from the pinned staging
    vals, rhs = solver.host_staging()
    vals[:] = lam.values
    assert solver.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, x_ref) < TOL
    eta = -lam.rhs
    assert solver.Solve_Again(eta) and rel_inf(eta, -x_ref) < TOL


def test_mixed_wide_and_narrow_blocks_and_a_not_positive_definite_one():
    rng = np.random.default_rng(8)
    dims = np.array([11, 3, 6, 12, 2, 9, 7, 3, 10, 5])
    n = int(dims.sum())
    G = rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.15)
    M = G @ G.T + 5.0 * np.eye(n)
    rhs = rng.standard_normal(n)
    lam = system_from_dense(M, dims, rhs)
    eta = rhs.copy()
    assert CLinearSolver_HIP().Solve_PosDef(lam, eta)
    assert rel_inf(eta, np.linalg.solve(M, rhs)) < TOL
    M[20, 20] = -50.0
    bad = system_from_dense(M, dims, rhs)
    assert CLinearSolver_HIP().Solve_PosDef(bad, rhs.copy()) is False


def test_cameras_with_intrinsics_in_the_vertex():
    """11-dimensional cameras (pose + 5 intrinsics), 3-d landmarks: not a block-size pair the Schur kernels are built for;
    the Schur class still solves it (sparse block path, the cameras' columns in two pieces)."""
    lam = synth.ba(10, 120, k=3, mode="uniform", seed=2, cam_dim=11)
    x_ref = direct_solve(lam)
    eta = lam.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL
