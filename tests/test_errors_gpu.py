"""Error behaviour of the C ABI on the GPU box, mirrored on the reference's contract:
false for not-positive-definite, exceptions (never exit) for bad input / unsupported structure, and
solver objects that can be freed and reused."""
import copy
import dataclasses

import numpy as np
import pytest

from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu


def rel_inf(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max())


def test_malformed_structure_is_rejected_with_an_exception():
    lam = synth.pose_chain(n=20, d=6)
    solver = CLinearSolver_HIP()
    lower = dataclasses.replace(lam, brow_idx=(lam.n_bcols - 1 - lam.brow_idx).astype(np.int32))   # below the diagonal
    with pytest.raises(ValueError):
        solver.Solve_PosDef(lower, lam.rhs.copy())
    no_diag = dataclasses.replace(lam, bcol_ptr=np.concatenate([lam.bcol_ptr[:-1], [lam.bcol_ptr[-1] - 1]]),
                                  brow_idx=lam.brow_idx[:-1], values=lam.values[:-36])
    with pytest.raises(ValueError):
        solver.Solve_PosDef(no_diag, lam.rhs.copy())
    with pytest.raises(ValueError):                      # eta of the wrong length
        solver.Solve_PosDef(lam, lam.rhs[:-1].copy())
    # the solver object survives and still works
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta) and rel_inf(eta, O.solve_sparse(lam)[1]) < 1e-10


def test_free_memory_and_reuse_and_copy_semantics():
    lam_a, lam_b = synth.sphere(12, 12, seed=1), synth.pose_chain(n=700, d=3, seed=2)
    solver = CLinearSolver_HIP(leaf_size=8)
    for lam in (lam_a, lam_b, lam_a):                   # structure changes between calls
        eta = lam.rhs.copy()
        solver.Clear_SymbolicDecomposition()
        assert solver.Solve_PosDef_Blocky(lam, eta)
        assert rel_inf(eta, O.solve_sparse(lam)[1]) < 1e-10
        solver.Free_Memory()
    clone = copy.copy(solver)                            # copies carry the configuration, not the state
    assert clone._options == solver._options and clone._h != solver._h
    eta = lam_b.rhs.copy()
    assert clone.Solve_PosDef(lam_b, eta) and rel_inf(eta, O.solve_sparse(lam_b)[1]) < 1e-10


def test_structure_change_without_notice_is_detected_by_the_python_mirror():
    solver = CLinearSolver_HIP()
    a, b = synth.pose_chain(n=300, d=6, seed=1), synth.pose_chain(n=300, d=6, seed=9, loop_every=7, loop_min=2, loop_max=6)
    ea, eb = a.rhs.copy(), b.rhs.copy()
    assert solver.Solve_PosDef_Blocky(a, ea) and solver.Solve_PosDef_Blocky(b, eb)
    assert rel_inf(eb, O.solve_sparse(b)[1]) < 1e-10


def test_schur_fallback_can_be_switched_off():
    """Option schur_fallback = 0: a structure the Schur kernels do not take is an error again (the default solves it
    through the sparse block path, tests/test_fallback_gpu.py)."""
    lam = synth.pose_chain(n=30, d=6)
    with pytest.raises((NotImplementedError, ValueError)):
        CLinearSolver_Schur_HIP(schur_fallback=0).Solve_PosDef(lam, lam.rhs.copy())


def test_nan_in_lambda_is_reported_as_failure():
    lam = synth.pose_chain(n=200, d=6)
    vals = lam.values.copy()
    vals[len(vals) // 2] = np.nan
    bad = dataclasses.replace(lam, values=vals)
    assert CLinearSolver_HIP().Solve_PosDef(bad, bad.rhs.copy()) is False


def test_a_handle_can_be_dropped_or_used_at_once_while_its_streams_are_still_coming_up():
    """Round 6: slampp_hip_create returns before the handle's streams exist (capi.hip: device_bringup, on a thread).  Destroying
    the handle straight away, asking for its stream straight away and analyzing straight away all have to meet that thread
    properly; options and the structure do not need it."""
    for _ in range(8):
        s = CLinearSolver_HIP()
        del s                                   # destroyed while the thread may still be creating the streams
    s = CLinearSolver_HIP()
    assert s.stream() != 0                      # waits for the thread
    del s
    lam = synth.pose_chain(n=300, seed=4)
    for _ in range(4):
        s = CLinearSolver_HIP()
        s.set_option("leaf_size", 4)            # host state only
        eta = lam.rhs.copy()
        assert s.Solve_PosDef(lam, eta)         # set_structure + analyze + solve on a handle fresh from create
        ok, x_ref, _ = O.solve_sparse(lam)
        assert ok and np.abs(eta - x_ref).max() / np.abs(x_ref).max() < 1e-10
        del s
