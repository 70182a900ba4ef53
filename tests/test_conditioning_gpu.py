"""Parity where real data lives: the HIP path on the conditioning sweep (tests/golden/cond_*.npz, written by the
compiled reference through tests/golden/make_golden.py): pose chains and BA systems at cond_2(Lambda) of about 1e6 /
1e9 / 1e12, the damped Lambda the reference's own CNonlinearSolver_Lambda_LM hands to its linear solver on 200
CVertexCam cameras, and two systems that are indefinite by twice their smallest eigenvalue.

What a correct Cholesky solver can be held to on such systems is what the reference's own five solvers show among
themselves ('Conditioning vs the 1e-10 bar'): every fixture records their largest pairwise rel-inf
distance (``spread``), their forward errors against a solution refined in 60-digit arithmetic (``err_true_*``) and their
verdicts.  The HIP path must stay within ten times the spread of CHOLMOD's solution and within ten times their worst
forward error of the refined one -- and within the north star's 1e-10 wherever the reference's solvers are that good --,
and must say "not positive definite" where CHOLMOD supernodal, CSparse and the native block solver say so
(/root/reference/src/slam/LinearSolver_CholMod.cpp:305-317, src/slam/BlockMatrix.cpp:9752-9771)."""
import numpy as np
import pytest

from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
from golden_util import cond_names, load_cond, cond_bounds, rel_inf, LLT_ORACLES

pytestmark = pytest.mark.gpu

PD = [n for n in cond_names() if bool(load_cond(n)[1]["positive_definite"])]
NOT_PD = [n for n in cond_names() if n not in PD]
BA = [n for n in PD if load_cond(n)[0].n_matrix_cut]

# every way the library factors: the sparse block path as it plans it, the caller's own order, with and without a dense
# top (the matrix-core Cholesky with its reciprocal pivots, csrc/dense_device.inl), one level per launch, column by column
SPARSE_VARIANTS = {
    "default": {},
    "natural_order": {"natural_order": 1},
    "dense_top": {"dense_top_nb": 2, "dense_top_min_dim": 0},
    "no_dense_top": {"dense_top_nb": 0},
    "levels": {"task_height": 1},
    "columns": {"panel": 0},
}
# the reduced camera system dense on the matrix cores / through the sparse block path; S from the contribution lists
SCHUR_VARIANTS = {
    "default": {},
    "dense_S": {"schur_sparse": 0},
    "sparse_S": {"schur_sparse": 1},
    "lists": {"schur_tiles": 0},
}


def _report(name, what, err_ref, err_true, ref):
    b_ref, b_true = cond_bounds(ref)
    print(f"{name} [{what}]: cond2 {float(ref['cond2']):.2e}  inter-oracle spread {float(ref['spread']):.2e}  "
          f"vs CHOLMOD {err_ref:.2e} (bound {b_ref:.1e})  vs refined {err_true:.2e} (bound {b_true:.1e})")


@pytest.mark.parametrize("variant", sorted(SPARSE_VARIANTS))
@pytest.mark.parametrize("name", PD)
def test_sparse_path_within_the_reference_solvers_spread(name, variant):
    lam, ref = load_cond(name)
    b_ref, b_true = cond_bounds(ref)
    solver = CLinearSolver_HIP(**SPARSE_VARIANTS[variant])
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    err_ref, err_true = rel_inf(eta, ref["x_cholmod_super"]), rel_inf(eta, ref["x_true"])
    _report(name, variant, err_ref, err_true, ref)
    assert err_ref < b_ref and err_true < b_true
    eta2 = lam.rhs.copy()                     # the kept factor on the same right-hand side: the same answer
    assert solver.Solve_Again(eta2)
    assert rel_inf(eta2, ref["x_true"]) < b_true


@pytest.mark.parametrize("variant", sorted(SCHUR_VARIANTS))
@pytest.mark.parametrize("name", BA)
def test_schur_path_within_the_reference_solvers_spread(name, variant):
    lam, ref = load_cond(name)
    b_ref, b_true = cond_bounds(ref)
    solver = CLinearSolver_Schur_HIP(**SCHUR_VARIANTS[variant])
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    err_ref, err_true = rel_inf(eta, ref["x_schur"]), rel_inf(eta, ref["x_true"])
    _report(name, "schur " + variant, err_ref, err_true, ref)
    assert err_ref < b_ref and err_true < b_true


@pytest.mark.parametrize("variant", sorted(SPARSE_VARIANTS))
@pytest.mark.parametrize("name", NOT_PD)
def test_sparse_path_agrees_with_the_llt_oracles_on_barely_indefinite_systems(name, variant):
    lam, ref = load_cond(name)
    assert not any(bool(ref[f"ok_{s}"]) for s in LLT_ORACLES)
    assert CLinearSolver_HIP(**SPARSE_VARIANTS[variant]).Solve_PosDef(lam, lam.rhs.copy()) is False


@pytest.mark.parametrize("variant", sorted(SCHUR_VARIANTS))
@pytest.mark.parametrize("name", [n for n in NOT_PD if load_cond(n)[0].n_matrix_cut])
def test_schur_path_agrees_with_the_llt_oracles_on_barely_indefinite_systems(name, variant):
    """The reference's own Schur solver says yes here (its base solver's LDL^T: SURVEY.md appendix A); the contract is the
    LL^T one: Lambda is not positive definite, so neither is S, and the factorization of S must notice."""
    lam, ref = load_cond(name)
    assert CLinearSolver_Schur_HIP(**SCHUR_VARIANTS[variant]).Solve_PosDef(lam, lam.rhs.copy()) is False
