"""Lambda / eta assembly: the oracle's restatement against what the reference's
CNonlinearSolver_Lambda assembled (golden fixtures recorded by `ref_harness lambda_dump`), and the host logic
around it.  CPU only."""
import numpy as np
import pytest

from golden_util import assembly_names, load_assembly, rel_inf
from oracle import oracle_lib
from slam_plus_plus_amd import synth


@pytest.mark.parametrize("name", assembly_names())
def test_oracle_assembly_matches_reference(name):
    lam, es, _ = load_assembly(name)
    assert int((es.v0 > es.v1).sum()) > 0          # the transposed off-diagonal case is present
    values, eta = oracle_lib.assemble_lambda(lam, es)
    assert rel_inf(values, lam.values) < 1e-14
    assert rel_inf(eta, lam.rhs) < 1e-14


def test_reference_weights_vertex0_rhs_twice():
    """The recorded eta only matches with w^2 on vertex 0's right-hand side (BaseTypes_Binary.h:813-815):
    numpy restatement with w on both sides is off by far more than rounding on the robust (SE3) fixture."""
    lam, es, _ = load_assembly("assembly_se3_n40")
    assert es.weight.min() < 0.9
    eta = np.zeros_like(lam.rhs)
    d = int(lam.cumsum[1])
    for e in range(es.n_edges):
        J0, J1, S = es.J0[e].T, es.J1[e].T, es.sigma_inv[e].T
        w = es.weight[e]
        eta[es.v0[e] * d:(es.v0[e] + 1) * d] += J0.T @ S @ es.err[e] * w
        eta[es.v1[e] * d:(es.v1[e] + 1) * d] += J1.T @ S @ es.err[e] * w
    eta[:d] += es.unary_error
    assert rel_inf(eta, lam.rhs) > 1e-6


@pytest.mark.parametrize("name", assembly_names())
def test_structure_from_edges(name):
    lam, es, _ = load_assembly(name)
    st = synth.structure_from_edges(np.diff(lam.cumsum), es.v0, es.v1)
    assert np.array_equal(st.cumsum, lam.cumsum)
    assert np.array_equal(st.bcol_ptr, lam.bcol_ptr)
    assert np.array_equal(st.brow_idx, lam.brow_idx)
    assert st.values.shape == lam.values.shape


@pytest.mark.parametrize("name", assembly_names())
def test_assembled_system_solves_to_reference_dx(name):
    lam, es, x_ref = load_assembly(name)
    values, eta = oracle_lib.assemble_lambda(lam, es)
    asm = synth.BlockSystem(lam.cumsum, lam.bcol_ptr, lam.brow_idx, values, eta, 0, name)
    ok, x, _ = oracle_lib.solve_sparse(asm)
    assert ok and rel_inf(x, x_ref) < 1e-10


def test_synthetic_edge_set_is_posdef_and_symmetric():
    rng = np.random.default_rng(3)
    n = 300
    v0 = np.concatenate([np.arange(n - 1), rng.integers(20, n, 80)])
    v1 = np.concatenate([np.arange(1, n), np.zeros(80, dtype=np.int64)])
    v1[n - 1:] = v0[n - 1:] - rng.integers(2, 20, 80)     # loop closures pointing backwards: flipped blocks
    dims = np.full(n, 6)
    es = synth.random_edge_set(dims, v0, v1, rd=6, seed=4)
    lam = synth.structure_from_edges(dims, v0, v1)
    lam.values, lam.rhs = oracle_lib.assemble_lambda(lam, es)
    A = lam.to_scipy().toarray()
    assert np.allclose(A, A.T) and np.linalg.eigvalsh(A).min() > 0
    # against a dense numpy assembly
    B = np.zeros_like(A)
    for e in range(es.n_edges):
        J = np.zeros((6, 6 * n))
        J[:, v0[e] * 6:v0[e] * 6 + 6] = es.J0[e].T
        J[:, v1[e] * 6:v1[e] * 6 + 6] = es.J1[e].T
        B += J.T @ (es.sigma_inv[e].T * es.weight[e]) @ J
    B[:6, :6] += es.unary_factor @ es.unary_factor.T
    assert np.abs(A - B).max() < 1e-11 * np.abs(B).max()


def test_oracle_rejects_edge_without_block():
    lam, es, _ = load_assembly("assembly_se2_n40")
    bad = synth.EdgeSet(es.n_verts, es.v0.copy(), es.v1.copy(), es.J0, es.J1, es.sigma_inv, es.err, es.weight,
                        0, es.unary_factor, es.unary_error)
    bad.v0[0], bad.v1[0] = 0, 39        # no such block in Lambda
    with pytest.raises(ValueError):
        oracle_lib.assemble_lambda(lam, bad)
