"""GPU parity tests of the sparse block Cholesky path (pose graphs): the HIP path, called through
the C ABI (slam_plus_plus_amd.hip_solver), against the CPU oracle on the same seeded inputs.
Tolerance: BASELINE.json's north_star, ||x_gpu - x_ref||_inf / ||x_ref||_inf < 1e-10 (fp64)."""
import numpy as np
import pytest

from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel_inf(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max())


CASES = {
    "chain6": lambda: synth.pose_chain(n=3000, d=6),
    "chain3": lambda: synth.pose_chain(n=2000, d=3, seed=7),
    "chain7": lambda: synth.pose_chain(n=1500, d=7, seed=8),
    "sphere": lambda: synth.sphere(50, 50),                 # C2 look-alike, full size
    "manhattan": lambda: synth.manhattan(3500),             # C1 look-alike, full size
    "tiny": lambda: synth.pose_chain(n=2, d=6, loop_every=50),
    "single": lambda: synth.pose_chain(n=1, d=6),
    "ba_sparse_path": lambda: synth.ba(30, 2000),           # mixed 6/3 blocks through the sparse path
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_parity_with_oracle(name):
    lam = CASES[name]()
    ok, x_ref, _ = O.solve_sparse(lam)
    assert ok
    solver = CLinearSolver_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL
    # warm path: structure cached, new values / rhs
    eta2 = 2.0 * lam.rhs
    assert solver.Solve_PosDef_Blocky(lam, eta2)
    assert rel_inf(eta2, 2.0 * x_ref) < TOL
    # another rhs with the kept factor
    eta3 = lam.rhs.copy()
    assert solver.Solve_Again(eta3)
    assert rel_inf(eta3, x_ref) < TOL


@pytest.mark.parametrize("leaf,sub,dense_nb", [(1, 1, 0), (4, 64, 24), (64, 4, 8), (1000000, 1000000, 0), (4, 16, 1)])
def test_schedule_knobs_do_not_change_the_answer(leaf, sub, dense_nb):
    lam = synth.sphere(20, 20)
    ok, x_ref, _ = O.solve_sparse(lam)
    solver = CLinearSolver_HIP(leaf_size=leaf, subtree_size=sub, dense_top_nb=dense_nb)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL


def test_dense_top_not_positive_definite_returns_false():
    """An indefinite block deep inside a big separator: the failure must surface from the dense factorization."""
    import dataclasses
    lam = synth.sphere(24, 24, seed=3)
    solver = CLinearSolver_HIP()
    assert solver.Solve_PosDef(lam, lam.rhs.copy()) and solver.stats()["schur_dim"] > 0
    dense = solver.plan()["dense_pos"] >= 0
    j_old = int(solver.plan()["perm"][np.nonzero(dense)[0][-1]])      # the last eliminated column
    off = lam.block_value_offsets()
    k = int(lam.bcol_ptr[j_old + 1] - 1)
    vals = lam.values.copy()
    vals[off[k]:off[k + 1]] -= 1e4 * np.eye(6).ravel()
    bad = dataclasses.replace(lam, values=vals)
    assert O.solve_sparse(bad)[0] is False
    assert solver.Solve_PosDef_Blocky(bad, bad.rhs.copy()) is False


def test_not_positive_definite_returns_false():
    lam = synth.indefinite()
    ok, _, _ = O.solve_sparse(lam)
    assert not ok
    solver = CLinearSolver_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta) is False


def test_sync_answers_for_every_solve_enqueued_since_the_last_one():
    """slampp_hip_sync() returns NOT_POSDEF if any factorization enqueued since the last sync was not positive definite,
    not only the last one (a per-solve reset of the flag used to erase an earlier failure); and the flag is clean again
    afterwards."""
    import torch
    lam = synth.pose_chain(n=400, seed=5)
    off = lam.block_value_offsets()
    k = int(lam.bcol_ptr[200 + 1] - 1)                              # a diagonal block in the middle
    vals_bad = lam.values.copy()
    vals_bad[off[k]:off[k + 1]] -= 1e4 * np.eye(6).ravel()
    dev = torch.device("cuda:0")
    solver = CLinearSolver_HIP()
    solver.SymbolicDecomposition_Blocky(lam)
    good, bad = torch.from_numpy(lam.values).to(dev), torch.from_numpy(vals_bad).to(dev)
    b1, b2, b3 = (torch.from_numpy(lam.rhs).to(dev) for _ in range(3))
    solver.factor_solve_device_async(bad.data_ptr(), b1.data_ptr())
    solver.factor_solve_device_async(good.data_ptr(), b2.data_ptr())
    assert solver.sync() is False                                    # the first of the two failed
    solver.factor_solve_device_async(good.data_ptr(), b3.data_ptr())
    assert solver.sync() is True
    ok, x_ref, _ = O.solve_sparse(lam)
    assert ok and rel_inf(b3.cpu().numpy(), x_ref) < TOL
    assert rel_inf(b2.cpu().numpy(), x_ref) < TOL                   # (the good solve between them was a solve all the same)


def test_residual_full_size_c3():
    """BASELINE config C3 (100k-pose SE(3)): too big for the scalar oracle to be quick, so check the
    size-independent property ||Lambda x - eta|| / ||eta|| and linearity in eta."""
    lam = synth.pose_chain()
    A = lam.to_scipy()
    solver = CLinearSolver_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    r = A @ eta - lam.rhs
    assert np.abs(r).max() / np.abs(lam.rhs).max() < 1e-9
    eta2 = -3.0 * lam.rhs
    assert solver.Solve_PosDef_Blocky(lam, eta2)
    assert rel_inf(eta2, -3.0 * eta) < TOL


@pytest.mark.parametrize("tiles", [0, 1])
@pytest.mark.parametrize("name", ["sphere", "manhattan", "grid"])
def test_dense_top_tile_schedule_matches_dense_schedule(name, tiles):
    """The dense top through the level schedule over its nonzero 64x64 tiles (independent chains aligned to tile
    boundaries, side by side) and through the plain tile-by-tile dense factorization."""
    lam = {"sphere": lambda: synth.sphere(40, 40, seed=5), "manhattan": lambda: synth.manhattan(3000, seed=6),
           "grid": lambda: synth.sphere(70, 70, seed=7)}[name]()
    ok, x_ref, _ = O.solve_sparse(lam)
    assert ok
    solver = CLinearSolver_HIP(dense_top_tiles=tiles)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta) and solver.stats()["schur_dim"] > 0
    assert rel_inf(eta, x_ref) < TOL
    eta2 = lam.rhs.copy()
    assert solver.Solve_Again(eta2)          # the kept factor: stand-alone forward substitution over the same tiles
    assert rel_inf(eta2, x_ref) < TOL


def dense_factor(lam, plan, l_values):
    """The lower factor as a dense matrix in the plan's (permuted) order."""
    dim, lptr, lrow, loff = plan["dim"], plan["lptr"], plan["lrow"], plan["loff"]
    cs = np.concatenate([[0], np.cumsum(dim)])
    Lm = np.zeros((cs[-1], cs[-1]))
    for j in range(len(dim)):
        for k in range(lptr[j], lptr[j + 1]):
            i = lrow[k]
            Lm[cs[i]:cs[i + 1], cs[j]:cs[j + 1]] = l_values[loff[k]:loff[k] + dim[i] * dim[j]].reshape(dim[j], dim[i]).T
    return Lm, cs


@pytest.mark.parametrize("natural", [1, 0])
@pytest.mark.parametrize("name", ["chain6", "sphere_small", "mixed"])
def test_factorize_returns_the_cholesky_factor(name, natural):
    """slampp_hip_factorize (the reference's Factorize_PosDef_Blocky hands its factor back to the nonlinear solver): L of
    the permuted Lambda against numpy's Cholesky; with natural_order the permutation is the identity."""
    lam = {"chain6": lambda: synth.pose_chain(n=300, d=6, seed=3), "sphere_small": lambda: synth.sphere(12, 12, seed=4),
           "mixed": lambda: synth.ba(10, 150, seed=5)}[name]()
    solver = CLinearSolver_HIP(natural_order=natural, dense_top_nb=0)
    ok, plan, l_values = solver.factorize(lam)
    assert ok
    perm = plan["perm"]
    assert (natural == 0) or np.array_equal(perm, np.arange(lam.n_bcols))
    Lm, cs_new = dense_factor(lam, plan, l_values)
    A = lam.to_scipy().toarray()
    cs_old = lam.cumsum
    idx = np.concatenate([np.arange(cs_old[o], cs_old[o + 1]) for o in perm])
    Lref = np.linalg.cholesky(A[np.ix_(idx, idx)])
    assert np.abs(np.triu(Lm, 1)).max() == 0.0
    assert np.abs(Lm - Lref).max() < 1e-11 * np.abs(Lref).max()


def test_factorize_not_posdef():
    lam = synth.indefinite(40, 6, seed=5)
    ok, _, _ = CLinearSolver_HIP(natural_order=1, dense_top_nb=0).factorize(lam)
    assert not ok


@pytest.mark.parametrize("tiles", [-1, 0])
@pytest.mark.parametrize("name", ["sphere", "manhattan", "grid"])
def test_factorize_hands_back_the_dense_top_too(name, tiles):
    """Round 4: the reference's Factorize_PosDef_Blocky takes any matrix (LinearSolver_CholMod.cpp:362-544); so does
    slampp_hip_factorize now -- the big separators of a 2-D-like graph are factored as one dense matrix on the matrix cores
    and their columns come back in the same block layout as the rest (until round 3: refused, dense_top_nb = 0 required)."""
    lam = {"sphere": lambda: synth.sphere(30, 30), "manhattan": lambda: synth.manhattan(3500),
           "grid": lambda: synth.sphere(24, 40, seed=8)}[name]()
    solver = CLinearSolver_HIP(dense_top_tiles=tiles)
    ok, st, l_values = solver.factorize(lam)
    assert ok and solver.stats()["schur_dim"] > 0            # there IS a dense top
    Lm, cs_new = dense_factor(lam, st, l_values)
    A = lam.to_scipy().toarray()
    idx = np.concatenate([np.arange(lam.cumsum[o], lam.cumsum[o + 1]) for o in st["perm"]])
    Lref = np.linalg.cholesky(A[np.ix_(idx, idx)])
    assert np.abs(np.triu(Lm, 1)).max() == 0.0
    assert np.abs(Lm - Lref).max() < 1e-10 * np.abs(Lref).max()
    eta = lam.rhs.copy()                                      # ... and the handle still solves
    assert solver.Solve_PosDef_Blocky(lam, eta)
    assert np.abs(lam.to_scipy() @ eta - lam.rhs).max() < 1e-9 * np.abs(lam.rhs).max()


@pytest.mark.parametrize("dims", [(11, 3), (9, 9), (25, 6), (16, 2)])
def test_factorize_puts_the_pieces_of_wide_block_columns_together_again(dims):
    """Block columns wider than 8 (cameras with their intrinsics in the vertex: 11) are factored in pieces of at most 8; the
    factor comes back in the CALLER's blocks (natural order: what Factorize_PosDef_Blocky asks for)."""
    rng = np.random.default_rng(sum(dims))
    n = 60
    d = np.where(rng.random(n) < 0.4, dims[0], dims[1])
    cs = np.concatenate([[0], np.cumsum(d)]).astype(np.int64)
    a, b = rng.integers(0, n, 90), rng.integers(0, n, 90)
    pairs = set(zip(range(n - 1), range(1, n))) | {(min(x, y), max(x, y)) for x, y in zip(a, b) if x != y}
    M = np.zeros((cs[-1], cs[-1]))
    for r, c in pairs:
        B = 0.3 * rng.standard_normal((d[r], d[c]))
        M[cs[r]:cs[r + 1], cs[c]:cs[c + 1]] = B
        M[cs[c]:cs[c + 1], cs[r]:cs[r + 1]] = B.T
    M += np.eye(cs[-1]) * (np.abs(M).sum(axis=1).max() + 1.0)      # well inside the positive definite cone
    bcol_ptr, brow, vals = [0], [], []
    for c in range(n):
        for r in range(c + 1):
            if r == c or (r, c) in pairs:
                brow.append(r)
                vals.append(M[cs[r]:cs[r + 1], cs[c]:cs[c + 1]].T.ravel())
        bcol_ptr.append(len(brow))
    lam = synth.BlockSystem(cs, np.asarray(bcol_ptr, dtype=np.int64), np.asarray(brow, dtype=np.int32), np.concatenate(vals),
                            rng.standard_normal(int(cs[-1])), 0)
    solver = CLinearSolver_HIP(natural_order=1)
    ok, st, l_values = solver.factorize(lam)
    assert ok and np.array_equal(st["perm"], np.arange(n)) and np.array_equal(st["dim"], d)
    Lm, _ = dense_factor(lam, st, l_values)
    Lref = np.linalg.cholesky(lam.to_scipy().toarray())
    assert np.abs(np.triu(Lm, 1)).max() == 0.0
    assert np.abs(Lm - Lref).max() < 1e-11 * np.abs(Lref).max()
    with pytest.raises(NotImplementedError):                 # the pieces of a column are only together in the caller's order
        CLinearSolver_HIP(natural_order=0).factorize(lam)


def random_system(seed):
    """Random connected block graph (a spanning chain plus random chords, some hubs), block dimensions uniform or a mix of
    2 .. 8, diagonally dominant symmetric positive definite values."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 400))
    dims = np.full(n, rng.choice([3, 6, 7])) if rng.random() < 0.6 else rng.integers(2, 9, n)
    chords = int(rng.integers(0, 3 * n))
    a, b = rng.integers(0, n, chords), rng.integers(0, n, chords)
    if rng.random() < 0.3 and n > 20:                       # a hub: a column with many blocks, a big separator
        hub = int(rng.integers(0, n))
        a = np.concatenate([a, np.full(n // 3, hub)])
        b = np.concatenate([b, rng.integers(0, n, n // 3)])
    v0 = np.concatenate([np.arange(n - 1), np.minimum(a, b)])
    v1 = np.concatenate([np.arange(1, n), np.maximum(a, b)])
    keep = v0 != v1
    lam = synth.structure_from_edges(dims, v0[keep], v1[keep])
    off = lam.block_value_offsets()
    vals = rng.standard_normal(lam.values.shape[0])
    col = np.repeat(np.arange(n), np.diff(lam.bcol_ptr))
    row_sum = np.zeros(int(lam.cumsum[-1]))                 # absolute row sums of the off-diagonal part, both triangles
    for k in range(lam.n_blocks):
        r, c = int(lam.brow_idx[k]), int(col[k])
        if r == c:
            continue
        blk = vals[off[k]:off[k + 1]].reshape(dims[c], dims[r]).T      # column-major dims[r] x dims[c]
        row_sum[lam.cumsum[r]:lam.cumsum[r + 1]] += np.abs(blk).sum(axis=1)
        row_sum[lam.cumsum[c]:lam.cumsum[c + 1]] += np.abs(blk).sum(axis=0)
    for v in range(n):
        k = int(lam.bcol_ptr[v + 1] - 1)
        d = int(dims[v])
        m = vals[off[k]:off[k + 1]].reshape(d, d)
        m = 0.5 * (m + m.T)
        m[np.arange(d), np.arange(d)] = np.abs(m).sum(axis=1) + row_sum[lam.cumsum[v]:lam.cumsum[v + 1]] + 1.0
        vals[off[k]:off[k + 1]] = m.ravel()
    lam.values[:] = vals
    lam.rhs[:] = rng.standard_normal(lam.rhs.shape[0])
    opts = {}
    if rng.random() < 0.5:
        opts = {"leaf_size": int(rng.integers(1, 9)), "subtree_size": int(rng.integers(1, 20)),
                "dense_top_nb": int(rng.choice([0, 2, 6, 24])), "nd_balance": int(rng.integers(5, 45))}
    return lam, opts


@pytest.mark.parametrize("seed", range(40))
def test_random_structures_match_oracle(seed):
    lam, opts = random_system(1000 + seed)
    ok, x_ref, _ = O.solve_sparse(lam)
    assert ok
    solver = CLinearSolver_HIP(**opts)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta), (seed, opts)
    assert rel_inf(eta, x_ref) < TOL, (seed, opts, lam.n_bcols)
    eta3 = lam.rhs.copy()
    assert solver.Solve_Again(eta3) and rel_inf(eta3, x_ref) < TOL


@pytest.mark.parametrize("name", ["chain6_n60", "chain3_n90", "chain7_n40", "sphere_8x8", "manhattan_n150"])
def test_marginals_match_reference(name):
    """Block diagonal of the covariance of a pose graph against what the reference's
    CMarginals::Calculate_DenseMarginals_Recurrent_FBS(.., mpart_Diagonal) returned (golden)."""
    from golden_util import load_golden
    lam, ref = load_golden(name)
    for opts in ({"dense_top_nb": 0}, {"dense_top_nb": 2, "dense_top_min_dim": 0}, {}):   # without / with a (forced) dense top / default
        solver = CLinearSolver_HIP(**opts)
        cov = solver.Marginals(lam)
        assert rel_inf(cov, ref["cov_diag"]) < TOL, opts
    err = np.abs(cov - ref["cov_diag"]).reshape(len(cov), -1).max(axis=1) / np.abs(ref["cov_diag"]).reshape(len(cov), -1).max(axis=1)
    assert err.max() < 1e-9
    eta = lam.rhs.copy()                                      # the factor the covariances left behind serves a solve
    assert solver.Solve_Again(eta) and rel_inf(eta, ref["x_cholmod_super"]) < TOL


@pytest.mark.parametrize("seed", range(12))
def test_marginals_random_structures(seed):
    lam, opts = random_system(3000 + seed)
    dims = np.diff(lam.cumsum)
    solver = CLinearSolver_HIP(**opts)
    if len(set(dims.tolist())) > 1 or int(dims[0]) not in (3, 6, 7):
        # any mix of block sizes up to 8 (round 4) -- with a dense top in the plan only the fixed block sizes
        full = np.linalg.inv(lam.to_scipy().toarray())
        try:
            cov = solver.Marginals(lam)
        except NotImplementedError:
            assert solver.stats()["schur_dim"] > 0
            return
        cs = lam.cumsum
        blocks = cov if isinstance(cov, list) else list(cov)
        for c in range(lam.n_bcols):
            assert np.abs(blocks[c] - full[cs[c]:cs[c + 1], cs[c]:cs[c + 1]]).max() < TOL * np.abs(full).max()
        return
    d = int(dims[0])
    full = np.linalg.inv(lam.to_scipy().toarray())
    ref = np.stack([full[d * c:d * c + d, d * c:d * c + d] for c in range(lam.n_bcols)])
    assert rel_inf(solver.Marginals(lam), ref) < TOL


@pytest.mark.parametrize("name", ["sphere", "manhattan"])
def test_marginals_with_dense_top(name):
    """C2- / C1-like graphs whose default plan has a dense top: the top's part of the inverse is a dense inverse on the
    matrix cores, the recursion continues below it.  Against numpy's inverse at a size it handles."""
    lam = synth.sphere(24, 24) if name == "sphere" else synth.manhattan(1200)
    solver = CLinearSolver_HIP() if name == "sphere" else CLinearSolver_HIP(dense_top_nb=8, dense_top_min_dim=0)
    cov = solver.Marginals(lam)
    assert solver.plan()["dense_dim"] > 0
    d = int(lam.cumsum[1])
    full = np.linalg.inv(lam.to_scipy().toarray())
    ref = np.stack([full[d * c:d * c + d, d * c:d * c + d] for c in range(lam.n_bcols)])
    assert rel_inf(cov, ref) < TOL
    err = np.abs(cov - ref).reshape(len(cov), -1).max(axis=1) / np.abs(ref).reshape(len(cov), -1).max(axis=1)
    assert err.max() < 1e-8
    eta = lam.rhs.copy()
    assert solver.Solve_Again(eta) and np.abs(lam.to_scipy() @ eta - lam.rhs).max() / np.abs(lam.rhs).max() < 1e-10


def test_marginals_full_size():
    lam = synth.sphere(50, 50)                               # C2: default plan with a dense top of 3712
    solver = CLinearSolver_HIP()
    cov = solver.Marginals(lam)
    assert solver.plan()["dense_dim"] > 0
    for c in (0, 1234, 2499):
        for j in (0, 5):
            e = np.zeros(lam.n_scalars)
            e[6 * c + j] = 1.0
            assert solver.Solve_PosDef_Blocky(lam, e)
            assert rel_inf(e[6 * c:6 * c + 6], cov[c][:, j]) < 1e-9
    lam = synth.pose_chain(n=100000)
    solver = CLinearSolver_HIP()
    cov = solver.Marginals(lam)
    for c in (0, 54321, 99999):
        for j in (0, 5):
            e = np.zeros(lam.n_scalars)
            e[6 * c + j] = 1.0
            assert solver.Solve_PosDef_Blocky(lam, e)
            assert rel_inf(e[6 * c:6 * c + 6], cov[c][:, j]) < 1e-9


@pytest.mark.parametrize("dims", [(6,), (3,), (7,)])
def test_separator_panels_agree_with_the_column_kernel(dims):
    """The separator stages as panels in LDS (panel_kernel.hip: external updates per factor block, then the task's columns
    inside an LDS image) against the column-by-column kernel (option panel = 0) and the oracle; same plan, different
    association of the sums."""
    d = dims[0]
    lam = synth.pose_chain(n=12000 if d == 6 else 6000, d=d, seed=21)
    ok, x_ref = O.solve_sparse(lam)[:2]
    assert ok
    xs = []
    for panel in (1, 0):
        solver = CLinearSolver_HIP(panel=panel)
        eta = lam.rhs.copy()
        assert solver.Solve_PosDef(lam, eta)
        assert rel_inf(eta, x_ref) < TOL
        xs.append(eta)
    assert rel_inf(xs[0], xs[1]) < 1e-11


@pytest.mark.parametrize("case", ["chain6", "chain3", "chain7", "sphere", "manhattan"])
def test_panel_tasks_as_rows_agree_with_the_block_wise_walk(case):
    """Option panel_rows = 1 (round 4; not the default: measured no faster): a level's block columns factored as rows -- one
    scalar row of the diagonal block, of the blocks below it, or the right-hand side per lane, multipliers by DPP row
    broadcast, no inverse of the diagonal block on the chain -- against the block-wise walk and the oracle.  The 2-D-like
    graphs have columns with more rows than one wave holds (several chunks per column, other waves, any order)."""
    lam = {"chain6": lambda: synth.pose_chain(n=12000, d=6, seed=21), "chain3": lambda: synth.pose_chain(n=6000, d=3, seed=22),
           "chain7": lambda: synth.pose_chain(n=6000, d=7, seed=23), "sphere": lambda: synth.sphere(50, 50),
           "manhattan": lambda: synth.manhattan(3500)}[case]()
    ok, x_ref = O.solve_sparse(lam)[:2]
    assert ok
    xs = []
    for rows in (1, 0):
        solver = CLinearSolver_HIP(panel_rows=rows)
        eta = lam.rhs.copy()
        assert solver.Solve_PosDef(lam, eta)
        assert rel_inf(eta, x_ref) < TOL
        xs.append(eta)
    assert rel_inf(xs[0], xs[1]) < 1e-10
    bad = type(lam)(lam.cumsum, lam.bcol_ptr, lam.brow_idx, lam.values.copy(), lam.rhs, lam.n_matrix_cut)
    off = bad.block_value_offsets()
    k = int(bad.bcol_ptr[bad.n_bcols // 2 + 1]) - 1          # a diagonal block in the middle: not positive definite
    d = int(bad.cumsum[1])
    bad.values[off[k]:off[k] + d * d] = -np.eye(d).ravel()
    assert CLinearSolver_HIP(panel_rows=1).Solve_PosDef(bad, bad.rhs.copy()) is False


@pytest.mark.parametrize("d", [6, 3, 7])
def test_lane_per_task_backward_substitution_and_lazy_inverses(d):
    """Option simt_backward = 1 (round 4; the default from ~12 000 leaf subtrees on): the leaf subtrees' backward substitution by
    backward_simt_kernel, which solves with L_jj^T itself, and a factorization that no longer stores inv(L_jj) for those columns
    -- until something asks for them: another right-hand side with the kept factor (the forward kernel multiplies by the
    inverses) and the covariances get them from a fix-up pass, and every later factorization stores them again."""
    lam = synth.pose_chain(n=9000, d=d, seed=31)
    ok, x_ref = O.solve_sparse(lam)[:2]
    assert ok
    solver = CLinearSolver_HIP(simt_backward=1, simt=1)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    again = 3.0 * lam.rhs
    assert solver.Solve_Again(again) and rel_inf(again, 3.0 * x_ref) < TOL        # inverses computed from the factor, late
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, x_ref) < TOL     # ... and stored by the factorization now
    again = -lam.rhs
    assert solver.Solve_Again(again) and rel_inf(again, -x_ref) < TOL
    cov = CLinearSolver_HIP(simt_backward=1, simt=1).Marginals(lam)
    cov_ref = CLinearSolver_HIP(simt_backward=0).Marginals(lam)
    assert rel_inf(cov, cov_ref) < 1e-10


@pytest.mark.parametrize("seed", range(6))
def test_marginals_mixed_block_sizes(seed):
    """Round 4: covariance blocks of graphs with more than one block size -- poses and landmarks, SE(2) and SE(3) vertices --
    as the reference's CMarginals::Calculate_DenseMarginals_Recurrent_FBS computes for any block matrix (Marginals.h:1694):
    the inverse subset's generic kernel against the diagonal blocks of the dense inverse."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(30, 200))
    dims = rng.choice([2, 3, 6, 7, 8] if seed % 2 else [3, 6], size=n)
    chords = int(rng.integers(n // 2, 2 * n))
    a, b = rng.integers(0, n, chords), rng.integers(0, n, chords)
    pairs = set(zip(range(n - 1), range(1, n))) | {(min(x, y), max(x, y)) for x, y in zip(a, b) if x != y}
    cs = np.concatenate([[0], np.cumsum(dims)]).astype(np.int64)
    M = np.zeros((cs[-1], cs[-1]))
    for r, c in pairs:
        B = 0.4 * rng.standard_normal((dims[r], dims[c]))
        M[cs[r]:cs[r + 1], cs[c]:cs[c + 1]] = B
        M[cs[c]:cs[c + 1], cs[r]:cs[r + 1]] = B.T
    M += np.eye(cs[-1]) * (np.abs(M).sum(axis=1).max() * 0.6 + 1.0)
    assert np.linalg.eigvalsh(M).min() > 0
    bcol_ptr, brow, vals = [0], [], []
    for c in range(n):
        for r in range(c + 1):
            if r == c or (r, c) in pairs:
                brow.append(r)
                vals.append(M[cs[r]:cs[r + 1], cs[c]:cs[c + 1]].T.ravel())
        bcol_ptr.append(len(brow))
    lam = synth.BlockSystem(cs, np.asarray(bcol_ptr, dtype=np.int64), np.asarray(brow, dtype=np.int32), np.concatenate(vals),
                            rng.standard_normal(int(cs[-1])), 0)
    cov = CLinearSolver_HIP(dense_top_nb=0).Marginals(lam)
    Minv = np.linalg.inv(M)
    assert isinstance(cov, list) and len(cov) == n
    scale = np.abs(Minv).max()
    for c in range(n):
        assert cov[c].shape == (dims[c], dims[c])
        assert np.abs(cov[c] - Minv[cs[c]:cs[c + 1], cs[c]:cs[c + 1]]).max() < 1e-10 * scale


@pytest.mark.parametrize("case", ["chain6", "chain3", "chain7", "sphere", "band_S"])
def test_panel_tasks_hand_their_contributions_up(case):
    """Option panel_handup (round 4, the default): a panel task computes what it owes the tasks of the next stage -- the products
    of its own finished blocks -- out of its LDS image and hands them up as ready-made blocks; the task above subtracts them
    with the loads that fetch its image, instead of fetching both operands of every such product itself.  Same answer as
    with the option off, and as the oracle's."""
    if case == "band_S":       # shaped like the reduced camera system of a band-visibility BA problem: separators of three columns
        n, d = 600, 6
        rng = np.random.default_rng(5)
        cols = [sorted({max(c - j, 0) for j in (0, 1, 2, 3)}) for c in range(n)]
        bcol_ptr = np.concatenate([[0], np.cumsum([len(r) for r in cols])]).astype(np.int64)
        brow = np.concatenate(cols).astype(np.int32)
        vals = []
        for c in range(n):
            for r in cols[c]:
                B = 0.1 * rng.standard_normal((d, d))
                vals.append(((B + B.T) * 0.05 + 4.0 * np.eye(d)).T.ravel() if r == c else B.T.ravel())
        lam = synth.BlockSystem(np.arange(n + 1, dtype=np.int64) * d, bcol_ptr, brow, np.concatenate(vals), rng.standard_normal(n * d), 0)
        opts = {"subtree_size": 4, "simt": 0}
    else:
        lam = {"chain6": lambda: synth.pose_chain(n=12000, d=6, seed=21), "chain3": lambda: synth.pose_chain(n=6000, d=3, seed=22),
               "chain7": lambda: synth.pose_chain(n=6000, d=7, seed=23), "sphere": lambda: synth.sphere(50, 50)}[case]()
        opts = {}
    ok, x_ref = O.solve_sparse(lam)[:2]
    assert ok
    xs = []
    for up in (1, 0):
        solver = CLinearSolver_HIP(panel_handup=up, **opts)
        for _ in range(2):                                   # (the hand-up buffer is written anew by every factorization)
            eta = lam.rhs.copy()
            assert solver.Solve_PosDef_Blocky(lam, eta)
            assert rel_inf(eta, x_ref) < TOL
        xs.append(eta)
    assert rel_inf(xs[0], xs[1]) < 1e-10


# ---- K value sets in one pass of launches (slampp_hip_factor_solve_batch_device_async; SURVEY.md section 8e: "replicas only
# (multiple independent problems / damping values per GPU)"; the reference's LM loop re-damps and re-solves one value after
# the other, NonlinearSolver_Lambda_LM.h:967-1001, 1660-1676) -------------------------------------------------------------------

def _damped(lam, alpha):
    """Lambda + alpha I on a copy (what ApplyDamping of the reference's LM solver does, NonlinearSolver_Lambda_LM.h:228-239)."""
    import dataclasses
    off = lam.block_value_offsets()
    v = lam.values.copy()
    for j in range(lam.n_bcols):
        k = int(lam.bcol_ptr[j + 1] - 1)
        d = int(lam.cumsum[j + 1] - lam.cumsum[j])
        v[off[k]:off[k + 1]].reshape(d, d)[...] += alpha * np.eye(d)
    return dataclasses.replace(lam, values=v)


BATCH_CASES = {
    "chain6": (lambda: synth.pose_chain(n=3000, d=6, seed=41), {}),                    # lane-per-task leaves, slices of the tree as panels
    "chain3": (lambda: synth.pose_chain(n=2500, d=3, seed=42), {}),
    "chain7_columns": (lambda: synth.pose_chain(n=1200, d=7, seed=43), {"panel": 0}),  # separators column by column
    "chain6_waves": (lambda: synth.pose_chain(n=2000, d=6, seed=44), {"simt": 0}),     # a wave per leaf subtree
    "sphere_dense_top": (lambda: synth.sphere(24, 24, seed=45), {}),                   # a dense top: the members one after the other
    "mixed_sizes": (lambda: synth.ba(30, 1500, seed=46), {}),                          # 6 / 3 blocks through the sparse path
}


@pytest.mark.parametrize("pad", [0, 6, 3])     # member strides: tight, padded and even, odd (16-byte loads impossible: one by one)
@pytest.mark.parametrize("name", sorted(BATCH_CASES))
def test_batch_of_damped_systems_matches_the_oracle_member_by_member(name, pad):
    import torch
    make, opts = BATCH_CASES[name]
    lam = make()
    alphas = [0.0, 1e-3, 0.5, 7.0, 1e-6]
    members = [_damped(lam, a) for a in alphas]
    K = len(members)
    sv, sr = lam.values.shape[0] + pad, lam.n_scalars + pad
    sv += sv % 2 if pad != 3 else (1 - sv % 2)     # even strides, or (pad == 3) odd ones on purpose
    sr += sr % 2 if pad != 3 else (1 - sr % 2)
    vals = torch.zeros(K * sv, dtype=torch.float64, device="cuda")
    rhs = torch.zeros(K * sr, dtype=torch.float64, device="cuda")
    for k, m in enumerate(members):
        vals[k * sv:k * sv + m.values.shape[0]] = torch.from_numpy(m.values).cuda()
        rhs[k * sr:k * sr + m.n_scalars] = torch.from_numpy((k + 1.0) * m.rhs).cuda()
    torch.cuda.synchronize()
    solver = CLinearSolver_HIP(**opts)
    assert solver.SymbolicDecomposition_Blocky(lam)
    solver.factor_solve_batch_device_async(K, vals.data_ptr(), sv, rhs.data_ptr(), sr)
    assert solver.sync_batch(K) == [True] * K
    x = rhs.cpu().numpy()
    for k, m in enumerate(members):
        ok, x_ref, _ = O.solve_sparse(m)
        assert ok and rel_inf(x[k * sr:k * sr + m.n_scalars], (k + 1.0) * x_ref) < TOL, (name, k)
    # the handle still solves single systems afterwards (its own factor was not the batch's)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, O.solve_sparse(lam)[1]) < TOL


def test_batch_member_that_is_not_positive_definite_fails_alone():
    import torch
    lam = synth.pose_chain(n=2000, d=6, seed=47)
    bad = _damped(lam, -40.0)
    assert O.solve_sparse(bad)[0] is False
    members = [lam, bad, _damped(lam, 0.25)]
    K, sv, sr = 3, lam.values.shape[0] + lam.values.shape[0] % 2, lam.n_scalars
    vals = torch.stack([torch.from_numpy(np.pad(m.values, (0, sv - m.values.shape[0]))) for m in members]).cuda().contiguous()
    rhs = torch.stack([torch.from_numpy(m.rhs) for m in members]).cuda().contiguous()
    solver = CLinearSolver_HIP()
    assert solver.SymbolicDecomposition_Blocky(lam)
    for _ in range(2):                         # twice: the flags of the first round must not leak into the second
        r = rhs.clone()
        torch.cuda.synchronize()
        solver.factor_solve_batch_device_async(K, vals.data_ptr(), sv, r.data_ptr(), sr)
        assert solver.sync_batch(K) == [True, False, True]
        x = r.cpu().numpy()
        for k in (0, 2):
            assert rel_inf(x[k], O.solve_sparse(members[k])[1]) < TOL
    good = torch.stack([torch.from_numpy(m.values) for m in (lam, lam)]).cuda().contiguous()
    if good.shape[1] % 2 == 0:
        r = rhs[:2].clone()
        torch.cuda.synchronize()
        solver.factor_solve_batch_device_async(2, good.data_ptr(), good.shape[1], r.data_ptr(), sr)
        assert solver.sync_batch(2) == [True, True]


def test_batch_solved_member_by_member_leaves_no_stale_factor():
    """Advisor, round 5: with a dense top the members go through the handle's own factor arrays one after the other.  The
    handle must not pass the last member's factor -- here a failed one -- off as its own: solve_again refuses until the next
    factorization; a batch of one IS the handle's factorization, dropped when sync_batch finds it not positive definite."""
    import torch
    lam = synth.sphere(24, 24, seed=45)             # a dense top (BATCH_CASES["sphere_dense_top"])
    bad = _damped(lam, -40.0)
    assert O.solve_sparse(bad)[0] is False
    members = [lam, _damped(lam, 0.5), bad]
    sv, sr = lam.values.shape[0] + lam.values.shape[0] % 2, lam.n_scalars + lam.n_scalars % 2
    vals = torch.stack([torch.from_numpy(np.pad(m.values, (0, sv - m.values.shape[0]))) for m in members]).cuda().contiguous()
    rhs = torch.stack([torch.from_numpy(np.pad(m.rhs, (0, sr - m.n_scalars))) for m in members]).cuda().contiguous()
    solver = CLinearSolver_HIP()
    assert solver.SymbolicDecomposition_Blocky(lam)
    assert solver.stats()["schur_dim"] > 0        # (sparse mode: the dimension of the dense top)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef_Blocky(lam, eta)     # the handle has a factor of its own
    again = lam.rhs.copy()
    assert solver.Solve_Again(again) and rel_inf(again, eta) < TOL
    torch.cuda.synchronize()
    solver.factor_solve_batch_device_async(3, vals.data_ptr(), sv, rhs.data_ptr(), sr)
    assert solver.sync_batch(3) == [True, True, False]
    x = rhs.cpu().numpy()
    for k in (0, 1):
        assert rel_inf(x[k][:lam.n_scalars], O.solve_sparse(members[k])[1]) < TOL
    with pytest.raises(Exception):
        solver.Solve_Again(lam.rhs.copy())          # no factor: neither the failed member's nor a stale one
    assert solver.Solve_PosDef_Blocky(lam, eta)     # the next factorization brings one back
    assert solver.Solve_Again(again)
    # a batch of one that fails: an ordinary factorization that failed
    r = rhs[2:3].clone()
    torch.cuda.synchronize()
    solver.factor_solve_batch_device_async(1, vals[2:3].data_ptr(), sv, r.data_ptr(), sr)
    assert solver.sync_batch(1) == [False]
    with pytest.raises(Exception):
        solver.Solve_Again(lam.rhs.copy())
    # and one that succeeds: its factor is the handle's
    r = rhs[1:2].clone()
    torch.cuda.synchronize()
    solver.factor_solve_batch_device_async(1, vals[1:2].data_ptr(), sv, r.data_ptr(), sr)
    assert solver.sync_batch(1) == [True]
    b = members[1].rhs.copy()
    assert solver.Solve_Again(b)
    assert rel_inf(b, O.solve_sparse(members[1])[1]) < TOL


def test_batch_is_refused_where_it_does_not_apply():
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    ba = synth.ba(20, 600, seed=3)
    schur = CLinearSolver_Schur_HIP()
    assert schur.SymbolicDecomposition_Blocky(ba)
    with pytest.raises(Exception):
        schur.factor_solve_batch_device_async(2, 8, ba.values.shape[0], 8, ba.n_scalars)
    lam = synth.pose_chain(n=50, d=6)
    solver = CLinearSolver_HIP()
    assert solver.SymbolicDecomposition_Blocky(lam)
    with pytest.raises(Exception):
        solver.factor_solve_batch_device_async(2, 8, lam.values.shape[0] - 1, 8, lam.n_scalars)   # overlapping members
    with pytest.raises(Exception):
        solver.factor_solve_batch_device_async(65, 8, lam.values.shape[0], 8, lam.n_scalars)      # more than SLAMPP_HIP_MAX_BATCH
