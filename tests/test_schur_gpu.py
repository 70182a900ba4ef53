"""GPU parity tests of the BA path (Schur complement + dense reduced camera system) against the CPU
oracle's restatement of CLinearSolver_Schur (oracle/slampp_oracle.c), through the C ABI.
Tolerance: ||x_gpu - x_ref||_inf / ||x_ref||_inf < 1e-10 (BASELINE.json north_star, fp64)."""
import dataclasses

import numpy as np
import pytest

from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel_inf(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max())


CASES = {
    "band_k4": lambda: synth.ba(50, 3000, k=4, mode="band"),
    "uniform_dense_S": lambda: synth.ba(40, 5000, k=4, mode="uniform"),
    "venice_ragged": lambda: synth.ba(60, 3000, mode="venice"),
    "k1_single_obs": lambda: synth.ba(30, 500, k=1),
    "sim3_7x7": lambda: synth.ba(25, 1500, k=3, cam_dim=7, pt_dim=3, seed=11),
    "se2_3x2": lambda: synth.ba(30, 1000, k=3, cam_dim=3, pt_dim=2, seed=12),
    "n63_fits_one_tile": lambda: synth.ba(9, 300, k=3, cam_dim=7, pt_dim=3, seed=13),   # N = 63: rhs row is row 63
    "n126": lambda: synth.ba(21, 800, k=4),                                          # N = 126 -> padded to 128
    "n192_tile_multiple": lambda: synth.ba(32, 1000, k=4),                            # N = 192 -> a whole extra tile
    "two_cams": lambda: synth.ba(2, 50, k=2),
    # feature tracks over consecutive cameras, 2 - 30 of them: the tracks born at one camera form prefix runs (round 4)
    "tracks_prefix_runs": lambda: synth.ba(150, 3000, mode="tracks", seed=9),
    "tracks_sim3": lambda: synth.ba(90, 1200, mode="tracks", cam_dim=7, pt_dim=3, seed=10),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_parity_with_oracle(name):
    lam = CASES[name]()
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    solver = CLinearSolver_Schur_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    assert rel_inf(eta[:n_x], x_ref[:n_x]) < TOL      # dx (cameras)
    assert rel_inf(eta[n_x:], x_ref[n_x:]) < TOL      # dl (landmarks)
    eta2 = -0.5 * lam.rhs                               # warm: structure cached
    assert solver.Solve_PosDef_Blocky(lam, eta2)
    assert rel_inf(eta2, -0.5 * x_ref) < TOL


@pytest.mark.parametrize("sparse", [0, 1])
@pytest.mark.parametrize("tiles", [0, 1, 2, 3])
@pytest.mark.parametrize("name", sorted(CASES))
def test_landmark_tiles_and_contribution_lists_agree_with_oracle(name, tiles, sparse):
    """The reduced system assembled landmark-major (schur_tiles.hip: every landmark read once; 1 = runs of landmarks with
    the same cameras on the matrix cores + LDS tiles for the rest, 2 = tiles only, 3 = runs only, of any length), from the
    contribution lists (0), or -- venice_ragged with tiles: landmarks seen by more than ten cameras do not fit a tile --
    by both; into the dense buffer or the inner sparse solver's values."""
    lam = CASES[name]()
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    solver = CLinearSolver_Schur_HIP(schur_tiles=tiles, schur_sparse=sparse)
    solver.set_option("profile", 1)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL
    eta2 = 3.0 * lam.rhs
    assert solver.Solve_PosDef_Blocky(lam, eta2)
    assert rel_inf(eta2, 3.0 * x_ref) < TOL
    phases = solver.profile()
    assert ("schur_tiles" in phases) == bool(tiles), phases
    if name == "venice_ragged" and tiles == 2:
        assert phases["schur_gather"][1] > 0.0                  # the heavy landmarks went through the lists


def test_tiles_chosen_where_landmarks_share_cameras():
    """Default (auto): band visibility -> tiles; random visibility -> lists (a tile would hold one block per contribution)."""
    for mode, expect in (("band", True), ("uniform", False)):
        lam = synth.ba(200, 40000, k=4, mode=mode)
        solver = CLinearSolver_Schur_HIP()
        solver.set_option("profile", 1)
        eta = lam.rhs.copy()
        assert solver.Solve_PosDef(lam, eta)
        assert ("schur_tiles" in solver.profile()) == expect, mode
        ok, x_ref, _, _ = O.solve_schur(lam)
        assert ok and rel_inf(eta, x_ref) < TOL


def test_not_positive_definite_landmark_is_reported_from_a_tile():
    lam = synth.ba(10, 100, k=3)
    off = lam.block_value_offsets()
    last = lam.n_blocks - 1
    vals = lam.values.copy()
    vals[off[last]:off[last + 1]] -= 50.0 * np.eye(3).ravel()
    bad = dataclasses.replace(lam, values=vals)
    eta = bad.rhs.copy()
    assert CLinearSolver_Schur_HIP(schur_tiles=1).Solve_PosDef(bad, eta) is False


def test_schur_and_sparse_paths_agree():
    lam = synth.ba(40, 2000, mode="venice", seed=3)
    a, b = lam.rhs.copy(), lam.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(lam, a)
    assert CLinearSolver_HIP().Solve_PosDef(lam, b)
    assert rel_inf(a, b) < TOL


def test_indefinite_landmark_block_returns_false():
    lam = synth.ba(10, 100, k=3)
    off = lam.block_value_offsets()
    last = lam.n_blocks - 1                      # C block of the last point
    vals = lam.values.copy()
    vals[off[last]:off[last + 1]] -= 50.0 * np.eye(3).ravel()
    bad = dataclasses.replace(lam, values=vals)
    assert O.solve_schur(bad)[0] is False
    eta = bad.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(bad, eta) is False


def test_indefinite_reduced_system_returns_false():
    lam = synth.ba(10, 100, k=3)
    off = lam.block_value_offsets()
    vals = lam.values.copy()
    vals[off[4]:off[5]] -= 1e4 * np.eye(6).ravel()   # camera 4's diagonal block
    bad = dataclasses.replace(lam, values=vals)
    assert O.solve_schur(bad)[0] is False
    eta = bad.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(bad, eta) is False


def test_full_size_c4_residual_and_linearity():
    """BASELINE config C4 (1k cams x 500k points): size-independent checks -- the residual of the
    full system and linearity in eta."""
    lam = synth.ba(1000, 500_000, k=4, mode="band")
    A = lam.to_scipy()
    solver = CLinearSolver_Schur_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert np.abs(A @ eta - lam.rhs).max() / np.abs(lam.rhs).max() < 1e-9
    eta2 = 4.0 * lam.rhs
    assert solver.Solve_PosDef_Blocky(lam, eta2)
    assert rel_inf(eta2, 4.0 * eta) < TOL


@pytest.mark.parametrize("sparse,know_ranks", [(0, False), (1, False), (1, True), (0, True)])
def test_two_landmark_shards_with_summing_callback(sparse, know_ranks):
    """Two ranks of the landmark-sharded path in one process (two solver handles on the same GPU, one thread each):
    the all-reduce callback is a barrier + sum, so the whole multi-GPU code path runs -- the agreement on the union
    of nonzero S blocks, the packed exchange, redundant dense solves -- and must reproduce the unsharded solution."""
    import threading
    import torch
    from slam_plus_plus_amd import sharding

    torch.zeros(1, device="cuda")                         # torch's lazy CUDA init belongs on the main thread
    lam = synth.ba(40, 4000, mode="venice", seed=21)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    world = 2
    barrier = threading.Barrier(world)
    slots, total, counts, errors = [None] * world, [None], [], []

    class DevPtr:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def make_fn(rank):
        def fn(ptr, count, stream):
            try:
                torch.cuda.synchronize()                  # the solver's stream has produced the partial sums
                slots[rank] = torch.as_tensor(DevPtr(ptr, count), device="cuda")
                barrier.wait()
                if rank == 0:
                    assert all(t.numel() == count for t in slots)
                    total[0] = torch.stack(slots).sum(dim=0)
                    counts.append(count)
                barrier.wait()
                slots[rank].copy_(total[0])
                torch.cuda.synchronize()
                barrier.wait()
                return 0
            except Exception as e:                        # pragma: no cover
                import traceback
                errors.append(traceback.format_exc())
                raise
        return fn

    results = [None] * world

    def run(rank):
        try:
            shard, sl = sharding.landmark_shard(lam, rank, world)
            solver = CLinearSolver_Schur_HIP(schur_sparse=sparse)
            if know_ranks:                                # block lists are concatenated instead of an indicator summed
                solver.set_option("shard_rank", rank)
                solver.set_option("shard_world", world)
            solver.set_allreduce(make_fn(rank))
            eta = shard.rhs.copy()
            assert solver.Solve_PosDef(shard, eta)
            eta2 = shard.rhs.copy()                       # second step: the agreed block list is reused
            assert solver.Solve_PosDef_Blocky(shard, eta2)
            assert np.array_equal(eta, eta2)
            results[rank] = (eta, sl)
        except Exception as e:                            # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(timeout=120) for t in threads]
    assert not errors, errors
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    x = np.zeros_like(x_ref)
    for eta, sl in results:
        x[sl] = eta[n_x:]
    x[:n_x] = results[0][0][:n_x]
    assert rel_inf(results[1][0][:n_x], results[0][0][:n_x]) < 1e-13      # dx is computed redundantly
    assert rel_inf(x, x_ref) < TOL
    # what travelled: first the indicator over the camera-block triangle, then packed blocks -- never the dense square
    nc, N = lam.n_matrix_cut, n_x
    if know_ranks:
        assert counts[0] == world and counts[1] <= world * (nc * (nc + 1) // 2)   # lengths, then the concatenated lists
        counts = counts[1:]
    else:
        assert counts[0] == nc * (nc + 1) // 2
    assert all(c < (N + 64) ** 2 // 2 for c in counts[1:]) and (counts[1] - N) % 36 == 0


@pytest.mark.parametrize("mode,k", [("band", 4), ("venice", 6)])
@pytest.mark.parametrize("sparse", [0, 1])
def test_reduced_system_sparse_or_dense_same_answer(mode, k, sparse):
    """The reduced camera system through the dense MFMA factorization and through the sparse block path."""
    lam = synth.ba(150, 6000, k=k, mode=mode, seed=31)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    solver = CLinearSolver_Schur_HIP(schur_sparse=sparse, profile=1)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL
    eta2 = 3.0 * lam.rhs
    assert solver.Solve_PosDef_Blocky(lam, eta2)
    assert rel_inf(eta2, 3.0 * x_ref) < TOL
    phases = solver.profile()
    assert ("reduced_sparse" in phases) == bool(sparse) and ("dense_chol" in phases) != bool(sparse)


def test_reduced_system_auto_picks_sparse_for_band_structure():
    lam = synth.ba(500, 10000, k=4, mode="band", seed=32)      # under 3 % of the camera pairs share a point
    solver = CLinearSolver_Schur_HIP(profile=1)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert "reduced_sparse" in solver.profile()
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok and rel_inf(eta, x_ref) < TOL
    # and an indefinite reduced system is reported from inside the sparse factorization
    bad = dataclasses.replace(lam, values=lam.values.copy())
    off = bad.block_value_offsets()
    k = int(bad.bcol_ptr[1] - 1)                                 # diagonal block of camera 0
    bad.values[off[k]:off[k] + 36] *= -1.0
    assert not solver.Solve_PosDef_Blocky(bad, bad.rhs.copy())
    assert solver.Solve_PosDef_Blocky(lam, lam.rhs.copy())       # and the solver recovers


@pytest.mark.parametrize("cam_dim,pt_dim", [(7, 3), (3, 2)])
def test_sparse_reduced_system_other_block_sizes(cam_dim, pt_dim):
    """Sim(3)-style 7x7 cameras and the planar (3, 2) case through the sparse reduced system (inner solver kernels for
    7x7 / 3x3 blocks) against the dense one and the oracle."""
    lam = synth.ba(300, 5000, k=4, mode="band", seed=41, cam_dim=cam_dim, pt_dim=pt_dim)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    out = {}
    for sparse in (0, 1):
        solver = CLinearSolver_Schur_HIP(schur_sparse=sparse)
        eta = lam.rhs.copy()
        assert solver.Solve_PosDef(lam, eta)
        assert rel_inf(eta, x_ref) < TOL
        out[sparse] = eta
    assert rel_inf(out[1], out[0]) < 1e-11


def test_many_cameras_sparse_reduced_system():
    """10 000 cameras: the dense reduced system would be a 60 000 x 60 000 buffer (28.8 GB); the sparse one is a few
    megabytes.  Also the comparison-sort branch of the contribution lists (camera-pair key space above 2^26)."""
    lam = synth.ba(10000, 40000, k=4, mode="band", seed=51)
    solver = CLinearSolver_Schur_HIP(profile=1)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert "reduced_sparse" in solver.profile()
    assert solver.stats()["device_bytes"] < 1 << 30
    resid = np.abs(lam.to_scipy() @ eta - lam.rhs).max() / np.abs(lam.rhs).max()
    assert resid < 1e-10
    # the covariances at this camera count: only through the sparse inverse subset (the dense inverse would take two
    # 28.8 GB buffers); checked against columns of the inverse obtained by solving with unit vectors
    cams, pts = solver.Schur_Marginals(lam)
    assert solver.stats()["device_bytes"] < 1 << 30
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    for (idx, blk, d, base) in ((1234, cams, 6, 0), (9999, cams, 6, 0), (0, pts, 3, n_x), (39999, pts, 3, n_x)):
        for j in (0, d - 1):
            e = np.zeros(lam.n_scalars)
            e[base + d * idx + j] = 1.0
            assert solver.Solve_PosDef_Blocky(lam, e)
            assert rel_inf(e[base + d * idx:base + d * idx + d], blk[idx][:, j]) < 1e-9


@pytest.mark.parametrize("name", ["ba_12x150_venice", "ba_10x120_band"])
def test_marginal_poses_matches_reference(name):
    """Solve_PosDef_Blocky_MarginalPoses: landmarks only, poses zeroed -- against the reference's output (golden)."""
    from golden_util import load_golden
    lam, ref = load_golden(name)
    solver = CLinearSolver_Schur_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef_Blocky_MarginalPoses(lam, eta)
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    assert np.all(eta[:n_x] == 0.0)
    assert rel_inf(eta, ref["x_schur_marginal_poses"]) < TOL
    # the full solve still works afterwards on the same handle
    eta2 = lam.rhs.copy()
    assert solver.Solve_PosDef_Blocky(lam, eta2) and rel_inf(eta2, ref["x_schur"]) < TOL


def test_marginal_poses_needs_schur_mode():
    lam = synth.pose_chain(n=50, d=6)
    with pytest.raises(NotImplementedError):
        CLinearSolver_Schur_HIP().Solve_PosDef_Blocky_MarginalPoses(lam, lam.rhs.copy())


@pytest.mark.parametrize("name", ["ba_12x150_venice", "ba_10x120_band"])
def test_schur_marginals_match_reference(name):
    """Block diagonal of the covariance against CSchurComplement_Marginals::Schur_Marginals' output (golden)."""
    from golden_util import load_golden
    lam, ref = load_golden(name)
    solver = CLinearSolver_Schur_HIP()
    cams, pts = solver.Schur_Marginals(lam)
    assert rel_inf(cams, ref["cam_cov"]) < TOL
    assert rel_inf(pts, ref["lm_cov"]) < TOL
    # per block as well: small blocks must not hide behind the largest one
    for got, want in ((cams, ref["cam_cov"]), (pts, ref["lm_cov"])):
        err = np.abs(got - want).reshape(len(got), -1).max(axis=1) / np.abs(want).reshape(len(got), -1).max(axis=1)
        assert err.max() < 1e-9
    # solving on the same handle before and after is unaffected
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, ref["x_schur"]) < TOL
    _, pts2 = solver.Schur_Marginals(lam, b_do_cam_marginals=False)
    assert np.array_equal(pts, pts2)


MARGINAL_CASES = {
    "band_sparse_reduced": (lambda: synth.ba(200, 6000, k=4, mode="band", seed=21), {}),     # sparse S: sparse inverse subset
    "band_sparse_S_dense_inverse": (lambda: synth.ba(200, 6000, k=4, mode="band", seed=21), {"marginals_dense": 1}),
    "forced_sparse_uniform": (lambda: synth.ba(40, 3000, k=4, mode="uniform", seed=22), {"schur_sparse": 1}),   # S full, still the subset
    "forced_sparse_venice_7x7": (lambda: synth.ba(60, 2000, mode="venice", seed=29, cam_dim=7, pt_dim=3), {"schur_sparse": 1}),
    "forced_sparse_3x2": (lambda: synth.ba(150, 2500, k=3, cam_dim=3, pt_dim=2, seed=30), {"schur_sparse": 1}),
    "uniform_dense_S": (lambda: synth.ba(40, 3000, k=4, mode="uniform", seed=22), {}),
    "venice_ragged": (lambda: synth.ba(70, 2500, mode="venice", seed=23), {}),
    "sim3_7x7": (lambda: synth.ba(25, 1200, k=3, cam_dim=7, pt_dim=3, seed=24), {}),
    "se2_3x2": (lambda: synth.ba(30, 900, k=3, cam_dim=3, pt_dim=2, seed=25), {}),
    "k1_single_obs": (lambda: synth.ba(30, 400, k=1, seed=26), {}),
    "n63_one_tile": (lambda: synth.ba(9, 300, k=3, cam_dim=7, pt_dim=3, seed=27), {}),
    "n192_tile_multiple": (lambda: synth.ba(32, 900, k=4, seed=28), {}),
}


@pytest.mark.parametrize("name", sorted(MARGINAL_CASES))
def test_schur_marginals_parity_with_oracle(name):
    make, opts = MARGINAL_CASES[name]
    lam = make()
    cams_ref, pts_ref = O.schur_marginals(lam)
    solver = CLinearSolver_Schur_HIP(**opts)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    cams, pts = solver.Schur_Marginals(lam)
    assert rel_inf(cams, cams_ref) < TOL and rel_inf(pts, pts_ref) < TOL
    assert rel_inf(cams, cams.transpose(0, 2, 1)) < 1e-13 and rel_inf(pts, pts.transpose(0, 2, 1)) < 1e-13
    prof = solver.profile() if opts.get("profile") else {}
    # and the solve afterwards still works on the same handle (the inner factor was redone by the covariances)
    eta2 = lam.rhs.copy()
    assert solver.Solve_PosDef_Blocky(lam, eta2) and rel_inf(eta2, eta) < 1e-12


def test_schur_marginals_full_size_properties():
    """C4-sized: no oracle at this size; the identity Sigma_pp = C_p^-1 + W_p^T Sigma_cc' W_p is checked through
    Lambda Sigma = I on sampled block columns instead: Lambda_pp Sigma_pp + sum_c U_cp^T Sigma_cp = I needs the
    off-diagonal blocks, so the check uses solves -- column j of Sigma is the solution of Lambda x = e_j."""
    lam = synth.ba(300, 60000, k=4, mode="band", seed=31)
    solver = CLinearSolver_Schur_HIP()
    cams, pts = solver.Schur_Marginals(lam)
    nc = lam.n_matrix_cut
    nx = int(lam.cumsum[nc])
    rng = np.random.default_rng(0)
    for c in rng.integers(0, nc, 3):
        for j in range(6):
            e = np.zeros(lam.n_scalars)
            e[6 * c + j] = 1.0
            assert solver.Solve_PosDef_Blocky(lam, e)
            assert rel_inf(e[6 * c:6 * c + 6], cams[c][:, j]) < 1e-9
    for p in rng.integers(0, lam.n_bcols - nc, 3):
        for j in range(3):
            e = np.zeros(lam.n_scalars)
            e[nx + 3 * p + j] = 1.0
            assert solver.Solve_PosDef_Blocky(lam, e)
            assert rel_inf(e[nx + 3 * p:nx + 3 * p + 3], pts[p][:, j]) < 1e-9


def test_schur_marginals_not_posdef_and_wrong_mode():
    lam = synth.ba(20, 400, k=3, seed=41)
    bad = dataclasses.replace(lam, values=lam.values.copy())
    off = lam.block_value_offsets()
    k = int(lam.bcol_ptr[3 + 1] - 1)        # diagonal block of camera 3
    bad.values[off[k]:off[k + 1]] *= -1.0
    with pytest.raises(ArithmeticError):
        CLinearSolver_Schur_HIP().Schur_Marginals(bad)
    chain = synth.pose_chain(n=50, d=6)
    with pytest.raises((NotImplementedError, ValueError)):
        CLinearSolver_Schur_HIP().Schur_Marginals(chain)


@pytest.mark.parametrize("sparse", [0, 1])
def test_schur_marginals_two_landmark_shards(sparse):
    """The covariances with the landmarks split over two ranks (threads, as above): the reduced system is summed through
    the callback (the dense buffer, or the packed blocks of the sparse reduced system), every rank gets all camera
    blocks and the blocks of its own landmarks."""
    import threading
    import torch
    from slam_plus_plus_amd import sharding

    torch.zeros(1, device="cuda")
    lam = synth.ba(30, 2400, mode="venice", seed=61)
    cams_ref, pts_ref = O.schur_marginals(lam)
    world = 2
    barrier = threading.Barrier(world)
    slots, total, errors, results = [None] * world, [None], [], [None] * world

    class DevPtr:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def make_fn(rank):
        def fn(ptr, count, stream):
            torch.cuda.synchronize()
            slots[rank] = torch.as_tensor(DevPtr(ptr, count), device="cuda")
            barrier.wait()
            if rank == 0:
                total[0] = torch.stack(slots).sum(dim=0)
            barrier.wait()
            slots[rank].copy_(total[0])
            torch.cuda.synchronize()
            barrier.wait()
            return 0
        return fn

    def run(rank):
        try:
            shard, sl = sharding.landmark_shard(lam, rank, world)
            solver = CLinearSolver_Schur_HIP(schur_sparse=sparse)
            solver.set_allreduce(make_fn(rank))
            cams, pts = solver.Schur_Marginals(shard)
            eta = shard.rhs.copy()                        # and a solve on the same handle afterwards
            assert solver.Solve_PosDef(shard, eta)
            results[rank] = (cams, pts, sl)
        except Exception:                                 # pragma: no cover
            import traceback
            errors.append(traceback.format_exc())
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(timeout=120) for t in threads]
    assert not errors, errors
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    pts = np.zeros_like(pts_ref)
    for cams, p, sl in results:
        assert rel_inf(cams, cams_ref) < TOL
        pts[(sl.start - n_x) // 3:(sl.stop - n_x) // 3] = p
    assert rel_inf(pts, pts_ref) < TOL


@pytest.mark.parametrize("seed", range(24))
def test_random_ba_systems_match_oracle(seed):
    """Random sizes, visibility patterns, block sizes and reduced-system modes through the Schur path -- solve and
    covariances against the oracle."""
    rng = np.random.default_rng(500 + seed)
    cam_dim, pt_dim = [(6, 3), (7, 3), (3, 2)][int(rng.integers(0, 3))]
    nc = int(rng.integers(2, 140))
    n_pts = int(rng.integers(1, 2500))
    mode = ["band", "uniform", "venice"][int(rng.integers(0, 3))]
    k = int(rng.integers(1, 7))
    lam = synth.ba(nc, n_pts, k=k, mode=mode, seed=900 + seed, cam_dim=cam_dim, pt_dim=pt_dim)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    sparse = int(rng.integers(-1, 2))
    solver = CLinearSolver_Schur_HIP(schur_sparse=sparse)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta), (seed, nc, n_pts, mode, k, sparse)
    assert rel_inf(eta, x_ref) < TOL, (seed, nc, n_pts, mode, k, sparse)
    if seed % 3 == 0:
        cams_ref, pts_ref = O.schur_marginals(lam)
        cams, pts = solver.Schur_Marginals(lam)
        assert rel_inf(cams, cams_ref) < TOL and rel_inf(pts, pts_ref) < TOL


def _relinearized(lam, points, rng):
    """The same BA system after a relinearization that moved the landmarks ``points``: their C blocks and the U blocks
    of their observations change, so do all camera blocks and the whole right-hand side."""
    nc = lam.n_matrix_cut
    off = lam.block_value_offsets()
    vals = lam.values.copy()
    for p in points:
        k0, k1 = int(lam.bcol_ptr[nc + p]), int(lam.bcol_ptr[nc + p + 1])
        for k in range(k0, k1 - 1):                                  # U blocks of the landmark's observations
            vals[off[k]:off[k + 1]] *= 1.0 + 0.3 * rng.standard_normal()
        d = int(lam.cumsum[nc + p + 1] - lam.cumsum[nc + p])
        vals[off[k1 - 1]:off[k1]] += (0.5 + rng.random()) * np.eye(d).ravel()   # C_p stays positive definite
    for c in range(nc):                                              # camera blocks: a little more on the diagonal
        k = int(lam.bcol_ptr[c + 1] - 1)
        d = int(lam.cumsum[c + 1] - lam.cumsum[c])
        vals[off[k]:off[k + 1]] += (0.1 + rng.random()) * np.eye(d).ravel()
    return dataclasses.replace(lam, values=vals, rhs=rng.standard_normal(lam.rhs.shape[0]))


@pytest.mark.parametrize("sparse", [0, 1])
@pytest.mark.parametrize("mode", ["venice", "band"])
def test_incremental_update_of_the_reduced_system_matches_full_recomputation(mode, sparse):
    """Option schur_incremental: a solve that names the landmarks whose blocks changed updates the reduced camera system of
    the previous solve (the reference's dog-leg solver: NonlinearSolver_Lambda_DL.h:2301-) -- against the oracle's full
    solve of the changed system, over three relinearizations, the last one changing no landmark at all."""
    rng = np.random.default_rng(17)
    lam = synth.ba(120, 6000, k=4, mode=mode, seed=19)
    solver = CLinearSolver_Schur_HIP(schur_incremental=2, schur_sparse=sparse, profile=1)   # 2: update whenever a list is given
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok and rel_inf(eta, x_ref) < TOL
    n_pts = lam.n_bcols - lam.n_matrix_cut
    for n_changed in (57, 1, 0):
        points = np.sort(rng.choice(n_pts, size=n_changed, replace=False))
        lam = _relinearized(lam, points, rng)
        ok, x_ref, _, _ = O.solve_schur(lam)
        assert ok
        solver.profile(reset=True)
        solver.Set_Changed_Landmarks(points)
        eta = lam.rhs.copy()
        assert solver.Solve_PosDef_Blocky(lam, eta)
        phases = solver.profile()
        assert phases.get("schur_update", (0, 0))[0] == 1 and _rebuilds(phases) == 0   # it was an update
        assert rel_inf(eta, x_ref) < TOL, n_changed
    # without a list the next solve rebuilds; with the option off the call is refused
    eta = lam.rhs.copy()
    solver.profile(reset=True)
    assert solver.Solve_PosDef_Blocky(lam, eta) and _rebuilds(solver.profile()) == 1
    assert rel_inf(eta, x_ref) < TOL
    with pytest.raises(ValueError):
        CLinearSolver_Schur_HIP().Set_Changed_Landmarks([0])


@pytest.mark.parametrize("sparse", [0, 1])
@pytest.mark.parametrize("n_obs", [30, 40, 70])
def test_incremental_update_of_landmarks_seen_by_many_cameras(n_obs, sparse):
    """The update kernel takes a landmark's observations 32 at a time: one tile against itself (30), two and three tiles
    (40, 70) -- every landmark of these systems; against the oracle's full solve of the changed system."""
    rng = np.random.default_rng(23)
    lam = synth.ba(90, 400, k=n_obs, mode="uniform", seed=29)
    solver = CLinearSolver_Schur_HIP(schur_incremental=2, schur_sparse=sparse, profile=1)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    n_pts = lam.n_bcols - lam.n_matrix_cut
    for n_changed in (37, 2):
        points = np.sort(rng.choice(n_pts, size=n_changed, replace=False))
        lam = _relinearized(lam, points, rng)
        ok, x_ref, _, _ = O.solve_schur(lam)
        assert ok
        solver.profile(reset=True)
        solver.Set_Changed_Landmarks(points)
        eta = lam.rhs.copy()
        assert solver.Solve_PosDef_Blocky(lam, eta)
        phases = solver.profile()
        assert phases.get("schur_update", (0, 0))[0] == 1 and _rebuilds(phases) == 0   # it was an update
        assert rel_inf(eta, x_ref) < TOL, n_changed


def test_incremental_update_only_where_it_pays():
    """Option schur_incremental = 1: a list naming more than 1 / 32 of the landmarks (landmark-major assembly) is answered
    with a full rebuild -- that is the shorter way --, a shorter list with the update; same solution either way."""
    rng = np.random.default_rng(3)
    lam = synth.ba(60, 6000, k=4, mode="band", seed=7)
    solver = CLinearSolver_Schur_HIP(schur_incremental=1, profile=1)
    assert solver.Solve_PosDef(lam, lam.rhs.copy())
    n_pts = lam.n_bcols - lam.n_matrix_cut
    for n_changed, b_update in ((400, False), (10, True)):
        points = np.sort(rng.choice(n_pts, size=n_changed, replace=False))
        lam = _relinearized(lam, points, rng)
        ok, x_ref, _, _ = O.solve_schur(lam)
        solver.profile(reset=True)
        solver.Set_Changed_Landmarks(points)
        eta = lam.rhs.copy()
        assert ok and solver.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, x_ref) < TOL
        phases = solver.profile()
        assert (phases.get("schur_update", (0, 0))[0] == 1) == b_update and (_rebuilds(phases) == 0) == b_update


def _rebuilds(phases):
    """How many full assemblies of the reduced system a profile holds: landmark-major (schur_tiles), from the contribution
    lists (schur_gather), or both for one assembly when some landmarks fit neither runs nor tiles."""
    return max(phases.get("schur_tiles", (0, 0))[0], phases.get("schur_gather", (0, 0))[0])


def test_incremental_update_after_a_failed_solve_rebuilds():
    """A solve that was not positive definite leaves nothing valid to update from: the next one rebuilds, list or not."""
    rng = np.random.default_rng(5)
    lam = synth.ba(40, 1500, k=3, seed=23)
    solver = CLinearSolver_Schur_HIP(schur_incremental=1, profile=1)
    assert solver.Solve_PosDef(lam, lam.rhs.copy())
    off = lam.block_value_offsets()
    bad_vals = lam.values.copy()
    last = lam.n_blocks - 1
    bad_vals[off[last]:off[last + 1]] -= 50.0 * np.eye(3).ravel()
    bad = dataclasses.replace(lam, values=bad_vals)
    solver.Set_Changed_Landmarks([lam.n_bcols - lam.n_matrix_cut - 1])
    assert solver.Solve_PosDef_Blocky(bad, bad.rhs.copy()) is False
    lam2 = _relinearized(lam, [3, 4], rng)
    ok, x_ref, _, _ = O.solve_schur(lam2)
    solver.profile(reset=True)
    solver.Set_Changed_Landmarks([3, 4])
    eta = lam2.rhs.copy()
    assert ok and solver.Solve_PosDef_Blocky(lam2, eta) and rel_inf(eta, x_ref) < TOL
    assert _rebuilds(solver.profile()) == 1


@pytest.mark.parametrize("tiles", [-1, 0, 1, 2, 3])
def test_landmarks_without_observations(tiles):
    """A landmark nobody sees any more (its block column holds C_p only): dl = C^-1 l, no contribution to S -- on every
    assembly path."""
    lam = synth.ba(30, 400, k=3, seed=5)
    nc = lam.n_matrix_cut
    off = lam.block_value_offsets()
    drop = {3, 4, 5, 120, 399}                                    # landmarks whose observations are removed
    keep_blk = np.ones(lam.n_blocks, dtype=bool)
    for p in drop:
        c = nc + p
        keep_blk[lam.bcol_ptr[c]:lam.bcol_ptr[c + 1] - 1] = False     # all but the diagonal block
    new_ptr = np.zeros_like(lam.bcol_ptr)
    new_ptr[1:] = np.cumsum([keep_blk[lam.bcol_ptr[c]:lam.bcol_ptr[c + 1]].sum() for c in range(lam.n_bcols)])
    vals = np.concatenate([lam.values[off[b]:off[b + 1]] for b in np.nonzero(keep_blk)[0]])
    cut = synth.BlockSystem(lam.cumsum.copy(), new_ptr, lam.brow_idx[keep_blk].copy(), vals, lam.rhs.copy(), nc)
    ok, x_ref, _, _ = O.solve_schur(cut)
    assert ok
    eta = cut.rhs.copy()
    assert CLinearSolver_Schur_HIP(schur_tiles=tiles).Solve_PosDef(cut, eta)
    assert rel_inf(eta, x_ref) < TOL


@pytest.mark.parametrize("sparse", [0, 1])
def test_prefix_runs_take_the_long_tracks_off_the_contribution_lists(sparse, monkeypatch):
    """Landmarks seen by more cameras than a tile takes (11 - 30) whose camera lists continue one another -- tracks born at
    the same camera, lost at different ones -- are assembled as ONE run on the matrix cores, each reading as zero beyond its
    own end (schur_run_kernel, run_k), instead of through the contribution lists: same solution as the oracle and as the
    lists; the lists' gather kernel has nothing left to do; the W the incremental update works from is stored for them."""
    lam = synth.ba(150, 3000, mode="tracks", seed=9)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    solver = CLinearSolver_Schur_HIP(schur_sparse=sparse, profile=1)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    phases = solver.profile()
    assert "schur_tiles" in phases, phases
    monkeypatch.setenv("SLAMPP_HIP_DEV", "1")                        # development knobs are read only with this set (csrc/plan.h)
    monkeypatch.setenv("SLAMPP_HIP_DEV_NO_PREFIX_RUNS", "1")
    lists = CLinearSolver_Schur_HIP(schur_sparse=sparse, profile=1)
    eta2 = lam.rhs.copy()
    assert lists.Solve_PosDef(lam, eta2) and rel_inf(eta2, x_ref) < TOL and rel_inf(eta, eta2) < 1e-11
    monkeypatch.delenv("SLAMPP_HIP_DEV_NO_PREFIX_RUNS")
    monkeypatch.delenv("SLAMPP_HIP_DEV")
    t_with = phases.get("schur_gather", (0, 0.0))[1] / max(phases.get("schur_gather", (1, 0))[0], 1)
    t_without = lists.profile()["schur_gather"][1] / lists.profile()["schur_gather"][0]
    assert t_with < 0.5 * t_without, (t_with, t_without)            # the long tracks were the lists' work
    # the incremental update of the reduced system reads the W the run kernel stored for those landmarks
    inc = CLinearSolver_Schur_HIP(schur_sparse=sparse, schur_incremental=2)
    eta = lam.rhs.copy()
    assert inc.Solve_PosDef(lam, eta)
    k = np.diff(lam.bcol_ptr)[lam.n_matrix_cut:] - 1
    points = np.nonzero(k > 10)[0][::7]
    vals = lam.values.copy()
    off = lam.block_value_offsets()
    for p_ in points:
        k1 = int(lam.bcol_ptr[lam.n_matrix_cut + p_ + 1])
        vals[off[k1 - 1]:off[k1]] += 0.7 * np.eye(3).ravel()
    lam2 = dataclasses.replace(lam, values=vals)
    ok, x2, _, _ = O.solve_schur(lam2)
    inc.Set_Changed_Landmarks(points)
    eta = lam2.rhs.copy()
    assert ok and inc.Solve_PosDef_Blocky(lam2, eta) and rel_inf(eta, x2) < TOL


@pytest.mark.parametrize("knob", ["SLAMPP_HIP_DEV_NO_QUAD_RUNS", "SLAMPP_HIP_DEV_NO_QUAD_WIDE"])
@pytest.mark.parametrize("mode", ["band", "venice", "tracks"])
def test_run_kernel_one_landmark_per_step_and_quads_agree(mode, knob, monkeypatch):
    """Round 5: the run kernel takes four landmarks per matrix-core step (K = landmark, three rounds for the coordinates)
    where a job stages a multiple of four; the development knobs bring back the one-landmark-per-step form (all jobs / the
    diagonal jobs of three and four tiles a side).  Same S, same solution, both against the oracle."""
    lam = synth.ba(150, 6000, k=4, mode=mode, seed=21)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    eta = lam.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    monkeypatch.setenv("SLAMPP_HIP_DEV", "1")
    monkeypatch.setenv(knob, "1")
    eta2 = lam.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(lam, eta2) and rel_inf(eta2, x_ref) < TOL
    assert rel_inf(eta2, eta) < 1e-11


@pytest.mark.parametrize("mult", [1, 2, 4])
@pytest.mark.parametrize("mode", ["band", "venice", "tracks"])
def test_run_pieces_longer_than_a_wave(mode, mult, monkeypatch):
    """Round 5: a job of the run kernel takes its landmarks 64 at a time and keeps its sums across those sub-pieces; pieces of
    2 x and 4 x the base length (a development knob: measured no faster) leave fewer partial blocks and
    the same S."""
    lam = synth.ba(60, 40000, k=4, mode=mode, seed=23)     # runs of hundreds of landmarks: several sub-pieces per job
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    monkeypatch.setenv("SLAMPP_HIP_DEV", "1")
    monkeypatch.setenv("SLAMPP_HIP_DEV_RUN_PIECE_MULT", str(mult))
    for opts in ({}, {"schur_incremental": 2}):            # (the second keeps W: the stores of every sub-piece)
        eta = lam.rhs.copy()
        assert CLinearSolver_Schur_HIP(**opts).Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL, (mult, opts)


@pytest.mark.parametrize("n_cams", [150, 160, 231])
def test_dense_factorization_with_128_row_update_jobs(n_cams, monkeypatch):
    """Round 5: the K = 256 update of the trailing matrix as 128 x 128 targets (csrc/dense_chol.hip, syrk_wide_tile) --
    taken by itself only from 96 trailing tiles on (n > 6 600); the development knob puts every far update on them: an odd
    number of trailing tiles (150 cameras: 15 tiles, the first tile column stays with the 64 x 64 jobs), an even one (160:
    16 tiles), several outer panels (231: 22 tiles).  Same solution as the 64 x 64 jobs, both against the oracle."""
    lam = synth.ba(n_cams, 4000, k=4, mode="uniform", seed=37)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    eta = lam.rhs.copy()
    assert CLinearSolver_Schur_HIP(schur_sparse=0).Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    monkeypatch.setenv("SLAMPP_HIP_DEV", "1")
    monkeypatch.setenv("SLAMPP_HIP_DEV_WIDE_MIN_TILES", "2")
    solver = CLinearSolver_Schur_HIP(schur_sparse=0, profile=1)
    eta2 = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta2) and rel_inf(eta2, x_ref) < TOL
    assert "dense_chol" in solver.profile()
    assert rel_inf(eta2, eta) < 1e-11


@pytest.mark.parametrize("mode", ["band", "venice", "tracks"])
def test_runs_found_by_the_hashes_alone(mode, monkeypatch):
    """Round 5: from 262 144 landmarks on, two landmarks are taken to be seen by the same cameras when both 64-bit hashes of
    their camera lists and the lengths agree (the lists themselves are no longer read side by side: csrc/schur_tiles.hip, "same
    lists"); smaller systems still compare the lists.  The development knob takes the hashes alone here too: same runs, same
    solution, against the oracle."""
    lam = synth.ba(120, 20000, k=5, mode=mode, seed=41)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    a = CLinearSolver_Schur_HIP()
    a.SymbolicDecomposition_Blocky(lam)
    eta = lam.rhs.copy()
    assert a.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, x_ref) < TOL
    monkeypatch.setenv("SLAMPP_HIP_DEV", "1")
    monkeypatch.setenv("SLAMPP_HIP_DEV_RUN_HASH_ONLY", "1")
    b = CLinearSolver_Schur_HIP()
    b.SymbolicDecomposition_Blocky(lam)
    eta2 = lam.rhs.copy()
    assert b.Solve_PosDef_Blocky(lam, eta2) and rel_inf(eta2, x_ref) < TOL
    assert np.array_equal(eta, eta2)                       # the same jobs in the same order: bit for bit
    assert a.stats()["device_bytes"] == b.stats()["device_bytes"]


def test_phases_are_also_reported_under_the_reference_names():
    """slampp_hip_get_profile_reference_names: the same event totals under the names the reference prints with
    __SCHUR_PROFILING (/root/reference/include/slam/LinearSolver_Schur.h:1895-1912), in its order, each the sum of the
    phases of slampp_hip_get_profile that do that step's work."""
    lam = synth.ba(60, 6000, k=4, mode="band")
    solver = CLinearSolver_Schur_HIP(profile=1)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    solver.profile(reset=True)
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef_Blocky(lam, eta)
    ours, theirs = solver.profile(), solver.profile_reference_names()
    assert list(theirs)[:7] == ["reperm", "slice", "transpose", "inverse + multiply + add", "RHS prep", "cholsol", "dy solve"]
    assert theirs["cholsol"][1] > 0 and theirs["inverse + multiply + add"][1] > 0
    groups = {"inverse + multiply + add": ("schur_init", "schur_tiles", "schur_points", "schur_gather"), "RHS prep": ("schur_rhs",),
              "cholsol": ("reduced_sparse", "dense_chol", "dense_solve"), "dy solve": ("backsubst",)}
    for name, members in groups.items():
        want = sum(ours[m][1] for m in members if m in ours)
        assert abs(theirs[name][1] - want) <= 1e-9 + 1e-6 * want, name
