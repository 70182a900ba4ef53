"""One handle over several devices (slampp_hip_create_multi, csrc/group.hip): a BA system cut into landmark shards inside the
library, one member and one host thread per listed device, the reduced camera system summed by the library's own exchange.
On a 1-GPU box the members share device 0 (the exchange then runs through peer pointers); the tests that need two
distinct devices skip there.  Reference: the unsharded oracle solution and the single-device handle, rel-inf 1e-10."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10      # north-star tolerance: ||x_gpu - x_ref||_inf / ||x_ref||_inf


def rel_inf(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max())


def n_devices():
    import torch
    return torch.cuda.device_count()


def systems():
    from slam_plus_plus_amd import synth
    return {
        "band": synth.ba(60, 4000, k=4, mode="band", seed=11),          # sparse reduced system at 200 cameras and up; dense here
        "venice": synth.ba(40, 2500, mode="venice", seed=77),
        "uniform": synth.ba(24, 1500, k=4, mode="uniform", seed=5),
        "band_sparse_S": synth.ba(200, 6000, k=4, mode="band", seed=3),  # under 15 % of the camera pairs: sparse block path
    }


@pytest.fixture(scope="module")
def oracle_solutions(built):
    from oracle import oracle_lib as O
    out = {}
    for name, lam in systems().items():
        ok, x, _, _ = O.solve_schur(lam)
        assert ok
        out[name] = (lam, x)
    return out


@pytest.mark.parametrize("name", ["band", "venice", "uniform", "band_sparse_S"])
@pytest.mark.parametrize("members", [2, 3])
def test_shards_on_one_device_match_oracle_and_single_handle(oracle_solutions, name, members):
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    lam, x_ref = oracle_solutions[name]
    multi = CLinearSolver_Schur_HIP(devices=[0] * members)
    eta = lam.rhs.copy()
    assert multi.Solve_PosDef(lam, eta)
    info = multi.group_info()
    assert info["members"] == members and info["exchange"] == "peer"
    assert info["point_bounds"][0] == 0 and info["point_bounds"][-1] == lam.n_bcols - lam.n_matrix_cut
    assert rel_inf(eta, x_ref) < TOL
    single = CLinearSolver_Schur_HIP(device=0)
    eta1 = lam.rhs.copy()
    assert single.Solve_PosDef(lam, eta1)
    assert rel_inf(eta, eta1) < TOL
    # the analysis is reused: a second solve with other values through the pinned staging
    vals, rhs = multi.host_staging()
    lam2 = type(lam)(lam.cumsum, lam.bcol_ptr, lam.brow_idx, lam.values * 1.5, lam.rhs * 3.0, lam.n_matrix_cut)
    vals[:] = lam2.values
    eta2 = lam2.rhs.copy()
    assert multi.Solve_PosDef_Blocky(lam2, eta2)
    assert rel_inf(eta2, 2.0 * x_ref) < TOL
    st = multi.stats()
    assert st["n_points"] == lam.n_bcols - lam.n_matrix_cut and st["n_cams"] == lam.n_matrix_cut


def test_not_positive_definite_landmark_fails_every_member(oracle_solutions):
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    lam, _ = oracle_solutions["venice"]
    off = lam.block_value_offsets()
    nc = lam.n_matrix_cut
    vals = lam.values.copy()
    k = int(lam.bcol_ptr[nc + 2000 + 1]) - 1      # the diagonal block of a landmark of the last shard
    vals[off[k]:off[k] + 9] = -np.eye(3).ravel()
    bad = type(lam)(lam.cumsum, lam.bcol_ptr, lam.brow_idx, vals, lam.rhs, nc)
    multi = CLinearSolver_Schur_HIP(devices=[0, 0, 0])
    assert multi.Solve_PosDef(bad, bad.rhs.copy()) is False
    eta = lam.rhs.copy()          # and the handle is fine afterwards
    assert multi.Solve_PosDef_Blocky(lam, eta)
    assert rel_inf(eta, oracle_solutions["venice"][1]) < TOL


@pytest.mark.parametrize("name", ["venice", "band_sparse_S"])
def test_marginals_and_marginal_poses_through_shards(oracle_solutions, name):
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    lam, _ = oracle_solutions[name]
    single, multi = CLinearSolver_Schur_HIP(device=0), CLinearSolver_Schur_HIP(devices=[0, 0])
    cams1, pts1 = single.Schur_Marginals(lam)
    cams2, pts2 = multi.Schur_Marginals(lam)
    assert rel_inf(cams2, cams1) < TOL and rel_inf(pts2, pts1) < TOL
    e1, e2 = lam.rhs.copy(), lam.rhs.copy()
    assert single.Solve_PosDef_Blocky_MarginalPoses(lam, e1) and multi.Solve_PosDef_Blocky_MarginalPoses(lam, e2)
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    assert np.all(e2[:n_x] == 0) and rel_inf(e2[n_x:], e1[n_x:]) < TOL


@pytest.mark.parametrize("members", [2, 3])
def test_distributed_dense_factorization_of_the_reduced_system(built, members):
    """Option schur_distributed: the dense reduced camera system is reduce-scattered by outer panels of 256 columns and
    factored by all members together (panel b by member b mod P, finished panels sent to the other members' copies) instead
    of summed everywhere and factored by everyone.  Same solution as the single handle, 1e-10; the members' phases say
    which way they went."""
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    from oracle import oracle_lib as O
    lam = synth.ba(210, 5000, k=4, mode="uniform", seed=13)        # n = 1260: 20 tiles, 5 outer panels
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    single = CLinearSolver_Schur_HIP(device=0, schur_sparse=0)
    e1 = lam.rhs.copy()
    assert single.Solve_PosDef(lam, e1) and rel_inf(e1, x_ref) < TOL
    multi = CLinearSolver_Schur_HIP(devices=[0] * members, schur_sparse=0, schur_distributed=1, profile=1)
    for _ in range(3):                                             # the events and the sequence numbers are reused
        e2 = lam.rhs.copy()
        assert multi.Solve_PosDef_Blocky(lam, e2)
        assert rel_inf(e2, x_ref) < TOL and rel_inf(e2, e1) < TOL
    phases = multi.profile()
    assert "dense_chol_distributed" in phases and "dense_chol" not in phases and "allreduce" not in phases, phases
    # not positive definite: the owner of the failing panel says so, the handle returns false, and works afterwards
    off = lam.block_value_offsets()
    vals = lam.values.copy()
    vals[off[150]:off[150] + 36] = -np.eye(6).ravel()             # camera 150's diagonal block
    bad = type(lam)(lam.cumsum, lam.bcol_ptr, lam.brow_idx, vals, lam.rhs, lam.n_matrix_cut)
    assert multi.Solve_PosDef_Blocky(bad, bad.rhs.copy()) is False
    e3 = lam.rhs.copy()
    assert multi.Solve_PosDef_Blocky(lam, e3) and rel_inf(e3, x_ref) < TOL


@pytest.mark.parametrize("name", ["band", "band_sparse_S"])
@pytest.mark.parametrize("fail_member", [0, 1, 2])
def test_failure_agreement_nobody_enqueues_when_a_member_fails(oracle_solutions, name, fail_member):
    """A member that fails on its way to the exchange (allocation, upload, device error; injected here with the option
    group_fail_member) must not leave the others parked in a collective: the members meet at a host barrier with their status
    before anything is enqueued -- group_members_agree() in csrc/group.hip, the one function both the RCCL and the peer branch
    of the callback call first -- and when one is missing NOBODY enqueues.  The call returns the member's error, the exchange
    counter does not move, and the very next solve is right."""
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    lam, x_ref = oracle_solutions[name]
    multi = CLinearSolver_Schur_HIP(devices=[0, 0, 0])
    eta = lam.rhs.copy()
    assert multi.Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    n_before = multi.exchange_count()
    assert n_before >= 1
    multi.set_option("group_fail_member", fail_member + 1)
    with pytest.raises(RuntimeError, match="injected failure"):
        multi.Solve_PosDef_Blocky(lam, lam.rhs.copy())
    assert multi.exchange_count() == n_before            # nobody went into the collective
    multi.set_option("group_fail_member", 0)
    for _ in range(2):
        eta = lam.rhs.copy()
        assert multi.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, x_ref) < TOL
    assert multi.exchange_count() > n_before


def test_failure_agreement_with_the_distributed_factorization(built):
    """The same with schur_distributed: the members order their event waits by a call count, and one that failed before it got
    to the factorization is a call behind -- the counts go back to zero with every failed solve."""
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    from oracle import oracle_lib as O
    lam = synth.ba(210, 5000, k=4, mode="uniform", seed=13)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    multi = CLinearSolver_Schur_HIP(devices=[0, 0], schur_sparse=0, schur_distributed=1)
    eta = lam.rhs.copy()
    assert multi.Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    for fail in (2, 1):
        multi.set_option("group_fail_member", fail)
        with pytest.raises(RuntimeError, match="injected failure"):
            multi.Solve_PosDef_Blocky(lam, lam.rhs.copy())
        multi.set_option("group_fail_member", 0)
        for _ in range(2):
            eta = lam.rhs.copy()
            assert multi.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, x_ref) < TOL


def test_a_refused_option_is_not_replayed_into_the_group(oracle_solutions):
    """An option the front handle refuses must not be remembered for the members (they come up with the first Schur-mode
    analysis and are given the options set so far): that used to fail every analysis after it."""
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    lam, x_ref = oracle_solutions["band"]
    multi = CLinearSolver_Schur_HIP(devices=[0, 0])
    with pytest.raises(ValueError):
        multi.set_option("no_such_option", 1)
    with pytest.raises(ValueError):
        multi.set_option("schur_tiles", 99)
    multi.set_option("schur_tiles", 1)
    eta = lam.rhs.copy()
    assert multi.Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    assert multi.group_info()["members"] == 2


def test_pose_graph_on_a_multi_device_handle_runs_on_the_first_device(built):
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
    from oracle import oracle_lib as O
    lam = synth.pose_chain(n=400, d=6)
    ok, x_ref, _ = O.solve_sparse(lam)
    solver = CLinearSolver_HIP(devices=[0, 0])
    eta = lam.rhs.copy()
    assert ok and solver.Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    assert solver.group_info()["members"] == 0


def test_device_entry_points_are_refused_while_sharded(oracle_solutions):
    import torch
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    lam, _ = oracle_solutions["band"]
    multi = CLinearSolver_Schur_HIP(devices=[0, 0])
    multi.SymbolicDecomposition_Blocky(lam)
    v = torch.from_numpy(lam.values).cuda()
    r = torch.from_numpy(lam.rhs).cuda()
    with pytest.raises(ValueError):
        multi.factor_solve_device(v.data_ptr(), r.data_ptr())
    with pytest.raises(ValueError):
        multi.set_allreduce(lambda p, n, s: 0)


def test_more_devices_than_landmarks(built):
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    from oracle import oracle_lib as O
    lam = synth.ba(6, 3, k=3, mode="uniform", seed=1)
    ok, x_ref, _, _ = O.solve_schur(lam)
    multi = CLinearSolver_Schur_HIP(devices=[0, 0, 0, 0, 0])
    eta = lam.rhs.copy()
    assert ok and multi.Solve_PosDef(lam, eta) and rel_inf(eta, x_ref) < TOL
    assert multi.group_info()["members"] == 3


def test_exchange_selftest_peer_and_rccl_calls(built):
    """The all-reduce by itself: peer pointers with three members on device 0, and the RCCL binding (dlopen, ncclCommInitAll,
    ncclAllReduce on the member's stream, ncclCommDestroy) with the one device a test box has."""
    from slam_plus_plus_amd import hip_solver
    rc, name = hip_solver.group_exchange_selftest([0, 0, 0], exchange=2, count=100_003)
    assert rc == 0 and name == "peer", (rc, name)
    rc, name = hip_solver.group_exchange_selftest([0], exchange=1, count=100_003)
    assert rc == 0 and name.startswith("rccl"), (rc, name)
    rc, name = hip_solver.group_exchange_selftest([0, 0], exchange=1, count=1000)     # RCCL cannot take a device twice
    assert rc != 0 and "distinct" in name, (rc, name)


def test_two_distinct_devices_rccl_and_peer(oracle_solutions):
    if n_devices() < 2:
        pytest.skip("needs two GPUs")
    from slam_plus_plus_amd import hip_solver
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    devs = list(range(min(n_devices(), 8)))
    for exchange in (1, 2):
        rc, name = hip_solver.group_exchange_selftest(devs, exchange=exchange, count=1 << 20)
        assert rc == 0, (exchange, rc, name)
    for name in ("venice", "band_sparse_S", "uniform"):
        lam, x_ref = oracle_solutions[name]
        for exchange in (0, 1, 2):
            multi = CLinearSolver_Schur_HIP(devices=devs, group_exchange=exchange)
            eta = lam.rhs.copy()
            assert multi.Solve_PosDef(lam, eta)
            info = multi.group_info()
            assert info["members"] == len(devs)
            assert info["exchange"].startswith("rccl" if exchange != 2 else "peer"), info
            assert rel_inf(eta, x_ref) < TOL
            # failure agreement on real devices, RCCL included: a member that does not come leaves nobody parked
            n_before = multi.exchange_count()
            multi.set_option("group_fail_member", len(devs))
            with pytest.raises(RuntimeError, match="injected failure"):
                multi.Solve_PosDef_Blocky(lam, lam.rhs.copy())
            assert multi.exchange_count() == n_before
            multi.set_option("group_fail_member", 0)
            eta = lam.rhs.copy()
            assert multi.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, x_ref) < TOL
    # the distributed dense factorization over distinct devices (peer pointers under either exchange)
    from slam_plus_plus_amd import synth
    from oracle import oracle_lib as O
    lam = synth.ba(210, 5000, k=4, mode="uniform", seed=13)
    ok, x_ref, _, _ = O.solve_schur(lam)
    for exchange in (1, 2):
        multi = CLinearSolver_Schur_HIP(devices=devs, group_exchange=exchange, schur_sparse=0, schur_distributed=1)
        for _ in range(2):
            eta = lam.rhs.copy()
            assert ok and multi.Solve_PosDef_Blocky(lam, eta) and rel_inf(eta, x_ref) < TOL
