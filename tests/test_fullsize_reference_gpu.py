"""Full-size parity at the configurations BASELINE.json names, against the *compiled reference itself*
(oracle/_ref/ref_harness, built from /root/reference by oracle/Makefile.ref and shipped to the GPU box as a binary):

  C3  100k-pose SE(3) chain + loops     vs CLinearSolver_CholMod(CHOLMOD_SUPERNODAL)::Solve_PosDef
                                         (/root/reference/src/slam/LinearSolver_CholMod.cpp:264-358)
  C4  BA 1k cameras x 500k landmarks    vs CLinearSolver_Schur<CLinearSolver_CholMod, ..>::Solve_PosDef
                                         (/root/reference/include/slam/LinearSolver_Schur.h:1525,1623-1935),
                                         band, Venice-like (ragged) and uniform (dense S) visibility
  C5  BA 2k cameras x 2M landmarks      one handle and two landmark shards: residual of the full system, agreement of the
                                         two, and the reference on a 100k-landmark cut of the same system (all cameras)

Tolerance: ||x_gpu - x_ref||_inf / ||x_ref||_inf < 1e-10 (BASELINE.json north_star).  Skipped -- not passed -- where the
reference binary is absent."""
import functools
import os
import tempfile
import threading

import numpy as np
import pytest

from slam_plus_plus_amd import sharding, synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
from oracle import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
TOL = 1e-10
needs_reference = pytest.mark.skipif(not O.have_reference(), reason="oracle/_ref/ref_harness is not built")


def test_the_compiled_reference_travelled_with_the_repository():
    """The full-size tests below compare with the LIVE reference (oracle/_ref/ref_harness, built in the build container by
    __graft_entry__.build() from /root/reference and shipped to the GPU box as a binary: git-ignored, not gpurun-ignored).
    Where it is missing they skip -- and a skip is easy to overlook, so this test is loud about it: it fails unless
    SLAMPP_ALLOW_NO_REFERENCE=1 says that a checkout without the reference's binaries is what was meant."""
    if os.environ.get("SLAMPP_ALLOW_NO_REFERENCE") == "1":
        pytest.skip("SLAMPP_ALLOW_NO_REFERENCE=1")
    missing = [f for f in ("ref_harness", "dropin_driver") if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", f))]
    assert not missing, f"oracle/_ref/{missing} not present: run __graft_entry__.build() where /root/reference exists, or set SLAMPP_ALLOW_NO_REFERENCE=1"


def rel_inf(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max())


def reference_solution(lam, solver):
    """x of the compiled reference's solver class on the same system."""
    with tempfile.TemporaryDirectory() as td:
        p, xp = os.path.join(td, "p.bin"), os.path.join(td, "x.bin")
        lam.save(p)
        r = O.reference_solve(p, solver, xp, reps=1, timeout=1800)
        assert r["ok"], r
        return np.fromfile(xp, dtype=np.float64)


def ba_residual_rel_inf(lam, x):
    """||Lambda x - eta||_inf / ||eta||_inf of a BA system straight from its block-CSC arrays (no scalar matrix: at C5 that
    would be 3e8 nonzeros)."""
    nc = lam.n_matrix_cut
    dims = np.diff(lam.cumsum)
    dc, dp = int(dims[0]), int(dims[nc])
    col = np.repeat(np.arange(lam.n_bcols), np.diff(lam.bcol_ptr))
    off = lam.block_value_offsets()
    r = -lam.rhs.copy()
    xc = x[:nc * dc].reshape(nc, dc)
    xp = x[nc * dc:].reshape(-1, dp)
    A = lam.values[:nc * dc * dc].reshape(nc, dc, dc)            # [cam, col, row]: column-major blocks
    r[:nc * dc] += np.einsum("ncr,nc->nr", A, xc).ravel()
    is_diag = lam.brow_idx == col
    u = np.nonzero(~is_diag)[0]                                  # U blocks: row = camera, column = landmark
    d = np.nonzero(is_diag)[0][nc:]
    cam, pt = lam.brow_idx[u].astype(np.int64), col[u] - nc
    rc = np.zeros((nc, dc))
    rp = np.zeros((len(d), dp))
    for c in range(dp):
        for rr in range(dc):
            v = lam.values[off[u] + (c * dc + rr)]              # U(rr, c) of every observation
            rc[:, rr] += np.bincount(cam, weights=v * xp[pt, c], minlength=nc)
            rp[:, c] += np.bincount(pt, weights=v * xc[cam, rr], minlength=len(d))
        for rr in range(dp):
            rp[:, rr] += lam.values[off[d] + (c * dp + rr)] * xp[:, c]
    r[:nc * dc] += rc.ravel()
    r[nc * dc:] += rp.ravel()
    return float(np.abs(r).max() / np.abs(lam.rhs).max())


@functools.lru_cache(maxsize=2)
def c4(mode):
    return synth.ba(1000, 500_000, k=4, mode=mode)


@needs_reference
def test_c3_full_size_against_reference_cholmod():
    lam = synth.pose_chain()                                     # C3: 100 000 poses, 201 998 upper blocks
    assert lam.n_bcols == 100_000 and lam.n_blocks == 201_998
    x_ref = reference_solution(lam, "cholmod_super")
    solver = CLinearSolver_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert rel_inf(eta, x_ref) < TOL
    eta2 = lam.rhs.copy()                                        # warm call, analysis reused
    assert solver.Solve_PosDef_Blocky(lam, eta2)
    assert rel_inf(eta2, x_ref) < TOL


@needs_reference
@pytest.mark.parametrize("mode", ["band", "venice", "uniform"])
def test_c4_full_size_against_reference_schur(mode):
    """band: S sparse (the library's sparse reduced solve); venice: ragged lists, 2..30 observations per landmark;
    uniform: S dense (the MFMA factorization, as the reference's dense LLT)."""
    lam = c4(mode)
    x_ref = reference_solution(lam, "schur")
    solver = CLinearSolver_Schur_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    assert rel_inf(eta[:n_x], x_ref[:n_x]) < TOL               # dx (cameras)
    assert rel_inf(eta[n_x:], x_ref[n_x:]) < TOL               # dl (landmarks)
    assert ba_residual_rel_inf(lam, eta) < 1e-9
    if mode == "band":                                            # the same system with S forced dense
        dense = CLinearSolver_Schur_HIP(schur_sparse=0)
        eta2 = lam.rhs.copy()
        assert dense.Solve_PosDef(lam, eta2)
        assert rel_inf(eta2, x_ref) < TOL


def cut_landmarks(lam, n_keep):
    """The same BA system with only its first n_keep landmarks (block columns are stored landmark by landmark)."""
    nc = lam.n_matrix_cut
    n = nc + n_keep
    nb = int(lam.bcol_ptr[n])
    off = lam.block_value_offsets()
    return synth.BlockSystem(lam.cumsum[:n + 1].copy(), lam.bcol_ptr[:n + 1].copy(), lam.brow_idx[:nb].copy(),
                             lam.values[:off[nb]].copy(), lam.rhs[:int(lam.cumsum[n])].copy(), nc)


@functools.lru_cache(maxsize=1)
def c5():
    return synth.ba(2000, 2_000_000, k=4, mode="band")


def test_c5_single_handle():
    """C5 (2k cameras x 2M landmarks) on one GPU: residual of the full system, linearity."""
    lam = c5()
    solver = CLinearSolver_Schur_HIP()
    eta = lam.rhs.copy()
    assert solver.Solve_PosDef(lam, eta)
    assert ba_residual_rel_inf(lam, eta) < 1e-9
    eta2 = -2.0 * lam.rhs
    assert solver.Solve_PosDef_Blocky(lam, eta2)
    assert rel_inf(eta2, -2.0 * eta) < TOL


@needs_reference
def test_c5_cut_against_reference_schur():
    """All 2 000 cameras of C5 with its first 100 000 landmarks: small enough for the reference's serial dense LLT of the
    12 000 x 12 000 reduced system, same camera-side structure."""
    lam = cut_landmarks(c5(), 100_000)
    x_ref = reference_solution(lam, "schur")
    for opts in ({}, {"schur_sparse": 0}):
        eta = lam.rhs.copy()
        assert CLinearSolver_Schur_HIP(**opts).Solve_PosDef(lam, eta)
        assert rel_inf(eta, x_ref) < TOL, opts


def test_c5_two_landmark_shards_match_single_handle():
    """C5 as two landmark shards (two handles on this GPU, one thread each, the all-reduce callback a barrier + sum):
    the block-list agreement, the packed exchange and the redundant reduced solves at full size."""
    import torch
    torch.zeros(1, device="cuda")
    lam = c5()
    single = lam.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(lam, single)
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
            solver = CLinearSolver_Schur_HIP()
            solver.set_option("shard_rank", rank)
            solver.set_option("shard_world", world)
            solver.set_allreduce(make_fn(rank))
            eta = shard.rhs.copy()
            assert solver.Solve_PosDef(shard, eta)
            results[rank] = (eta, sl)
        except Exception:                                         # pragma: no cover
            import traceback
            errors.append(traceback.format_exc())
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(timeout=600) for t in threads]
    assert not errors, errors
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    x = np.zeros_like(single)
    for eta, sl in results:
        x[sl] = eta[n_x:]
    x[:n_x] = results[0][0][:n_x]
    assert rel_inf(results[1][0][:n_x], results[0][0][:n_x]) < 1e-13
    assert rel_inf(x, single) < TOL


# ---- full size, ill conditioned: the bar is what the compiled reference's own solvers show among themselves ------------------

def reference_solutions(lam, solvers):
    """{solver: x or None (the solver returned false)} of the compiled reference on one system."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "p.bin")
        lam.save(p)
        for s in solvers:
            xp = os.path.join(td, f"x_{s}.bin")
            r = O.reference_solve(p, s, xp, reps=1, timeout=1800)
            out[s] = np.fromfile(xp, dtype=np.float64) if r["ok"] else None
    return out


def spread_of(xs):
    xs = [x for x in xs.values() if x is not None]
    return max(rel_inf(a, b) for a in xs for b in xs if a is not b)


C3_HARD = {   # 100k poses; sigma, weak prior and information matrices spread over 2 x info_decades decades (synth.pose_chain)
    "spread_1e-8": dict(sigma=0.1, prior=1e-2, info_decades=2.5, seed=302),
    "spread_1e-7": dict(sigma=0.15, prior=1e-3, info_decades=2.0, seed=303),
}


@needs_reference
@pytest.mark.parametrize("case", sorted(C3_HARD))
def test_c3_ill_conditioned_within_the_reference_solvers_spread(case):
    """C3's structure with the conditioning real pose graphs have: CHOLMOD (supernodal), CSparse and the reference's native
    block solver differ by 1e-8 .. 1e-7 on these systems (measured in the build container; re-measured here, live); the
    HIP path must be within ten times their spread of CHOLMOD -- 1e-10 is not a bar any Cholesky solver meets here."""
    lam = synth.pose_chain(**C3_HARD[case])
    refs = reference_solutions(lam, ["cholmod_super", "csparse", "uberblock"])
    assert all(x is not None for x in refs.values())
    spread = spread_of(refs)
    eta = lam.rhs.copy()
    assert CLinearSolver_HIP().Solve_PosDef(lam, eta)
    err = rel_inf(eta, refs["cholmod_super"])
    print(f"C3 {case}: inter-oracle spread {spread:.2e}, HIP vs CHOLMOD {err:.2e}")
    assert 1e-10 < spread < 1e-5                                  # (the case is what it claims to be)
    assert err < 10 * spread


@needs_reference
def test_c3_numerically_singular_verdict_matches_the_llt_oracles():
    """sigma = 0.3 at 100k poses: Lambda is positive semi-definite by construction and singular to working precision; all
    three LL^T oracles return false, and so must the HIP path (not a solution of garbage)."""
    lam = synth.pose_chain(sigma=0.3, prior=1e-4, info_decades=1.5, seed=302)
    refs = reference_solutions(lam, ["cholmod_super", "csparse", "uberblock"])
    assert all(x is None for x in refs.values())
    assert CLinearSolver_HIP().Solve_PosDef(lam, lam.rhs.copy()) is False


@needs_reference
def test_c4_venice_ill_conditioned_within_the_reference_solvers_spread():
    """C4's Venice-like system with damping 1e-6, small baselines and rotation columns twenty times the translation ones:
    the reference's Schur solver against its own CHOLMOD on the whole of Lambda gives the spread (on a 100k-landmark cut:
    CHOLMOD on the full system is a minute of CPU); the HIP Schur path is held to ten times that on the cut and on the full
    system (there against the reference's Schur solver alone)."""
    lam = synth.ba(1000, 500_000, mode="venice", damping=1e-6, baseline=0.05, rot_scale=20, seed=778)
    cut = cut_landmarks(lam, 100_000)
    refs = reference_solutions(cut, ["schur", "cholmod_super"])
    assert all(x is not None for x in refs.values())
    spread = spread_of(refs)
    eta = cut.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(cut, eta)
    err_cut = rel_inf(eta, refs["schur"])
    x_ref = reference_solution(lam, "schur")
    eta = lam.rhs.copy()
    assert CLinearSolver_Schur_HIP().Solve_PosDef(lam, eta)
    err = rel_inf(eta, x_ref)
    print(f"C4 Venice-like, ill conditioned: spread (Schur vs CHOLMOD, 100k landmarks) {spread:.2e}, HIP vs Schur {err_cut:.2e} (cut) {err:.2e} (full)")
    bound = max(10 * spread, 1e-10)
    assert err_cut < bound and err < bound
