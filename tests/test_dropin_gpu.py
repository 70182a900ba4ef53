"""End-to-end drop-in check on the GPU box: the reference's own CNonlinearSolver_Lambda (compiled from
/root/reference into oracle/_ref/dropin_driver by oracle/Makefile.ref, in the build container) runs
unchanged with CLinearSolver_HIP (include/slam/LinearSolver_HIP.h) as its linear solver, and
CLinearSolver_Schur_HIP solves CUberBlockMatrix BA systems next to the reference's CLinearSolver_Schur
-- same process, same inputs, 1e-10 agreement.  Skipped where the prebuilt driver is absent."""
import json
import os
import subprocess

import pytest

from slam_plus_plus_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "dropin_driver")


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/dropin_driver was not prebuilt")
def test_reference_nonlinear_solver_with_hip_linear_solver(tmp_path):
    lam = synth.ba(24, 1500, mode="venice", seed=42)
    p = tmp_path / "ba.bin"
    lam.save(str(p))
    from oracle import oracle_lib as O
    out = subprocess.run([DRIVER, str(p)], capture_output=True, text=True, timeout=900, env=O.reference_env())
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert line, out.stdout + out.stderr
    r = json.loads(line[-1])
    print(line[-1])
    assert r["se2_lambda_solver"]["state_rel_inf"] < 1e-9
    assert r["se3_lambda_solver"]["state_rel_inf"] < 1e-9
    # the reference's incremental solver (CNonlinearSolver_FastL) driving Solve_PosDef + Factorize_PosDef_Blocky
    for k in ("se3_fastl_solver", "se3_fastl_incremental"):
        assert r[k]["state_rel_inf"] < 1e-9, (k, r[k])
        assert r[k]["hip_factorize_calls"] + r[k]["hip_solve_calls"] > 0, (k, r[k])
    assert r["schur_cams_first"]["rel_inf"] < 1e-10
    assert r["schur_interleaved"]["rel_inf"] < 1e-10
    assert r["schur_sparse_reduced"]["ok_hip"] == 1 and r["schur_sparse_reduced"]["rel_inf"] < 1e-10
    for k in ("factorize_6x6_R", "factorize_6x6_L", "factorize_3x3_R", "factorize_3x3_L"):
        assert r[k]["ok_ref"] == 1 and r[k]["ok_hip"] == 1 and r[k]["rel_max"] < 1e-11, (k, r[k])
    assert r["pose_graph_marginals"]["ok_ref"] == 1 and r["pose_graph_marginals"]["ok_hip"] == 1
    assert r["pose_graph_marginals"]["rel_inf"] < 1e-10
    for k in ("schur_marginals_cams_first", "schur_marginals_interleaved"):   # block diagonal of the covariance
        assert r[k]["ok_ref"] == 1 and r[k]["ok_hip"] == 1 and r[k]["cam_rel_inf"] < 1e-10 and r[k]["lm_rel_inf"] < 1e-10, (k, r[k])
    for k in ("marginal_poses_cams_first", "marginal_poses_interleaved"):
        assert r[k]["ok_ref"] == 1 and r[k]["ok_hip"] == 1 and r[k]["rel_inf"] < 1e-10
    for k in ("schur_incremental_cams_first", "schur_incremental_interleaved"):   # reduced system updated, not rebuilt
        assert r[k]["ok_ref"] == 1 and r[k]["ok_hip"] == 1 and r[k]["changed_landmarks"] > 0 and r[k]["rel_inf"] < 1e-10, (k, r[k])
    # FastL with a loop closure at every step: many Factorize_PosDef_Blocky calls of different shapes on one instance
    # (sensitive to rounding: the yardstick is how far two of the reference's own solvers end apart)
    k = r["se3_fastl_loops_every_step"]
    assert k["state_rel_inf"] < 10 * k["reference_cholmod_vs_csparse_rel_inf"] + 1e-9 and k["hip_factorize_calls"] > 20, k
    # matrices of one shape and different patterns back to back on one instance (NonlinearSolver_FastL.h:2131, 2388):
    # the cached analysis must not be reused
    k = r["same_shape_new_pattern"]
    assert k["ok"] == 1 and k["factor_rel_max"] < 1e-11 and k["solve_rel_inf"] < 1e-10, k
    # BA through the reference's CNonlinearSolver_Lambda_LM with the Schur complement on: its m_schur_solver is
    # CLinearSolver_Schur<CLinearSolver_HIP, ..> = the GPU solver (NonlinearSolver_Base.h:345-346), nothing patched
    k = r["ba_lm_schur"]
    assert k["iterations_ref"] == k["iterations_hip"] > 0 and k["hip_schur_solves"] >= k["iterations_hip"], k
    assert abs(k["chi2_ref"] - k["chi2_hip"]) <= 1e-10 * abs(k["chi2_ref"]) and k["state_rel_inf"] < 1e-5, k
    # the same run with a device list taken from the environment (SLAMPP_HIP_DEVICES, here "0,0": two members on the one GPU
    # of the test box): every Schur solve of the unchanged LM solver ran as landmark shards inside the library
    k = r["ba_lm_schur_devices"]
    assert k["sharded_solves"] >= k["iterations_hip"] > 0 and k["state_rel_inf"] < 1e-5, k
    assert abs(k["chi2_ref"] - k["chi2_hip"]) <= 1e-10 * abs(k["chi2_ref"]), k
    assert out.returncode == 0 and r["failures"] == 0, r


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/dropin_driver was not prebuilt")
@pytest.mark.parametrize("kind", ["pose", "ba"])
def test_timing_mode_agrees_with_the_reference(tmp_path, kind):
    """`dropin_driver time`: the reference's solver class and the HIP one on one CUberBlockMatrix (gather into pinned
    staging in chunks, uploads overlapped, solution back) -- here only that both give the same answer."""
    lam = synth.pose_chain(n=5000) if kind == "pose" else synth.ba(100, 20000, mode="venice", seed=5)
    p = tmp_path / "p.bin"
    lam.save(str(p))
    from oracle import oracle_lib as O
    out = subprocess.run([DRIVER, "time", str(p), "3"], capture_output=True, text=True, timeout=600, env=O.reference_env())
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert line, out.stdout + out.stderr
    r = json.loads(line[-1])
    assert out.returncode == 0 and r["ok"] and r["rel_inf"] < 1e-10, r
