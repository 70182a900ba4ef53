#!/usr/bin/env python3
"""Generates the golden vectors in tests/golden/*.npz by running the *compiled reference*
(oracle/_ref/ref_harness, built from /root/reference by oracle/Makefile.ref) on small seeded systems.

Run in the build container only (the GPU box has no /root/reference):
    make -f oracle/Makefile.ref -j8 && python tests/golden/make_golden.py

Each .npz holds the inputs (upper block-CSC structure, packed values, eta) and, as outputs of the
reference's own solver classes on those inputs:
    x_cholmod_super / x_cholmod_simp : CLinearSolver_CholMod(CHOLMOD_SUPERNODAL|SIMPLICIAL, AMD)::Solve_PosDef
    x_csparse                        : CLinearSolver_CSparse::Solve_PosDef_Blocky
    x_uberblock                      : CLinearSolver_UberBlock<...>::Solve_PosDef_Blocky
    x_schur, S, rhs_reduced (BA)     : CLinearSolver_Schur<...>::Solve_PosDef and the intermediates of
                                       its steps replayed through public CUberBlockMatrix calls
    x_schur_marginal_poses (BA)      : CLinearSolver_Schur<...>::Solve_PosDef_Blocky_MarginalPoses (landmarks only)
    cov_diag (pose graphs)           : CMarginals::Calculate_DenseMarginals_Recurrent_FBS(.., mpart_Diagonal), fed as
                                       NonlinearSolver_Lambda.h:696-760 feeds it: the diagonal blocks of Lambda^-1
    cam_cov, lm_cov (BA)             : CSchurComplement_Marginals::Schur_Marginals, fed as NonlinearSolver_Lambda_DL.h:1590-1640
                                       feeds it: the diagonal blocks of the covariance Lambda^-1
    ok_* (negative case)             : the boolean each solver returned
    assembly_*.npz                   : `ref_harness lambda_dump`: a pose graph built from the reference's own
                                       vertex / edge types; per edge the Jacobians, Sigma^-1, error and robust
                                       weight at the initial point (inputs), and the Lambda / eta that
                                       CNonlinearSolver_Lambda handed to its linear solver (outputs), plus
                                       the CHOLMOD solution of that system
    dump_*.mtx / .bla / .npz         : a system written by the reference's Save_MatrixMarket / Save_BlockLayout (the format of
                                       its -dsm option), next to the arrays it was written from
    cond_*.npz                       : the conditioning sweep ('Conditioning vs the 1e-10 bar'): pose chains
                                       and BA systems at cond_2(Lambda) of about 1e6 / 1e9 / 1e12, two systems that are
                                       indefinite by twice their smallest eigenvalue, and the damped Lambda the reference's own
                                       CNonlinearSolver_Lambda_LM hands to its linear solver on 200 CVertexCam cameras
                                       (`ref_harness lambda_dump ba_lm`, the solve at which its damping is lowest).  Per
                                       system: the five reference solvers' solutions and verdicts (ok_*), ``spread`` = the
                                       largest rel-inf distance between two of them, ``cond2`` (dense eigenvalues),
                                       ``cond_proxy`` = (max / min diagonal entry of the Cholesky factor)^2, and ``x_true``:
                                       a dense solve refined with residuals in 60-digit arithmetic (mpmath), the
                                       yardstick the forward errors ``err_true_*`` of the five are measured against
Fixtures are data only; no reference source text is stored.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from slam_plus_plus_amd import synth  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")

CASES = {
    "chain6_n60": lambda: synth.pose_chain(n=60, d=6, loop_every=10, loop_min=4, loop_max=9, seed=101),
    "chain3_n90": lambda: synth.pose_chain(n=90, d=3, loop_every=12, loop_min=5, loop_max=11, seed=102),
    "chain7_n40": lambda: synth.pose_chain(n=40, d=7, loop_every=9, loop_min=3, loop_max=8, seed=103),
    "sphere_8x8": lambda: synth.sphere(8, 8, seed=104),
    "manhattan_n150": lambda: synth.manhattan(150, seed=105),
    "ba_12x150_venice": lambda: synth.ba(12, 150, mode="venice", seed=106),
    "ba_10x120_band": lambda: synth.ba(10, 120, k=3, mode="band", seed=107),
    "indefinite_n40": lambda: synth.indefinite(40, 6, seed=5),
}


def _shift_diag(lam, eps):
    """Lambda - eps * I (on a copy)."""
    import dataclasses
    off = lam.block_value_offsets()
    v = lam.values.copy()
    for j in range(lam.n_bcols):
        b = int(lam.bcol_ptr[j + 1] - 1)
        d = int(lam.cumsum[j + 1] - lam.cumsum[j])
        v[off[b]:off[b + 1]].reshape(d, d)[...] -= eps * np.eye(d)
    return dataclasses.replace(lam, values=v)


def _indefinite_by(lam, factor):
    """Lambda - factor * lambda_min * I: indefinite by a margin of the size of its own smallest eigenvalue."""
    w = np.linalg.eigvalsh(lam.to_scipy().toarray())
    return _shift_diag(lam, factor * w[0])


_CHAIN = dict(n=120, d=6, loop_every=10, loop_min=4, loop_max=9)
COND_CASES = {
    "cond_chain6_1e6": lambda: synth.pose_chain(seed=201, sigma=0.1, prior=1.0, info_decades=1.8, **_CHAIN),
    "cond_chain6_1e9": lambda: synth.pose_chain(seed=202, sigma=0.3, prior=1e-4, info_decades=1.5, **_CHAIN),
    "cond_chain6_1e12": lambda: synth.pose_chain(seed=203, sigma=0.3, prior=1e-8, info_decades=4.0, **_CHAIN),
    "cond_chain3_1e7": lambda: synth.pose_chain(n=200, d=3, loop_every=12, loop_min=5, loop_max=11, seed=205, sigma=0.3,
                                                prior=1e-4, info_decades=1.7),
    "cond_ba_1e6": lambda: synth.ba(40, 400, k=3, mode="band", seed=211, damping=1e-6, baseline=0.05),
    "cond_ba_1e9": lambda: synth.ba(40, 400, mode="venice", seed=212, damping=1e-6, baseline=0.02, rot_scale=20),
    "cond_ba_1e12": lambda: synth.ba(40, 400, k=2, mode="band", seed=213, damping=1e-6, baseline=1e-3, rot_scale=100),
    "cond_chain6_1e9_indefinite": lambda: _indefinite_by(COND_CASES["cond_chain6_1e9"](), 2.0),
    "cond_ba_1e9_indefinite": lambda: _indefinite_by(COND_CASES["cond_ba_1e9"](), 2.0),
}
BA_LM_DUMP = ("cond_ba_lm_200cams", 200, 1000, 4, 11, 6)   # name, cameras, points, observations per point, seed, n_solve
LLT_SOLVERS = ("cholmod_super", "csparse", "uberblock")     # the oracles whose not-PD verdict counts


def refined_solution(lam):
    """A dense solve refined with residuals in 60-digit arithmetic (mpmath) until the correction is below 1e-25 of
    the solution: the exact solution of the system as stored, rounded to doubles; (x, cond2, cond_proxy).  x is None if
    the refinement does not contract (cond * 1e-16 >= 1)."""
    A = lam.to_scipy().toarray()
    w = np.linalg.eigvalsh(A)
    cond2 = float(abs(w[-1]) / abs(w[0])) if w[0] != 0 else np.inf
    try:
        Lc = np.linalg.cholesky(A)
        dg = np.diag(Lc)
        proxy = float((dg.max() / dg.min()) ** 2)
    except np.linalg.LinAlgError:
        proxy = np.nan
    import mpmath as mp
    import scipy.linalg as sl
    mp.mp.dps = 60
    lu = sl.lu_factor(A)
    coo = lam.to_scipy().tocoo()
    rows, cols = coo.row.tolist(), coo.col.tolist()
    vals = [mp.mpf(float(v)) for v in coo.data]
    b = [mp.mpf(float(v)) for v in lam.rhs]
    x = [mp.mpf(float(v)) for v in sl.lu_solve(lu, lam.rhs)]
    step = np.inf
    for _ in range(40):   # every pass gains about -log10(cond * 1e-16) digits
        r = list(b)
        for i, j, v in zip(rows, cols, vals):
            r[i] -= v * x[j]
        dx = sl.lu_solve(lu, np.array([float(v) for v in r]))
        x = [xi + mp.mpf(float(d)) for xi, d in zip(x, dx)]
        xf = np.array([float(v) for v in x])
        step = float(np.abs(dx).max() / np.abs(xf).max())
        if step < 1e-25:
            break
    return (xf if step < 1e-20 else None), cond2, proxy


def rel_inf(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def cond_record(lam):
    rec = {"cumsum": lam.cumsum, "bcol_ptr": lam.bcol_ptr, "brow_idx": lam.brow_idx, "values": lam.values,
           "rhs": lam.rhs, "n_matrix_cut": np.int64(lam.n_matrix_cut)}
    xs = {}
    with tempfile.TemporaryDirectory() as td:
        prob = os.path.join(td, "p.bin")
        lam.save(prob)
        for s in ["cholmod_super", "cholmod_simp", "csparse", "uberblock"] + (["schur"] if lam.n_matrix_cut else []):
            xf = os.path.join(td, f"x_{s}.bin")
            r = run(["solve", prob, s, xf, "1"], check=False)
            rec[f"ok_{s}"] = np.bool_(r["ok"])
            if r["ok"]:
                xs[s] = rec[f"x_{s}"] = np.fromfile(xf)
    x_true, cond2, proxy = refined_solution(lam)
    rec["cond2"], rec["cond_proxy"] = np.float64(cond2), np.float64(proxy)
    llt_ok = all(bool(rec[f"ok_{s}"]) for s in LLT_SOLVERS)
    rec["positive_definite"] = np.bool_(llt_ok)
    if llt_ok:   # solutions of an indefinite system (simplicial CHOLMOD's LDL^T "succeeds") are not compared
        names = sorted(xs)
        rec["spread"] = np.float64(max(rel_inf(xs[a], xs[b]) for a in names for b in names if a != b))
        if x_true is not None:
            rec["x_true"] = x_true
            for s in names:
                rec[f"err_true_{s}"] = np.float64(rel_inf(xs[s], x_true))
    return rec


def make_cond_fixtures():
    for name, make in COND_CASES.items():
        lam = make()
        rec = cond_record(lam)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{name}: n={lam.n_scalars} cond2={float(rec['cond2']):.2e} proxy={float(rec['cond_proxy']):.2e} "
              f"spread={float(rec.get('spread', np.nan)):.2e}", {k[3:]: bool(v) for k, v in rec.items() if k.startswith("ok_")},
              {k[9:]: f"{float(v):.1e}" for k, v in rec.items() if k.startswith("err_true_")})
    name, n_cams, n_pts, k, seed, n_solve = BA_LM_DUMP
    with tempfile.TemporaryDirectory() as td:
        prefix = os.path.join(td, "lm")
        run(["lambda_dump", "ba_lm", str(n_cams), str(seed), prefix, str(n_pts), str(k), str(n_solve)])
        lam = synth.BlockSystem.load(prefix + ".lambda.bin")
    rec = cond_record(lam)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: n={lam.n_scalars} blocks={lam.n_blocks} cond2={float(rec['cond2']):.2e} spread={float(rec['spread']):.2e} "
          f"-> {os.path.getsize(path) / 1024:.1f} KiB", {k[9:]: f"{float(v):.1e}" for k, v in rec.items() if k.startswith("err_true_")})


def run(args, check=True):
    out = subprocess.run([HARNESS] + args, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise RuntimeError(out.stdout + out.stderr)
    return json.loads(line[-1])


def main():
    if not os.path.exists(HARNESS):
        raise SystemExit("build the reference first: make -f oracle/Makefile.ref -j8")
    if len(sys.argv) > 1 and sys.argv[1] == "cond":
        return make_cond_fixtures()
    for name, make in CASES.items():
        lam = make()
        rec = {"cumsum": lam.cumsum, "bcol_ptr": lam.bcol_ptr, "brow_idx": lam.brow_idx, "values": lam.values,
               "rhs": lam.rhs, "n_matrix_cut": np.int64(lam.n_matrix_cut)}
        with tempfile.TemporaryDirectory() as td:
            prob = os.path.join(td, "p.bin")
            lam.save(prob)
            solvers = ["cholmod_super", "cholmod_simp", "csparse", "uberblock"] + \
                (["schur", "schur_marginal_poses"] if lam.n_matrix_cut else [])
            for s in solvers:
                xf = os.path.join(td, f"x_{s}.bin")
                r = run(["solve", prob, s, xf, "1"])
                rec[f"ok_{s}"] = np.bool_(r["ok"])
                if r["ok"]:
                    rec[f"x_{s}"] = np.fromfile(xf)
            if not lam.n_matrix_cut and rec.get("ok_cholmod_super", False):
                assert run(["sparse_marginals", prob, os.path.join(td, "pm")])["ok"]
                d = int(lam.cumsum[1])
                rec["cov_diag"] = np.fromfile(os.path.join(td, "pm.cov_diag.bin")).reshape(-1, d, d)   # symmetric blocks
            if lam.n_matrix_cut:
                r = run(["schur_dump", prob, os.path.join(td, "sd")])
                N = int(lam.cumsum[lam.n_matrix_cut])
                rec["S"] = np.fromfile(os.path.join(td, "sd.S.bin")).reshape(N, N).T.copy()   # [row, col], upper triangle
                rec["rhs_reduced"] = np.fromfile(os.path.join(td, "sd.rhs_reduced.bin"))
                rec["x_schur_steps"] = np.fromfile(os.path.join(td, "sd.x.bin"))
                assert run(["schur_marginals", prob, os.path.join(td, "sm")])["ok"]
                rec["cam_cov"] = np.fromfile(os.path.join(td, "sm.cam_cov.bin")).reshape(-1, 6, 6)   # symmetric blocks
                rec["lm_cov"] = np.fromfile(os.path.join(td, "sm.lm_cov.bin")).reshape(-1, 3, 3)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{name}: n={lam.n_scalars} blocks={lam.n_blocks} -> {os.path.getsize(path) / 1024:.1f} KiB",
              {k: bool(v) for k, v in rec.items() if k.startswith("ok_")})
    for name, kind, n_poses, seed in (("assembly_se2_n40", "se2", 40, 1), ("assembly_se3_n40", "se3", 40, 2)):
        with tempfile.TemporaryDirectory() as td:
            prefix = os.path.join(td, "ld")
            run(["lambda_dump", kind, str(n_poses), str(seed), prefix])
            lam = synth.BlockSystem.load(prefix + ".lambda.bin")
            es = synth.EdgeSet.load(prefix + ".edges.bin")
            xf = os.path.join(td, "x.bin")
            assert run(["solve", prefix + ".lambda.bin", "cholmod_super", xf, "1"])["ok"]
            rec = {"cumsum": lam.cumsum, "bcol_ptr": lam.bcol_ptr, "brow_idx": lam.brow_idx, "values": lam.values,
                   "rhs": lam.rhs, "v0": es.v0, "v1": es.v1, "J0": es.J0, "J1": es.J1, "sigma_inv": es.sigma_inv,
                   "err": es.err, "weight": es.weight, "unary_vertex": np.int64(es.unary_vertex),
                   "unary_factor": es.unary_factor, "unary_error": es.unary_error, "x_cholmod_super": np.fromfile(xf)}
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{name}: verts={lam.n_bcols} edges={es.n_edges} flipped={int((es.v0 > es.v1).sum())} "
              f"-> {os.path.getsize(path) / 1024:.1f} KiB")
    # the reference's own dump format (-dsm): written by CUberBlockMatrix::Save_MatrixMarket / Save_BlockLayout
    for name, make in (("dump_chain6_n12", lambda: synth.pose_chain(n=12, d=6, loop_every=5, loop_min=2, loop_max=4, seed=3)),
                       ("dump_ba_5x40", lambda: synth.ba(5, 40, k=3, seed=4))):
        lam = make()
        with tempfile.TemporaryDirectory() as td:
            prob = os.path.join(td, "p.bin")
            lam.save(prob)
            mtx, bla = os.path.join(HERE, name + ".mtx"), os.path.join(HERE, name + ".bla")
            assert run(["dump_mm", prob, mtx, bla])["ok"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), cumsum=lam.cumsum, bcol_ptr=lam.bcol_ptr,
                            brow_idx=lam.brow_idx, values=lam.values, rhs=lam.rhs, n_matrix_cut=np.int64(lam.n_matrix_cut))
        print(f"{name}: {os.path.getsize(mtx) / 1024:.1f} KiB .mtx")
    make_cond_fixtures()


if __name__ == "__main__":
    main()
