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


def run(args):
    out = subprocess.run([HARNESS] + args, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise RuntimeError(out.stdout + out.stderr)
    return json.loads(line[-1])


def main():
    if not os.path.exists(HARNESS):
        raise SystemExit("build the reference first: make -f oracle/Makefile.ref -j8")
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


if __name__ == "__main__":
    main()
