import glob
import os

import numpy as np

from slam_plus_plus_amd.synth import BlockSystem

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_names():
    """Linear systems with the reference solvers' solutions."""
    return [n for n in _names() if not n.startswith(("assembly_", "dump_", "cond_"))]


def cond_names():
    """The conditioning sweep: systems at cond 1e6 .. 1e12, barely indefinite ones, the reference's own LM-damped BA Lambda."""
    return [n for n in _names() if n.startswith("cond_")]


LLT_ORACLES = ("cholmod_super", "csparse", "uberblock")   # simplicial CHOLMOD is LDL^T and "solves" indefinite systems


def load_cond(name):
    """(BlockSystem, dict): x_* / ok_* / err_true_* of the five reference solvers, spread, cond2, cond_proxy, x_true,
    positive_definite (the LL^T oracles' common verdict)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    lam = BlockSystem(z["cumsum"].astype(np.int64), z["bcol_ptr"].astype(np.int64), z["brow_idx"].astype(np.int32),
                      z["values"], z["rhs"], int(z["n_matrix_cut"]), name)
    ref = {k: z[k] for k in z.files if k.startswith(("x_", "ok_", "err_true_", "spread", "cond", "positive_definite"))}
    return lam, ref


def cond_bounds(ref):
    """(bound on rel-inf distance to x_cholmod_super, bound on the forward error against x_true): ten times what the
    reference's own solvers show among themselves -- and the north star's 1e-10 where that is tighter than they are."""
    worst_true = max(float(v) for k, v in ref.items() if k.startswith("err_true_"))
    return max(10.0 * float(ref["spread"]), 1e-10), max(10.0 * worst_true, 1e-10)


def dump_names():
    """Systems with the files the reference's Save_MatrixMarket / Save_BlockLayout wrote for them."""
    return [n for n in _names() if n.startswith("dump_")]


def load_dump(name):
    """(BlockSystem the dump was written from, path of the .mtx, path of the .bla)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    lam = BlockSystem(z["cumsum"].astype(np.int64), z["bcol_ptr"].astype(np.int64), z["brow_idx"].astype(np.int32),
                      z["values"], z["rhs"], int(z["n_matrix_cut"]), name)
    return lam, os.path.join(GOLDEN_DIR, name + ".mtx"), os.path.join(GOLDEN_DIR, name + ".bla")


def assembly_names():
    """Edge sets with the Lambda / eta the reference's nonlinear solver assembled from them."""
    return [n for n in _names() if n.startswith("assembly_")]


def load_assembly(name):
    """(BlockSystem holding the reference's Lambda and eta, EdgeSet, dx of the reference's CHOLMOD solve)."""
    from slam_plus_plus_amd.synth import EdgeSet
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    lam = BlockSystem(z["cumsum"].astype(np.int64), z["bcol_ptr"].astype(np.int64), z["brow_idx"].astype(np.int32),
                      z["values"], z["rhs"], 0, name)
    es = EdgeSet(int(lam.n_bcols), z["v0"], z["v1"], z["J0"], z["J1"], z["sigma_inv"], z["err"], z["weight"],
                 int(z["unary_vertex"]), z["unary_factor"], z["unary_error"])
    return lam, es, z["x_cholmod_super"]


def load_golden(name):
    """(BlockSystem, dict of reference outputs)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    lam = BlockSystem(z["cumsum"].astype(np.int64), z["bcol_ptr"].astype(np.int64), z["brow_idx"].astype(np.int32),
                      z["values"], z["rhs"], int(z["n_matrix_cut"]), name)
    ref = {k: z[k] for k in z.files if k.startswith(("x_", "ok_", "S", "rhs_reduced", "cam_cov", "lm_cov", "cov_diag"))}
    return lam, ref


def rel_inf(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max())
