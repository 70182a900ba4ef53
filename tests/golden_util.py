import glob
import os

import numpy as np

from slam_plus_plus_amd.synth import BlockSystem

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    """(BlockSystem, dict of reference outputs)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    lam = BlockSystem(z["cumsum"].astype(np.int64), z["bcol_ptr"].astype(np.int64), z["brow_idx"].astype(np.int32),
                      z["values"], z["rhs"], int(z["n_matrix_cut"]), name)
    ref = {k: z[k] for k in z.files if k.startswith(("x_", "ok_", "S", "rhs_reduced"))}
    return lam, ref


def rel_inf(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max())
