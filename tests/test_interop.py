"""Interop with the reference's matrix dumps: the .mtx / .bla pair its
CUberBlockMatrix::Save_MatrixMarket / Save_BlockLayout write (the format of the -dsm option and of slam_schur_orderings).
The fixtures were written by the compiled reference (tests/golden/make_golden.py).  CPU only."""
import os
import subprocess

import numpy as np
import pytest

from golden_util import dump_names, load_dump, rel_inf
from oracle import oracle_lib
from slam_plus_plus_amd import synth


@pytest.mark.parametrize("name", dump_names())
def test_reads_the_references_dump(name):
    lam, mtx, bla = load_dump(name)
    got = synth.load_matrix_market(mtx, bla, rhs=lam.rhs, n_matrix_cut=lam.n_matrix_cut)
    assert np.array_equal(got.cumsum, lam.cumsum)
    assert np.array_equal(got.bcol_ptr, lam.bcol_ptr) and np.array_equal(got.brow_idx, lam.brow_idx)
    # the dump prints 15 significant digits; the lower triangle of a diagonal block is not in it and is mirrored
    A, B = lam.to_scipy().toarray(), got.to_scipy().toarray()
    assert np.abs(A - B).max() <= 1e-14 * np.abs(A).max()
    ok, x_ref, _ = oracle_lib.solve_sparse(lam)
    ok2, x, _ = oracle_lib.solve_sparse(got)
    assert ok and ok2 and rel_inf(x, x_ref) < 1e-11


@pytest.mark.parametrize("name", dump_names())
def test_writer_round_trip_and_reference_reader(name, tmp_path):
    lam, _, _ = load_dump(name)
    mtx, bla = str(tmp_path / "out.mtx"), str(tmp_path / "out.bla")
    synth.save_matrix_market(lam, mtx, bla)
    back = synth.load_matrix_market(mtx, bla)
    assert np.array_equal(back.brow_idx, lam.brow_idx)
    assert np.array_equal(np.triu(back.to_scipy().toarray()), np.triu(lam.to_scipy().toarray()))   # 17 digits: exact
    # the layout file is byte-identical to what the reference writes for the same matrix
    assert open(bla).read().split() == open(os.path.join(os.path.dirname(load_dump(name)[1]), name + ".bla")).read().split()
    if oracle_lib.have_reference():     # the reference's own Load_MatrixMarket accepts the pair (build container only)
        prob = str(tmp_path / "p.bin")
        lam.save(prob)
        out = subprocess.run([oracle_lib.REF_HARNESS, "load_mm", mtx, bla, prob], capture_output=True, text=True, timeout=120)
        assert '"ok": true' in out.stdout and '"max_abs_diff_upper": 0' in out.stdout, out.stdout + out.stderr


def test_rejects_mismatched_layout(tmp_path):
    lam, mtx, bla = load_dump("dump_chain6_n12")
    other = load_dump("dump_ba_5x40")[2]
    with pytest.raises(ValueError):
        synth.load_matrix_market(mtx, other)
