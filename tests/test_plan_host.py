"""Host logic of the product without a GPU: nested-dissection ordering, block symbolic factorization,
update lists and the stage schedule (libslampp_hip.so, slampp_hip_plan_*), replayed on the CPU by
oracle_exec_plan -- which performs the same left-looking arithmetic as the HIP kernels and checks
that every operand was produced in an earlier stage or earlier in the same task."""
import numpy as np
import pytest

from oracle import oracle_lib as O
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import host_plan
from golden_util import golden_names, load_golden, rel_inf

TOL = 1e-10


def check_plan_invariants(lam, plan):
    n = lam.n_bcols
    perm = plan["perm"]
    assert sorted(perm.tolist()) == list(range(n))
    lptr, lrow = plan["lptr"], plan["lrow"]
    for j in range(n):
        rows = lrow[lptr[j]:lptr[j + 1]]
        assert rows[0] == j and np.all(np.diff(rows) > 0)         # diagonal first, sorted, lower triangular
    dense = plan["dense_pos"] >= 0                               # dense-top columns are not scheduled block by block
    assert sorted(plan["task_cols"].tolist()) == np.nonzero(~dense)[0].tolist()   # every other column exactly once
    if plan["dense_dim"]:
        parent = np.array([lrow[lptr[j] + 1] if lptr[j + 1] - lptr[j] > 1 else -1 for j in range(n)])
        assert all(parent[j] < 0 or dense[parent[j]] for j in np.nonzero(dense)[0])   # closed upwards in the etree
        # positions ascend with the elimination order and do not overlap; independent chains may start at a tile
        # boundary (the gaps get an identity diagonal), so the dimension is at least the sum of the column dimensions
        pos, dim = plan["dense_pos"][dense], plan["dim"][dense]
        assert pos[0] >= 0 and np.all(pos[1:] >= pos[:-1] + dim[:-1])
        assert plan["dense_dim"] == int(pos[-1] + dim[-1]) >= int(dim.sum())
        assert np.all(pos[1:][pos[1:] > pos[:-1] + dim[:-1]] % 64 == 0)
    assert np.all(np.diff(plan["stage_ptr"]) > 0)
    dims = np.diff(lam.cumsum)[perm]
    assert np.array_equal(plan["dim"], dims)
    # every source block of Lambda lands in exactly one factor block
    assert np.count_nonzero(plan["asrc"] >= 0) == lam.n_blocks


@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith("indefinite")])
@pytest.mark.parametrize("leaf,sub,dense_nb", [(0, 0, -1), (1, 1, 0), (3, 7, 4), (1000, 1000, 0), (2, 4, 2)])
def test_plan_replay_matches_reference(name, leaf, sub, dense_nb):
    lam, ref = load_golden(name)
    plan, stats = host_plan(lam, leaf, sub, dense_nb)
    check_plan_invariants(lam, plan)
    status, x = O.exec_plan(lam, plan)
    assert status == 0
    assert rel_inf(x, ref["x_cholmod_super"]) < TOL
    assert stats["l_blocks"] == plan["l_blocks"] and stats["n_stages"] == plan["n_stages"]


def test_plan_replay_detects_indefinite():
    lam, _ = load_golden("indefinite_n40")
    plan, _ = host_plan(lam)
    status, _ = O.exec_plan(lam, plan)
    assert status == 1


@pytest.mark.parametrize("n,every", [(10000, 50), (20000, 50), (8000, 10)])
def test_pose_chains_are_cut_by_vertex_number(n, every):
    """Round 4: where few edges cross "the vertices below some number", that cut is the bisection (plan.cpp,
    Find_Index_Cut): no traversal at the top levels of the recursion, and halves as even as the numbering allows --
    level structures from a peripheral vertex left the elimination tree of a 10k-pose chain 26 levels deep, 7 stages.
    The plan must replay to the reference's solution like any other."""
    lam = synth.pose_chain(n=n, d=6, seed=31, loop_every=every)
    plan, st = host_plan(lam, dense_top_nb=0)
    check_plan_invariants(lam, plan)
    assert st["etree_height"] <= (1.6 if every == 50 else 3.2) * np.log2(n)
    ok, x_ref, _ = O.solve_sparse(lam)
    status, x = O.exec_plan(lam, plan)
    assert ok and status == 0 and rel_inf(x, x_ref) < TOL


@pytest.mark.parametrize("make", [lambda: synth.pose_chain(n=12000, d=6, seed=5, loop_every=40),
                                  lambda: synth.pose_chain(n=3000, d=3, seed=6, loop_every=7)])
def test_plan_does_not_depend_on_how_many_threads_searched_for_it(make, monkeypatch):
    """Round 6: the ordering recursion hands subgraphs to other threads, and the candidates of the plan search (dense-top
    threshold x balance) are built side by side; which threads exist is the machine's business, the plan is the structure's.
    SLAMPP_HIP_DEV_ND_PAR_MIN above the system's size keeps the recursion on one thread (plan.cpp, CNestedDissection);
    every array of the plan must come out the same."""
    lam = make()
    plan_a, st_a = host_plan(lam)
    monkeypatch.setenv("SLAMPP_HIP_DEV", "1")
    monkeypatch.setenv("SLAMPP_HIP_DEV_ND_PAR_MIN", "100000000")
    monkeypatch.setenv("SLAMPP_HIP_DEV_SETUP_THREADS", "1")
    plan_b, st_b = host_plan(lam)
    assert set(plan_a) == set(plan_b)
    for k in plan_a:
        if isinstance(plan_a[k], np.ndarray):
            assert np.array_equal(plan_a[k], plan_b[k]), k
        else:
            assert plan_a[k] == plan_b[k], k
    assert st_a["etree_height"] == st_b["etree_height"]
