"""The library's own shard splitter (slampp_hip_landmark_shard, csrc/group.hip: host code, runs without a GPU) against
slam_plus_plus_amd/sharding.py, the Python statement of the same rule: the landmark ranges, the shard's block structure,
and the value / right-hand side ranges a member of a multi-device handle fetches from the caller's arrays."""
import numpy as np
import pytest

from slam_plus_plus_amd import hip_solver, sharding, synth


@pytest.mark.parametrize("mode,seed", [("band", 3), ("venice", 77), ("uniform", 5)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_splitter_matches_sharding_py(built, mode, seed, world):
    lam = synth.ba(12, 300, mode=mode, seed=seed)
    off = lam.block_value_offsets()
    nc = lam.n_matrix_cut
    bounds = sharding.shard_bounds(lam, world)
    assert bounds[0] == 0 and bounds[-1] == lam.n_bcols - nc and np.all(np.diff(bounds) >= 1)
    covered = 0
    for rank in range(world):
        ref, sl = sharding.landmark_shard(lam, rank, world)
        got = hip_solver.landmark_shard_structure(lam, rank, world)
        assert (got["n_point_begin"], got["n_point_end"]) == (bounds[rank], bounds[rank + 1])
        np.testing.assert_array_equal(got["cumsum"], ref.cumsum)
        np.testing.assert_array_equal(got["bcol_ptr"], ref.bcol_ptr)
        np.testing.assert_array_equal(got["brow_idx"], ref.brow_idx)
        assert (got["n_scalar_begin"], got["n_scalar_end"]) == (sl.start, sl.stop)
        assert got["n_camera_scalars"] == lam.cumsum[nc] and got["n_camera_values"] == off[lam.bcol_ptr[nc]]
        # the shard's values are the camera blocks followed by one contiguous piece of the full array
        piece = lam.values[got["n_value_begin"]:got["n_value_end"]]
        np.testing.assert_array_equal(piece, ref.values[got["n_camera_values"]:])
        covered += got["n_value_end"] - got["n_value_begin"]
    assert covered == lam.values.shape[0] - off[lam.bcol_ptr[nc]]


def test_more_shards_than_landmarks_is_refused_by_neither(built):
    """Fewer landmarks than ranks: the ranges stay ordered and cover everything (some are empty; a multi-device handle
    then uses as many members as there are landmarks)."""
    lam = synth.ba(6, 3, mode="uniform", seed=1)
    b = sharding.shard_bounds(lam, 5)
    assert b[0] == 0 and b[-1] == 3 and np.all(np.diff(b) >= 0)
    for rank in range(5):
        got = hip_solver.landmark_shard_structure(lam, rank, 5)
        assert (got["n_point_begin"], got["n_point_end"]) == (b[rank], b[rank + 1])


def test_bad_arguments(built):
    lam = synth.ba(6, 30, mode="band", seed=1)
    with pytest.raises(ValueError):
        hip_solver.landmark_shard_structure(lam, 2, 2)
    lam.n_matrix_cut = 0
    with pytest.raises(ValueError):
        hip_solver.landmark_shard_structure(lam, 0, 2)
