"""Landmark sharding of a BA system for multi-GPU Schur solves (new
functionality -- the reference is single-process).

Landmarks are independent units: C is block diagonal (asserted by the reference,
/root/reference/src/slam/LinearSolver_Schur_GPU.cpp:2417, LinearSolver_Schur.h:1721), and the
reduced camera system is a sum over landmarks,  S = A - sum_p U_p C_p^-1 U_p^T.  Every rank keeps
all cameras and a contiguous range of landmarks balanced by observation count; the camera block A
and the camera part of eta are split additively (rank r gets A / world), so that the RCCL
all-reduce of the partial [S | r] buffers yields the full reduced system on every rank.  dx is then
computed redundantly and dl shard-locally: no other exchange.
"""
from __future__ import annotations

import numpy as np

from .synth import BlockSystem


def shard_bounds(lam: BlockSystem, world: int) -> np.ndarray:
    """Landmark range [b[r], b[r+1]) of every rank, balanced by number of observations."""
    nc = lam.n_matrix_cut
    n_pts = lam.n_bcols - nc
    obs = np.diff(lam.bcol_ptr[nc:]) - 1
    cum = np.concatenate([[0], np.cumsum(obs)])
    targets = cum[-1] * np.arange(1, world) / world
    inner = np.searchsorted(cum, targets, side="left")
    b = np.maximum.accumulate(np.minimum(np.concatenate([[0], inner, [n_pts]]), n_pts).astype(np.int64))
    if n_pts >= world:      # no rank without landmarks (csrc/group.hip: shard_bounds applies the same rule)
        for r in range(1, world):
            b[r] = max(b[r], b[r - 1] + 1)
        for r in range(world - 1, 0, -1):
            b[r] = min(b[r], b[r + 1] - 1)
    return b


def landmark_shard(lam: BlockSystem, rank: int, world: int) -> tuple[BlockSystem, slice]:
    """The sub-system rank ``rank`` solves, and the slice of the global solution its landmarks map to."""
    nc = lam.n_matrix_cut
    if nc <= 0:
        raise ValueError("not a BA system (n_matrix_cut is 0)")
    b = shard_bounds(lam, world)
    p0, p1 = int(b[rank]), int(b[rank + 1])
    off = lam.block_value_offsets()
    a_blocks = int(lam.bcol_ptr[nc])
    k0, k1 = int(lam.bcol_ptr[nc + p0]), int(lam.bcol_ptr[nc + p1])
    n_x = int(lam.cumsum[nc])
    cumsum = np.concatenate([lam.cumsum[:nc + 1], lam.cumsum[nc + p0 + 1:nc + p1 + 1] - lam.cumsum[nc + p0] + n_x])
    bcol_ptr = np.concatenate([lam.bcol_ptr[:nc + 1], lam.bcol_ptr[nc + p0 + 1:nc + p1 + 1] - k0 + a_blocks])
    brow = np.concatenate([lam.brow_idx[:a_blocks], lam.brow_idx[k0:k1]]).astype(np.int32)
    # landmark diagonal blocks carry global row indices: renumber them
    cols = np.repeat(np.arange(p1 - p0), np.diff(lam.bcol_ptr[nc + p0:nc + p1 + 1]))
    tail = brow[a_blocks:]
    is_diag = tail >= nc
    tail[is_diag] = (nc + cols[is_diag]).astype(np.int32)
    values = np.concatenate([lam.values[:off[a_blocks]] / world, lam.values[off[k0]:off[k1]]])
    l0, l1 = int(lam.cumsum[nc + p0]), int(lam.cumsum[nc + p1])
    rhs = np.concatenate([lam.rhs[:n_x] / world, lam.rhs[l0:l1]])
    shard = BlockSystem(cumsum.astype(np.int64), bcol_ptr.astype(np.int64), brow, values, rhs, nc,
                        f"{lam.name}_shard{rank}of{world}")
    return shard, slice(l0, l1)
