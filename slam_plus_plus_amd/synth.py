"""Seeded synthetic block-sparse normal-equation systems ``Lambda * dx = eta``.

None of the datasets BASELINE.json names (Manhattan3500, Sphere2500, Venice) ships
with the reference and there is no network, so every configuration is exercised
with a look-alike of the same size and block structure.  The
generators only produce *data*; they are used by the tests, by bench.py and by
tests/golden/make_golden.py, never by the solver itself.

Storage mirrors what the reference's solvers read out of ``CUberBlockMatrix``
(/root/reference/include/slam/BlockMatrixBase.h:380-503): block-CSC, only the
upper triangle (block row <= block column) populated, block rows sorted inside a
block column, every block dense column-major.
"""
from __future__ import annotations

import dataclasses
import os
import struct
from typing import Optional

import numpy as np

MAGIC = b"SPPLAM01"


@dataclasses.dataclass
class BlockSystem:
    """Upper-triangular block-CSC matrix + right-hand side."""

    cumsum: np.ndarray      # int64 [n_bcols+1] scalar offset of every block column / row
    bcol_ptr: np.ndarray    # int64 [n_bcols+1]
    brow_idx: np.ndarray    # int32 [n_blocks]
    values: np.ndarray      # float64, blocks packed in block-CSC order, each column-major
    rhs: np.ndarray         # float64 [n_scalars]
    n_matrix_cut: int = 0   # BA: number of leading camera block columns (0 = not a BA system)
    name: str = ""

    @property
    def n_bcols(self) -> int:
        return len(self.cumsum) - 1

    @property
    def n_blocks(self) -> int:
        return len(self.brow_idx)

    @property
    def n_scalars(self) -> int:
        return int(self.cumsum[-1])

    def block_value_offsets(self) -> np.ndarray:
        """int64 [n_blocks+1] offset of every block in ``values``."""
        dims = np.diff(self.cumsum)
        cols = np.repeat(np.arange(self.n_bcols), np.diff(self.bcol_ptr))
        sz = dims[self.brow_idx] * dims[cols]
        off = np.zeros(self.n_blocks + 1, dtype=np.int64)
        np.cumsum(sz, out=off[1:])
        return off

    def to_scipy(self):
        """Full symmetric scalar CSC matrix (tests only)."""
        import scipy.sparse as sp

        dims = np.diff(self.cumsum)
        cols = np.repeat(np.arange(self.n_bcols), np.diff(self.bcol_ptr))
        off = self.block_value_offsets()
        ri, ci, vv = [], [], []
        for (h, w) in sorted(set(zip(dims[self.brow_idx].tolist(), dims[cols].tolist()))):
            sel = np.nonzero((dims[self.brow_idx] == h) & (dims[cols] == w))[0]
            if not len(sel):
                continue
            idx = off[sel][:, None] + np.arange(h * w)[None, :]
            v = self.values[idx]                                  # [nb, w*h] col-major
            rr = np.tile(np.arange(h), w)[None, :] + self.cumsum[self.brow_idx[sel]][:, None]
            cc = np.repeat(np.arange(w), h)[None, :] + self.cumsum[cols[sel]][:, None]
            ri.append(rr.ravel()); ci.append(cc.ravel()); vv.append(v.ravel())
        ri = np.concatenate(ri); ci = np.concatenate(ci); vv = np.concatenate(vv)
        n = self.n_scalars
        up = sp.triu(sp.coo_matrix((vv, (ri, ci)), shape=(n, n)).tocsc(), k=0)
        return (up + sp.triu(up, k=1).T).tocsc()  # diagonal blocks hold both triangles: keep one

    def save(self, path: str) -> None:
        with open(path, "wb") as f:
            f.write(MAGIC)
            f.write(struct.pack("<8q", self.n_bcols, self.n_blocks, self.n_scalars,
                                len(self.values), self.n_matrix_cut, 0, 0, 0))
            f.write(np.ascontiguousarray(self.cumsum, dtype="<i8").tobytes())
            f.write(np.ascontiguousarray(self.bcol_ptr, dtype="<i8").tobytes())
            f.write(np.ascontiguousarray(self.brow_idx, dtype="<i8").tobytes())
            f.write(np.ascontiguousarray(self.values, dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(self.rhs, dtype="<f8").tobytes())

    @staticmethod
    def load(path: str) -> "BlockSystem":
        with open(path, "rb") as f:
            if f.read(8) != MAGIC:
                raise ValueError(f"{path}: not a SPPLAM01 file")
            n_bcols, n_blocks, n_scalars, n_values, cut, _, _, _ = struct.unpack("<8q", f.read(64))
            cumsum = np.frombuffer(f.read(8 * (n_bcols + 1)), dtype="<i8").copy()
            bcol_ptr = np.frombuffer(f.read(8 * (n_bcols + 1)), dtype="<i8").copy()
            brow = np.frombuffer(f.read(8 * n_blocks), dtype="<i8").astype(np.int32)
            values = np.frombuffer(f.read(8 * n_values), dtype="<f8").copy()
            rhs = np.frombuffer(f.read(8 * n_scalars), dtype="<f8").copy()
        return BlockSystem(cumsum, bcol_ptr, brow, values, rhs, int(cut))


# --------------------------------------------------------------------------------------
# pose graphs (uniform d x d blocks)
# --------------------------------------------------------------------------------------

def _segment_sum(idx: np.ndarray, blocks: np.ndarray, n: int) -> np.ndarray:
    """out[idx[e]] += blocks[e], summed in the order of e (what np.add.at does, element by element with bincount)."""
    out = np.zeros((n,) + blocks.shape[1:])
    if blocks.shape[0] == 0:
        return out
    flat_in = blocks.reshape(blocks.shape[0], -1)
    flat_out = out.reshape(n, -1)
    for c in range(flat_in.shape[1]):
        flat_out[:, c] = np.bincount(idx, weights=flat_in[:, c], minlength=n)
    return out


def _assemble_pose_graph(n: int, d: int, ei: np.ndarray, ej: np.ndarray, rng, sigma: float,
                         prior: float, name: str, info_decades: float = 0.0) -> BlockSystem:
    """Lambda = sum over edges [Ja Jb]^T W [Ja Jb] + prior*I on pose 0, eta ~ N(0,1).

    Ja = I + sigma*G, Jb = -I + sigma*G are near-orthogonal relative-pose Jacobians,
    which keeps Lambda well conditioned (SURVEY.md section 7 'Conditioning vs the 1e-10 bar').
    The conditioning sweep (tests/golden/make_golden.py, ``cond_*``) leaves that regime on purpose: a weak
    ``prior`` (the gauge is then held by almost nothing), ``sigma`` up to 0.3 and, with ``info_decades`` = s > 0,
    an information matrix W = diag(10^u), u ~ U(-s, s) per edge and residual row (translations against rotations,
    odometry against loop closures: real pose graphs mix precisions over many decades).  W = I otherwise."""
    assert np.all(ei < ej)
    key = ei.astype(np.int64) * n + ej
    _, first = np.unique(key, return_index=True)      # drop duplicate edges, keep a stable order
    first.sort()
    ei, ej = ei[first], ej[first]
    E = len(ei)
    eye = np.eye(d)
    Ja = eye[None] + sigma * rng.standard_normal((E, d, d))
    Jb = -eye[None] + sigma * rng.standard_normal((E, d, d))
    if info_decades > 0:
        w = 10.0 ** rng.uniform(-info_decades, info_decades, size=(E, d))
        WJa, WJb = w[:, :, None] * Ja, w[:, :, None] * Jb
    else:
        WJa, WJb = Ja, Jb
    Hii = np.einsum("eki,ekj->eij", Ja, WJa)
    Hij = np.einsum("eki,ekj->eij", Ja, WJb)
    Hjj = np.einsum("eki,ekj->eij", Jb, WJb)
    diag = _segment_sum(np.concatenate([ei, ej]), np.concatenate([Hii, Hjj]), n)
    diag[0] += prior * eye
    # block-CSC: column j holds its off-diagonal blocks (rows i < j, sorted) then the diagonal block
    order = np.lexsort((ei, ej))
    ei_s, ej_s, Hij_s = ei[order], ej[order], Hij[order]
    cnt = np.bincount(ej_s, minlength=n) + 1
    bcol_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(cnt, out=bcol_ptr[1:])
    nb = int(bcol_ptr[-1])
    brow = np.empty(nb, dtype=np.int32)
    vals = np.empty((nb, d * d))
    diag_pos = bcol_ptr[1:] - 1
    brow[diag_pos] = np.arange(n, dtype=np.int32)
    vals[diag_pos] = diag.transpose(0, 2, 1).reshape(n, d * d)
    mask = np.ones(nb, dtype=bool)
    mask[diag_pos] = False
    off_pos = np.nonzero(mask)[0]
    brow[off_pos] = ei_s
    vals[off_pos] = Hij_s.transpose(0, 2, 1).reshape(E, d * d)
    cumsum = np.arange(n + 1, dtype=np.int64) * d
    rhs = rng.standard_normal(n * d)
    return BlockSystem(cumsum, bcol_ptr, brow, vals.ravel(), rhs, 0, name)


def pose_chain(n: int = 100_000, d: int = 6, loop_every: int = 50, loop_min: int = 26,
               loop_max: int = 50, sigma: float = 0.02, prior: float = 100.0,
               seed: int = 12345, info_decades: float = 0.0) -> BlockSystem:
    """C3 look-alike: odometry chain (i, i+1) + one loop closure per ``loop_every`` poses to a
    pose ``loop_min..loop_max`` back.  n=100000, d=6 gives 201 998 upper blocks."""
    rng = np.random.default_rng(seed)
    ci = np.arange(n - 1)
    ends = np.arange(loop_every, n, loop_every)
    back = rng.integers(loop_min, loop_max + 1, size=len(ends))
    li = ends - back
    ok = li >= 0
    ei = np.concatenate([ci, li[ok]])
    ej = np.concatenate([ci + 1, ends[ok]])
    return _assemble_pose_graph(n, d, ei, ej, rng, sigma, prior, f"pose_chain_n{n}_d{d}", info_decades)


def sphere(n_rings: int = 50, per_ring: int = 50, d: int = 6, sigma: float = 0.02,
           prior: float = 100.0, seed: int = 2500) -> BlockSystem:
    """C2 look-alike (Sphere2500): poses on rings; each pose is linked to its successor along the
    spiral (odometry) and to the pose one ring below (loop closures).  2 500 poses, 4 949 edges."""
    rng = np.random.default_rng(seed)
    n = n_rings * per_ring
    ci = np.arange(n - 1)
    vi = np.arange(n - per_ring)
    ei = np.concatenate([ci, vi])
    ej = np.concatenate([ci + 1, vi + per_ring])
    return _assemble_pose_graph(n, d, ei, ej, rng, sigma, prior, f"sphere_{n_rings}x{per_ring}_d{d}")


def manhattan(n: int = 3500, d: int = 3, world: int = 30, sigma: float = 0.02,
              prior: float = 100.0, seed: int = 3500, max_loops_per_pose: int = 2) -> BlockSystem:
    """C1 look-alike (Manhattan3500): random walk on a grid, loop closure whenever the walk
    revisits a cell (SE(2), 3x3 blocks; about 5.5k edges at n=3500)."""
    rng = np.random.default_rng(seed)
    steps = np.array([[1, 0], [-1, 0], [0, 1], [0, -1]])
    pos = np.zeros((n, 2), dtype=np.int64)
    heading = 0
    for i in range(1, n):
        if rng.random() < 0.3:
            heading = int(rng.integers(0, 4))
        p = pos[i - 1] + steps[heading]
        if np.any(np.abs(p) > world):
            heading ^= 1
            p = pos[i - 1] + steps[heading]
        pos[i] = p
    seen: dict[tuple[int, int], list[int]] = {}
    ei, ej = list(range(n - 1)), list(range(1, n))
    for i in range(n):
        key = (int(pos[i, 0]), int(pos[i, 1]))
        prev = seen.setdefault(key, [])
        cands = [p for p in prev if p < i - 1]
        for p in cands[-max_loops_per_pose:]:
            ei.append(p); ej.append(i)
        prev.append(i)
    return _assemble_pose_graph(n, d, np.array(ei), np.array(ej), rng, sigma, prior,
                                f"manhattan_n{n}_d{d}")


# --------------------------------------------------------------------------------------
# bundle adjustment (6x6 cameras first, 3x3 points last)
# --------------------------------------------------------------------------------------

def ba(n_cams: int = 1000, n_pts: int = 500_000, k: int = 4, mode: str = "band",
       damping: float = 0.1, seed: int = 777, cam_dim: int = 6, pt_dim: int = 3,
       cam_damping: float | None = None, baseline: float | None = None,
       rot_scale: float | None = None) -> BlockSystem:
    """C4/C5 look-alike.  Every point is seen by ``k`` cameras (``mode='venice'``: k drawn from a
    clipped geometric distribution, mean about 5.3; ``'tracks'``: longer ones, mean about 11, consecutive cameras).  Cameras of a point: ``band`` = c0 + 7j mod nc
    (sparse S), ``uniform`` = k distinct random cameras (dense S).  Per observation
    Jc in R^{2x6}, Jp in R^{2x3} ~ N(0,1):  A_cc += Jc^T Jc, C_pp += Jp^T Jp, U_cp = Jc^T Jp;
    ``damping``*I on every diagonal block.  ``cam_damping`` overrides the
    damping of the camera blocks (landmark shards of one system each carry 1/world of it).
    The conditioning sweep (``cond_ba_*`` fixtures) adds: ``baseline`` = b: every camera sees a point through nearly the
    same Jp (one per point + b * noise: a small baseline, C_pp within b^2 of rank 2 and held by the damping);
    ``rot_scale`` = r: the last three columns of Jc times r (rotations in radians against translations: focal-length-sized
    entries), and the Jc of a point's cameras drawn around one common Jacobian (near-collinear cameras)."""
    rng = np.random.default_rng(seed)
    if mode in ("venice", "tracks"):
        kk = np.clip(rng.geometric(0.19 if mode == "venice" else 0.08, size=n_pts) + 1, 2, min(30, n_cams))
    else:
        kk = np.full(n_pts, min(k, n_cams), dtype=np.int64)
    n_obs = int(kk.sum())
    pt_of = np.repeat(np.arange(n_pts), kk)
    first = np.zeros(n_pts + 1, dtype=np.int64)
    np.cumsum(kk, out=first[1:])
    j_in_pt = np.arange(n_obs) - first[pt_of]
    if mode == "uniform":
        # k distinct cameras per point: random start + distinct random strides would bias; use
        # rejection-free construction: sorted sample via random offsets in disjoint strata
        strata = (n_cams // kk.max()) if kk.max() else 1
        strata = max(int(strata), 1)
        cam_of = (j_in_pt * strata + rng.integers(0, strata, size=n_obs)
                  + np.repeat(rng.integers(0, n_cams, size=n_pts), kk)) % n_cams
    elif mode == "tracks":
        # feature tracks: a point is seen by kk consecutive cameras from the one it was first seen in (every fifth camera
        # starts tracks) -- the lists of the tracks born at one camera are prefixes of the longest of them
        c0 = np.minimum(5 * rng.integers(0, max(n_cams // 5, 1), size=n_pts), n_cams - kk)
        cam_of = c0[pt_of] + j_in_pt
    else:
        c0 = rng.integers(0, n_cams, size=n_pts)
        cam_of = (c0[pt_of] + 7 * j_in_pt) % n_cams
    # make sure cameras are distinct within a point (band with 7*j mod nc may wrap for tiny nc)
    key = pt_of.astype(np.int64) * n_cams + cam_of
    _, uniq = np.unique(key, return_index=True)
    uniq.sort()
    pt_of, cam_of = pt_of[uniq], cam_of[uniq]
    n_obs = len(pt_of)
    cd, pd_ = cam_dim, pt_dim
    Jc = rng.standard_normal((n_obs, 2, cd))
    Jp = rng.standard_normal((n_obs, 2, pd_))
    if baseline is not None:
        Jp = rng.standard_normal((n_pts, 2, pd_))[pt_of] + baseline * Jp
    if rot_scale is not None:
        Jc = rng.standard_normal((n_pts, 2, cd))[pt_of] + 0.25 * Jc
        Jc[:, :, cd - 3:] *= rot_scale
    Acc = _segment_sum(cam_of, np.einsum("oki,okj->oij", Jc, Jc), n_cams)
    Cpp = _segment_sum(pt_of, np.einsum("oki,okj->oij", Jp, Jp), n_pts)
    Ucp = np.einsum("oki,okj->oij", Jc, Jp)                 # [n_obs, 6, 3]
    Acc += (damping if cam_damping is None else cam_damping) * np.eye(cd)[None]
    Cpp += damping * np.eye(pd_)[None]
    # block-CSC layout
    order = np.lexsort((cam_of, pt_of))
    pt_s, cam_s, U_s = pt_of[order], cam_of[order], Ucp[order]
    cnt_pt = np.bincount(pt_s, minlength=n_pts) + 1
    n_bc = n_cams + n_pts
    bcol_ptr = np.zeros(n_bc + 1, dtype=np.int64)
    bcol_ptr[1:n_cams + 1] = np.arange(1, n_cams + 1)
    np.cumsum(cnt_pt, out=bcol_ptr[n_cams + 1:])
    bcol_ptr[n_cams + 1:] += n_cams
    nb = int(bcol_ptr[-1])
    brow = np.empty(nb, dtype=np.int32)
    brow[:n_cams] = np.arange(n_cams)
    diag_pos = bcol_ptr[n_cams + 1:] - 1
    brow[diag_pos] = n_cams + np.arange(n_pts, dtype=np.int32)
    mask = np.ones(nb, dtype=bool)
    mask[:n_cams] = False
    mask[diag_pos] = False
    off_pos = np.nonzero(mask)[0]
    brow[off_pos] = cam_s
    # values: cams 36 each, then per point column: k blocks of 18 + one of 9
    sz = np.empty(nb, dtype=np.int64)
    sz[:n_cams] = cd * cd
    sz[diag_pos] = pd_ * pd_
    sz[off_pos] = cd * pd_
    off = np.zeros(nb + 1, dtype=np.int64)
    np.cumsum(sz, out=off[1:])
    vals = np.empty(int(off[-1]))
    n_a = n_cams * cd * cd
    vals[:n_a] = Acc.transpose(0, 2, 1).ravel()
    # the landmark columns: per point its U blocks (sorted by camera: U_s is in storage order), then its C block; blocks
    # are column-major.  One masked assignment per kind instead of a scatter per block element.
    scalar_is_u = np.repeat(mask[n_cams:], sz[n_cams:])
    pts = vals[n_a:]
    pts[scalar_is_u] = U_s.transpose(0, 2, 1).ravel()
    pts[~scalar_is_u] = Cpp.transpose(0, 2, 1).ravel()
    cumsum = np.concatenate([np.arange(n_cams + 1, dtype=np.int64) * cd,
                             n_cams * cd + np.arange(1, n_pts + 1, dtype=np.int64) * pd_])
    rhs = rng.standard_normal(int(cumsum[-1]))
    return BlockSystem(cumsum, bcol_ptr, brow, vals, rhs, n_cams,
                       f"ba_{n_cams}x{n_pts}_{mode}_k{k}")


def indefinite(n: int = 40, d: int = 6, seed: int = 5) -> BlockSystem:
    """A small system that is *not* positive definite (negative test: solvers must return false)."""
    s = pose_chain(n=n, d=d, loop_every=10, loop_min=3, loop_max=8, seed=seed)
    off = s.block_value_offsets()
    j = n // 2
    diag_block = int(s.bcol_ptr[j + 1] - 1)
    blk = s.values[off[diag_block]:off[diag_block + 1]].reshape(d, d)
    blk -= 50.0 * np.eye(d)
    return dataclasses.replace(s, name=f"indefinite_n{n}_d{d}")


# ------------------------------------------------------------------------------------------------
# edge sets: the per-edge inputs of Lambda assembly (Jacobians, Sigma^-1, errors, robust weights)
# ------------------------------------------------------------------------------------------------

@dataclasses.dataclass
class EdgeSet:
    """One homogeneous set of binary edges at a linearization point -- what the reference's
    Calculate_Hessians_v2 works from (BaseTypes_Binary.h:759-777).  Every matrix is column-major:
    J0[e] is (rd x d0) stored as J0[e, col, row]."""
    n_verts: int
    v0: np.ndarray          # int64 [n_edges]
    v1: np.ndarray          # int64 [n_edges]
    J0: np.ndarray          # float64 [n_edges, d0, rd]   (column-major rd x d0)
    J1: np.ndarray          # float64 [n_edges, d1, rd]
    sigma_inv: np.ndarray   # float64 [n_edges, rd, rd]
    err: np.ndarray         # float64 [n_edges, rd]
    weight: Optional[np.ndarray] = None   # float64 [n_edges] robust weights, None = 1
    unary_vertex: int = 0
    unary_factor: Optional[np.ndarray] = None   # [d, d] column-major (stored transposed, as the others)
    unary_error: Optional[np.ndarray] = None    # [d]

    @property
    def n_edges(self) -> int:
        return int(self.v0.shape[0])

    @property
    def rd(self) -> int:
        return int(self.err.shape[1])

    @staticmethod
    def load(path: str) -> "EdgeSet":
        """Reads an SPPASM01 edge-set file (the format tests/golden/make_golden.py converts into fixtures)."""
        with open(path, "rb") as f:
            if f.read(8) != b"SPPASM01":
                raise ValueError("not an SPPASM01 file")
            n_verts, n_edges, d, rd = (int(x) for x in np.fromfile(f, np.int64, 4))
            v0 = np.fromfile(f, np.int64, n_edges)
            v1 = np.fromfile(f, np.int64, n_edges)
            J0 = np.fromfile(f, np.float64, n_edges * rd * d).reshape(n_edges, d, rd)
            J1 = np.fromfile(f, np.float64, n_edges * rd * d).reshape(n_edges, d, rd)
            si = np.fromfile(f, np.float64, n_edges * rd * rd).reshape(n_edges, rd, rd)
            err = np.fromfile(f, np.float64, n_edges * rd).reshape(n_edges, rd)
            w = np.fromfile(f, np.float64, n_edges)
            uf = np.fromfile(f, np.float64, d * d).reshape(d, d)
            ue = np.fromfile(f, np.float64, d)
        return EdgeSet(n_verts, v0, v1, J0, J1, si, err, w, 0, uf, ue)


def structure_from_edges(dims: np.ndarray, v0: np.ndarray, v1: np.ndarray) -> BlockSystem:
    """The upper block structure of Lambda for a graph: a diagonal block per vertex and a block
    (min, max) per distinct vertex pair (NonlinearSolver_Lambda_Base.h:1634-1660); values zero."""
    dims = np.asarray(dims, dtype=np.int64)
    n = int(dims.shape[0])
    r = np.minimum(v0, v1).astype(np.int64)
    c = np.maximum(v0, v1).astype(np.int64)
    key = np.unique(np.concatenate([c * n + r, np.arange(n, dtype=np.int64) * (n + 1)]))
    cols, rows = key // n, key % n
    bcol_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(bcol_ptr, cols + 1, 1)
    bcol_ptr = np.cumsum(bcol_ptr)
    cumsum = np.concatenate([[0], np.cumsum(dims)]).astype(np.int64)
    n_values = int(np.sum(dims[rows] * dims[cols]))
    return BlockSystem(cumsum=cumsum, bcol_ptr=bcol_ptr, brow_idx=rows.astype(np.int32),
                       values=np.zeros(n_values), rhs=np.zeros(int(cumsum[-1])), n_matrix_cut=0)


def random_edge_set(dims: np.ndarray, v0: np.ndarray, v1: np.ndarray, rd: int, seed: int = 0,
                    robust: bool = True, anchor: int = 0) -> EdgeSet:
    """Seeded synthetic linearization: Jacobians near [-I, +I]-like full-rank blocks, SPD Sigma^-1,
    small errors, robust weights in (0.5, 1] -- Lambda comes out positive definite thanks to the anchor."""
    rng = np.random.default_rng(seed)
    dims = np.asarray(dims, dtype=np.int64)
    v0 = np.asarray(v0, dtype=np.int64)
    v1 = np.asarray(v1, dtype=np.int64)
    ne = int(v0.shape[0])
    d0, d1 = int(dims[v0[0]]), int(dims[v1[0]])

    def jac(d, sign):
        J = 0.2 * rng.standard_normal((ne, d, rd))
        k = min(d, rd)
        J[:, np.arange(k), np.arange(k)] += sign
        return J
    A = 0.3 * rng.standard_normal((ne, rd, rd))
    si = np.einsum("eij,ekj->eik", A, A) + np.eye(rd)[None] * (1.0 + rng.random((ne, 1, 1)) * 20)
    d_a = int(dims[anchor])
    return EdgeSet(int(dims.shape[0]), v0, v1, jac(d0, -1.0), jac(d1, 1.0), si,
                   0.05 * rng.standard_normal((ne, rd)),
                   (0.5 + 0.5 * rng.random(ne)) if robust else None,
                   anchor, np.eye(d_a) * 100.0, np.zeros(d_a))


# ------------------------------------------------------------------------------------------------
# interop: the reference's matrix dumps (-dsm: MatrixMarket values + .bla block layout)
# ------------------------------------------------------------------------------------------------

def load_matrix_market(mtx_path: str, bla_path: str, rhs: Optional[np.ndarray] = None,
                       n_matrix_cut: int = 0) -> BlockSystem:
    """Reads a system matrix dumped by the reference (``CUberBlockMatrix::Save_MatrixMarket`` +
    ``Save_BlockLayout``, /root/reference/src/slam/BlockMatrix.cpp:12063-12199; what ``-dsm`` and
    ``slam_schur_orderings`` exchange).  The .bla layout is four lines: ``rows x cols (nnz)``, ``brows x bcols
    (nblocks)``, the block-row and the block-column cumulative sums.  The .mtx holds every scalar of every stored
    block (zeros included); symmetric dumps hold the upper triangle written as MatrixMarket *lower* entries
    (``col row value``), general dumps all entries.  Only the upper block triangle is kept, and the diagonal blocks
    are completed by symmetry.  ``rhs`` (not part of a dump) defaults to zeros."""
    with open(bla_path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    rows_cs = np.array(lines[2].split(), dtype=np.int64)
    cols_cs = np.array(lines[3].split(), dtype=np.int64)
    if not np.array_equal(rows_cs, cols_cs) or rows_cs[0] != 0:
        raise ValueError("not a square symmetric block layout")
    cumsum = cols_cs
    n = int(cumsum.shape[0]) - 1
    with open(mtx_path) as f:
        header = f.readline()
        if not header.startswith("%%MatrixMarket matrix coordinate real"):
            raise ValueError("not a real coordinate MatrixMarket file")
        b_symmetric = "symmetric" in header
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        n_rows, n_cols, n_nnz = (int(t) for t in line.split())
        data = np.loadtxt(f, dtype=np.float64, ndmin=2) if n_nnz else np.zeros((0, 3))
    if n_rows != cumsum[-1] or n_cols != cumsum[-1] or data.shape[0] != n_nnz:
        raise ValueError("matrix and layout disagree")
    i = data[:, 0].astype(np.int64) - 1
    j = data[:, 1].astype(np.int64) - 1
    v = data[:, 2]
    if b_symmetric:      # stored as (larger index, smaller index): the upper entry is (j, i)
        i, j = np.minimum(i, j), np.maximum(i, j)
    else:
        keep = i <= j
        i, j, v = i[keep], j[keep], v[keep]
    bi = np.searchsorted(cumsum, i, side="right") - 1
    bj = np.searchsorted(cumsum, j, side="right") - 1
    key = np.unique(bj * n + bi)
    bcols, brows = key // n, key % n
    bcol_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(bcol_ptr, bcols + 1, 1)
    bcol_ptr = np.cumsum(bcol_ptr)
    dims = np.diff(cumsum)
    sizes = dims[brows] * dims[bcols]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    values = np.zeros(int(offs[-1]))
    k = np.searchsorted(key, bj * n + bi)                 # block of every scalar
    pos = offs[k] + (i - cumsum[bi]) + (j - cumsum[bj]) * dims[bi]
    values[pos] = v
    diag = bi == bj                                       # mirror the diagonal blocks' upper triangle
    values[offs[k[diag]] + (j[diag] - cumsum[bj[diag]]) + (i[diag] - cumsum[bi[diag]]) * dims[bi[diag]]] = v[diag]
    return BlockSystem(cumsum=cumsum, bcol_ptr=bcol_ptr, brow_idx=brows.astype(np.int32), values=values,
                       rhs=np.zeros(int(cumsum[-1])) if rhs is None else np.asarray(rhs, dtype=np.float64),
                       n_matrix_cut=int(n_matrix_cut), name=os.path.basename(mtx_path))


def save_matrix_market(lam: BlockSystem, mtx_path: str, bla_path: str) -> None:
    """Writes the pair of files the reference's ``Load_MatrixMarket`` / ``Load_BlockLayout`` read: the upper triangle as
    a symmetric MatrixMarket matrix (entries ``col row value``) and the four-line block layout."""
    off = lam.block_value_offsets()
    dims = np.diff(lam.cumsum)
    col = np.repeat(np.arange(lam.n_bcols), np.diff(lam.bcol_ptr))
    rows, cols, vals = [], [], []
    for k in range(lam.n_blocks):
        r, c = int(lam.brow_idx[k]), int(col[k])
        blk = lam.values[off[k]:off[k + 1]].reshape(dims[c], dims[r]).T      # [row, col]
        ii, jj = np.meshgrid(np.arange(dims[r]), np.arange(dims[c]), indexing="ij")
        gi, gj = lam.cumsum[r] + ii, lam.cumsum[c] + jj
        keep = gj >= gi
        rows.append(gi[keep]); cols.append(gj[keep]); vals.append(blk[keep])
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    n = int(lam.cumsum[-1])
    with open(bla_path, "w") as f:
        f.write(f"{n} x {n} ({int(off[-1])})\n{lam.n_bcols} x {lam.n_bcols} ({lam.n_blocks})\n")
        f.write(" ".join(str(int(x)) for x in lam.cumsum) + "\n")
        f.write(" ".join(str(int(x)) for x in lam.cumsum) + "\n")
    with open(mtx_path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n% block matrix dump (upper triangle)\n")
        f.write(f"{n} {n} {vals.shape[0]}\n")
        for a, b, x in zip(cols, rows, vals):
            f.write(f"{int(a) + 1} {int(b) + 1} {x:.17g}\n")
