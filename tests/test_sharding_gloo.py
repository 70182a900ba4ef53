"""The N > 1 path on CPU: landmark sharding of a BA system + the one exchange step (sum all-reduce of
the partial reduced camera systems), with torch.distributed/gloo, world_size 2.  The per-shard Schur
arithmetic is done by the CPU oracle here (no GPU in this container); on MI355X the same callback
hands the solver's device buffer to RCCL (bench.py make_allreduce)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from slam_plus_plus_amd import synth, sharding
    from oracle import oracle_lib as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lam = synth.ba(16, 400, mode="venice", seed=77)           # every rank builds the same global system
        shard, sl = sharding.landmark_shard(lam, rank, world)
        ok, x_s, S, rr = O.solve_schur(shard, want_S=True)          # partial S_r, r_r and (unused) local solve
        N = S.shape[0]
        buf = torch.from_numpy(np.concatenate([S.ravel(), rr]))   # [S | r], as the solver lays it out
        dist.all_reduce(buf)                                      # the one exchange step
        S_full = buf[:N * N].numpy().reshape(N, N)
        r_full = buf[N * N:].numpy()
        S_sym = S_full + np.triu(S_full, 1).T                     # oracle returns the upper triangle
        dx = np.linalg.solve(S_sym, r_full)                       # redundant on every rank
        # shard-local back-substitution dl = C^-1 (l - U^T dx), via the shard system with dx known:
        A_sh = shard.to_scipy().tocsr()
        n_x = N
        Cl = A_sh[n_x:, n_x:]
        Ul = A_sh[:n_x, n_x:]
        import scipy.sparse.linalg as spl
        dl = spl.spsolve(Cl.tocsc(), shard.rhs[n_x:] - Ul.T @ dx)
        q.put((rank, dx, sl.start, sl.stop, dl))
    finally:
        dist.destroy_process_group()


def test_two_rank_landmark_sharded_schur_matches_single_system():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from slam_plus_plus_amd import synth
    from oracle import oracle_lib as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lam = synth.ba(16, 400, mode="venice", seed=77)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    x = np.zeros_like(x_ref)
    for rank, dx, l0, l1, dl in results:
        assert np.abs(dx - x_ref[:n_x]).max() / np.abs(x_ref[:n_x]).max() < 1e-10   # every rank has the full dx
        x[:n_x] = dx
        x[l0:l1] = dl
    assert np.abs(x - x_ref).max() / np.abs(x_ref).max() < 1e-10


def test_shard_bounds_balance_observations():
    sys.path.insert(0, ROOT)
    from slam_plus_plus_amd import synth, sharding
    lam = synth.ba(20, 5000, mode="venice", seed=5)
    b = sharding.shard_bounds(lam, 8)
    assert b[0] == 0 and b[-1] == lam.n_bcols - lam.n_matrix_cut and np.all(np.diff(b) > 0)
    obs = np.diff(lam.bcol_ptr[lam.n_matrix_cut:]) - 1
    per = np.array([obs[b[i]:b[i + 1]].sum() for i in range(8)])
    assert per.max() / per.mean() < 1.05
    # shards tile the landmark range and their additive camera parts sum to the original
    tot = 0
    for r in range(8):
        sh, sl = sharding.landmark_shard(lam, r, 8)
        tot += sh.rhs[:int(lam.cumsum[lam.n_matrix_cut])]
        assert sh.n_matrix_cut == lam.n_matrix_cut
    assert np.allclose(tot, lam.rhs[:int(lam.cumsum[lam.n_matrix_cut])], rtol=1e-13)
