"""The multi-rank BA path of the *product*, as processes: two ranks (spawned processes, both on this box's one GPU),
torch.distributed with the gloo backend, each rank a CLinearSolver_Schur_HIP on its landmark shard whose all-reduce
callback (slampp_hip_set_allreduce) sums the library's device buffer over the ranks -- the block-list agreement, the
packed exchange, the redundant reduced solve and the shard-local back-substitution, end to end.  On an 8-GPU node the
same callback hands the buffer to RCCL (bench.py make_allreduce); tests/test_sharding_gloo.py checks the sharding
algebra alone, with the CPU oracle doing the arithmetic."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _system(kind):
    from slam_plus_plus_amd import synth
    if kind == "sparse_S":
        return synth.ba(150, 8000, k=4, mode="band", seed=31)       # few camera pairs share landmarks: sparse reduced solve
    return synth.ba(40, 4000, mode="venice", seed=21)                # dense reduced system


def _worker(rank, world, port, kind, poison_rank, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from slam_plus_plus_amd import sharding
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        lam = _system(kind)
        shard, sl = sharding.landmark_shard(lam, rank, world)
        if rank == poison_rank:                                    # one landmark block of this rank only is indefinite
            off = shard.block_value_offsets()
            last = shard.n_blocks - 1
            shard.values[off[last]:off[last + 1]] -= 50.0 * np.eye(3).ravel()
        n_calls = [0]

        class DevPtr:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        def allreduce(ptr, count, stream):
            torch.cuda.synchronize()                               # the solver's stream has produced the partial sums
            t = torch.as_tensor(DevPtr(ptr, count), device="cuda")
            h = t.cpu()
            dist.all_reduce(h)                                     # gloo, on the host
            t.copy_(h)
            torch.cuda.synchronize()
            n_calls[0] += 1
            return 0

        solver = CLinearSolver_Schur_HIP()
        solver.set_option("shard_rank", rank)
        solver.set_option("shard_world", world)
        solver.set_allreduce(allreduce)
        eta = shard.rhs.copy()
        ok = solver.Solve_PosDef(shard, eta)
        eta2 = shard.rhs.copy()
        ok2 = solver.Solve_PosDef_Blocky(shard, eta2)             # second step: the agreed block list is reused
        q.put((rank, bool(ok), bool(ok2), eta, eta2, sl.start, sl.stop, n_calls[0], solver.stats()["n_points"]))
    finally:
        dist.destroy_process_group()


def _run(kind, poison_rank=-1):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, poison_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(results)


@pytest.mark.parametrize("kind", ["dense_S", "sparse_S"])
def test_two_processes_one_gpu_match_the_single_system(kind):
    from oracle import oracle_lib as O
    lam = _system(kind)
    ok, x_ref, _, _ = O.solve_schur(lam)
    assert ok
    results = _run(kind)
    n_x = int(lam.cumsum[lam.n_matrix_cut])
    x = np.zeros_like(x_ref)
    for rank, ok1, ok2, eta, eta2, l0, l1, n_calls, n_pts in results:
        assert ok1 and ok2 and n_calls >= 3                       # block-list agreement (2 calls) + one per step
        assert np.array_equal(eta, eta2)
        assert np.abs(eta[:n_x] - x_ref[:n_x]).max() / np.abs(x_ref[:n_x]).max() < 1e-10   # every rank has the full dx
        x[:n_x] = eta[:n_x]
        x[l0:l1] = eta[n_x:]
    assert sum(r[8] for r in results) == lam.n_bcols - lam.n_matrix_cut
    assert np.abs(x - x_ref).max() / np.abs(x_ref).max() < 1e-10


@pytest.mark.parametrize("kind", ["dense_S", "sparse_S"])
def test_not_positive_definite_on_one_rank_is_seen_by_all(kind):
    """A landmark block that is not positive definite on rank 1 alone: both ranks must return false (the status travels
    with the exchanged data), or a caller's LM loop would carry on with one rank missing from the next collective."""
    results = _run(kind, poison_rank=1)
    for rank, ok1, ok2, *_ in results:
        assert ok1 is False and ok2 is False, (rank, ok1, ok2)
