"""bench.py itself, at toy sizes: the one JSON line with its roofline objects, and the two-rank path (two processes on one GPU
over gloo: SLAMPP_BENCH_ONE_DEVICE) -- every rank has to walk through the same solves, the BA solve holds a collective."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--poses", "3000", "--ba-cams", "60", "--ba-points", "4000", "--steps", "3", "--warmup", "1", "--ba-steps", "2",
         "--no-cpu-baseline", "--ba-legs", "band"]


def last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("workload", ["c3", "ba"])
def test_bench_line(workload):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0
    if workload == "ba":
        assert d["ba_schur"]["solve_residual_rel_inf"] < 1e-9
    else:
        assert d["solve_residual_rel_inf"] < 1e-9


def test_bench_two_ranks_on_one_device():
    env = dict(os.environ, SLAMPP_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["ba_schur"]["n_gpus"] == 2 and "allreduce" in d["ba_schur"]["phases_ms"]
