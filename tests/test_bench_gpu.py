"""bench.py itself, at toy sizes: the one JSON line with its roofline objects, and the two-rank path (two processes on one GPU
over gloo: SLAMPP_BENCH_ONE_DEVICE) -- every rank has to walk through the same solves, the BA solve holds a collective."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--poses", "3000", "--ba-cams", "60", "--ba-points", "4000", "--steps", "3", "--warmup", "1", "--ba-steps", "2",
         "--no-cpu-baseline", "--ba-legs", "band", "--c5-cams", "60", "--c5-points", "6000", "--target-cams", "40", "--target-points", "3000"]


def last_json(text, full=True):
    """The compact stdout line (checked: under 8 KB, carries the contract's keys) merged over the full record it points to."""
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    assert len(lines[-1]) < 8000, len(lines[-1])
    line = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "full"):
        assert k in line, k
    if not full:
        return line
    d = json.load(open(os.path.join(ROOT, line["full"])))
    for k in ("metric", "n_gpus", "scaling"):
        assert d[k] == line[k], k
    assert abs(d["value"] - line["value"]) <= 1e-3 * abs(d["value"])
    return d


@pytest.mark.parametrize("workload", ["c3", "ba"])
def test_bench_line(workload):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--full-json", f"gpurun_out/test_bench_{workload}.json"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = last_json(r.stdout, full=False)
    assert line["roofline"]["frac"] > 0 and line["ms_per_step"] > 0
    d = last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0
    if workload == "ba":
        assert d["ba_schur"]["solve_residual_rel_inf"] < 1e-9
    else:
        assert d["solve_residual_rel_inf"] < 1e-9


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_bench_ranks_on_one_device_strong_scaling(world):
    """`--gpus N`: one fixed BA system cut into N landmark shards (strong scaling), every rank's phases with the all-reduce in
    them, and the residual of the FULL system assembled from the ranks' pieces -- the parity guard of the sharded solve."""
    env = dict(os.environ, SLAMPP_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--full-json",
           f"gpurun_out/test_bench_ranks{world}.json"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = last_json(r.stdout, full=False)
    assert line["legs"]["device_group"]["members"] == world and line["legs"]["device_group"]["resid"] < 1e-9
    d = last_json(r.stdout)
    g = d["device_group"]     # one process, one handle over `world` members (here: all on device 0, peer exchange)
    assert g["ok"] and g["members"] == world and g["exchange"] and g["resid"] < 1e-9, g
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["value"] > 0
    assert "C5" in d["config"]["workload"] and "60 cams x 6000 points" in d["config"]["workload"]
    for key in ("ba_c5", "ba_1k_1m"):
        leg = d[key]
        assert leg["n_gpus"] == world and len(leg["phases_ms_by_rank"]) == world
        assert all("allreduce" in ph for ph in leg["phases_ms_by_rank"])
        assert leg["solve_residual_rel_inf"] < 1e-9, leg["solve_residual_rel_inf"]
    assert d["pose_graph_replicas"]["scaling"] == "weak"


def test_bench_default_run_carries_the_n1_points_of_the_scaling_curves():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-json", "gpurun_out/test_bench_default.json"] + SMALL,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = last_json(r.stdout, full=False)
    assert set(("ba_schur", "ba_c5", "ba_1k_1m")) <= set(line["legs"])
    # the model of BOTH strong-scaling systems, device-resident and from host arrays (where the 8 PCIe links are)
    assert {"ba_c5", "ba_1k_1m"} <= set(line["scaling_model"])
    for key, m in line["scaling_model"].items():
        if key.startswith("c4_"):    # (round 6) the other visibility models, from the C4-size legs of the run: the 8-GPU figures
            assert m["serial_ms"] > 0 and m["device_resident_at_8"] > 0 and m["host_arrays_at_8"] > m["device_resident_at_8"]
            continue
        assert m["serial_ms"] > 0 and set(m["device_resident"]) == set(m["host_arrays"]) == {"2", "4", "8"}
        assert m["host_arrays"]["8"] > m["device_resident"]["8"]
    assert any(key.startswith("c4_") for key in line["scaling_model"])
    # K concurrent solves on one device ("replicas only"): every K reported, all of them solved
    assert set(line["legs"]["replicas_one_gpu"]) == {"1", "2", "4", "8"}
    assert all(v["GFLOP/s"] > 0 for v in line["legs"]["replicas_one_gpu"].values())
    d = last_json(r.stdout)
    assert d["n_gpus"] == 1 and "C3" in d["config"]["workload"]
    for key in ("ba_schur", "ba_c5", "ba_1k_1m"):
        assert d[key]["n_gpus"] == 1 and d[key]["solve_residual_rel_inf"] < 1e-9, key
        # (round 6) analyze_ms_cold comes from a process of its own (bench_legs/cold.py), this process's measurement beside it
        assert len(d[key]["analyze_ms_cold_all"]) == 3 and d[key]["analyze_ms_cold"] in d[key]["analyze_ms_cold_all"], key
        assert d[key]["analyze_ms_cold_in_bench_process"] > 0, key
    assert d["own_ordering"]["analyze_ms_cold_in_bench_process"] > 0 and "analyze_ms_cold_process" not in d


def test_bench_c3_with_the_reference_prints_conditioning_beside_the_error():
    """"the acceptance report must print cond-proxy + inter-oracle spread beside our error" -- and the
    three like-for-like ratios that replaced round 4's single speed-up figure."""
    from oracle import oracle_lib as O
    if not O.have_reference():
        pytest.skip("oracle/_ref/ref_harness not built")
    args = [a for a in SMALL if a != "--no-cpu-baseline"] + ["--workload", "c3"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--full-json", "gpurun_out/test_bench_c3_ref.json"] + args,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = last_json(r.stdout, full=False)
    assert line["cond_proxy"] > 1 and 0 < line["inter_oracle_spread"] < 1e-9 and line["solve_x_vs_reference_rel_inf"] < 1e-10
    assert "numeric_phases_device_resident" in line["speedup_vs_reference"]


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` by itself: the ranks are started by bench.py (here both on device 0: SLAMPP_BENCH_ONE_DEVICE), the
    line says n_gpus 2 -- never a silent one-GPU run."""
    env = dict(os.environ, SLAMPP_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ba", "--full-json",
                        "gpurun_out/test_bench_spawn2.json"] + SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = last_json(r.stdout, full=False)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["rccl_ranks"] == 2 and line["dist_backend"] == "gloo"      # (gloo: only under SLAMPP_BENCH_ONE_DEVICE=1)
    assert line["host_path_speedup_vs_single_device"] > 0 and "host_path" in line["north_star_4x_read_on"]


def test_bench_refuses_more_gpus_than_there_are():
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SLAMPP_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1)] + SMALL, capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout[-500:]
