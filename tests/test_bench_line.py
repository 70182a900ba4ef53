"""bench.py's stdout contract without a GPU: the compact line stays under the driver's 8 KB window whatever the full record
holds (round 3's 21.5 KB line was not parsed), and `--gpus N` never degrades to a silent one-GPU run."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_compact_line_of_every_committed_full_record():
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench*.json")))
    files = [f for f in files if "under_rocprof" not in f]
    assert files
    for f in files:
        d = json.load(open(f))
        if "metric" not in d:
            continue
        if d.get("ba_c5") and d["ba_c5"].get("phases_ms"):
            d["scaling_model"] = {key: bench.scaling_model(d[key], d[key].get("n_values", 145e6), d[key].get("n_scalars", 6e6),
                                                           d[key].get("n_exchange_doubles", 3e5))
                                  for key in ("ba_c5", "ba_1k_1m") if d.get(key)}
        text = bench.compact_line(d, "gpurun_out/bench_full_n1.json")
        assert len(text) < bench.COMPACT_LIMIT and "\n" not in text
        line = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in line, (f, k)
        assert line["roofline"]["frac"] > 0 and line["value"] > 0
        if "scaling_model" in line:   # both readings of the north star's 8-GPU target, for every strong-scaling system
            for m in line["scaling_model"].values():
                assert "host_arrays" in m and m["host_arrays"]["8"] > m["device_resident"]["8"] > 1.0


def test_compact_line_sheds_optional_parts_rather_than_grow():
    import bench
    d = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_bench.json")))[0]))
    d["other_configs"] = {f"X{i}": {"workload": "w" * 200, "ms_per_solve": 1.0, "roofline": {"bound": "hbm", "frac": 0.1}} for i in range(200)}
    text = bench.compact_line(d, None)
    assert len(text) < bench.COMPACT_LIMIT
    assert json.loads(text)["roofline"]["frac"] > 0


def test_gpus_n_without_devices_fails_loudly():
    """No GPU in this container: `bench.py --gpus 2` must exit non-zero and print no JSON line (never n_gpus: 1 in disguise)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SLAMPP_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # ... and a launcher whose rank count disagrees with --gpus is refused as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
