"""The host planner under sanitizers (CPU only: GPU AddressSanitizer is not available on the pool).  csrc/plan.cpp has no
HIP dependency, so it is compiled here with g++ -fsanitize=address,undefined (and, second case, =thread: the dissection
runs the halves of large bisections on threads) next to a small driver that plans a chain with loop closures, a grid,
a random graph with isolated vertices and a disconnected chain, a band, a single vertex and a clique."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", ["address,undefined", "thread"])
def test_planner_is_clean_under_sanitizers(flags, tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "plan_sanitize"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + flags, "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "slam_plus_plus_amd", "csrc"), os.path.join(ROOT, "tests", "plan_sanitize_driver.cpp"),
           os.path.join(ROOT, "slam_plus_plus_amd", "csrc", "plan.cpp"), "-o", str(exe), "-lpthread"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitizer" in (build.stderr or "").lower() and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
    lines = [l for l in run.stdout.splitlines() if ": n " in l]
    assert len(lines) == 6 and all("err ''" in l for l in lines), run.stdout
