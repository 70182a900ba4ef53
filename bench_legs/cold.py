"""analyze_ms_cold in a process of its own: `python -m bench_legs.cold <spec> ...` prints one JSON line {spec: [ms, ...]}.

What the first Solve_PosDef_Blocky of a structure pays before any arithmetic (set_structure + analyze) on a fresh handle, an idle
device and in a process that holds that one system.  Inside the bench process the same call takes up to twice as long for the
big BA systems (C5 86 against 54-63 ms, uniform 114 against 58): the analysis is host work on arrays of 10^6 - 10^7 records, and
what it costs depends on what the process's heap has been through (a dozen systems generated and dropped, gigabytes of
tensors) -- that is the bench's history, not the analysis.  The in-process number stays in the full record beside this one.
spec: chain:<poses> | manhattan:<n> | sphere:<a>:<b> | ba:<cams>:<points>:<mode>"""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(spec):
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP, CLinearSolver_Schur_HIP
    p = spec.split(":")
    if p[0] == "chain":
        return CLinearSolver_HIP, synth.pose_chain(n=int(p[1]))
    if p[0] == "manhattan":
        return CLinearSolver_HIP, synth.manhattan(int(p[1]))
    if p[0] == "sphere":
        return CLinearSolver_HIP, synth.sphere(int(p[1]), int(p[2]))
    if p[0] == "ba":
        return CLinearSolver_Schur_HIP, synth.ba(int(p[1]), int(p[2]), mode=p[3], seed=777)
    raise SystemExit(f"bench_legs.cold: unknown spec {spec}")


def main(specs, reps=3, settle_ms=30.0):
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
    warm = CLinearSolver_HIP()      # the process's first handle pays the runtime's start-up, not an analysis
    warm.SymbolicDecomposition_Blocky(synth.pose_chain(n=64))
    out = {}
    for spec in specs:
        cls, lam = make(spec)
        times = []
        for _ in range(reps):
            s = cls()
            time.sleep(settle_ms * 1e-3)     # (the driver is still releasing the previous handle's memory: bench_legs/common.py)
            t0 = time.perf_counter()
            s.SymbolicDecomposition_Blocky(lam)
            times.append((time.perf_counter() - t0) * 1e3)
            del s
        out[spec] = times
        del lam
    print(json.dumps(out), flush=True)


def run_in_subprocess(specs, timeout=900):
    """{spec: [ms, ...]} from a child process (bench.py: before its own legs, while the parent holds nothing: a child measured
    beside a parent that held the bench's systems and tensors saw C5 at 73-83 ms, on its own 52-58)."""
    import subprocess
    r = subprocess.run([sys.executable, "-m", "bench_legs.cold"] + list(specs), cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stdout + r.stderr)[-300:]}
    return json.loads(lines[-1])


if __name__ == "__main__":
    main(sys.argv[1:] or ["chain:100000"], reps=int(os.environ.get("REPS", "3")), settle_ms=float(os.environ.get("SETTLE_MS", "30")))
