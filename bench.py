#!/usr/bin/env python3
"""bench.py -- Lambda-solve benchmark on MI355X (BASELINE.json metric: "Lambda solve wall-clock ms +
GFLOP/s vs roofline, 100k-pose SE(3) and BA Schur").

A "step" is one pass of the hot path over one synthetic system: numeric factorization of Lambda plus
the two substitutions, i.e. what every Gauss-Newton iteration after the first asks of
Solve_PosDef_Blocky (/root/reference/include/slam/NonlinearSolver_Lambda.h:618).  Lambda (packed
block values) and eta are resident in HBM when the timed region starts; PCIe-inclusive numbers
are in DESIGN.md, never here.

  python bench.py --gpus 1 --steps 20 --warmup 3            # default workload: C3, 100k-pose SE(3)
  python bench.py --workload ba                             # C4: 1k cams x 500k points, Schur path
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU (N > 1): the headline is STRONG scaling of one fixed BA system -- BASELINE config 5, 2 000 cameras x
2 000 000 landmarks (and, beside it, the north star's 1 000 x 1 000 000) -- cut into N landmark shards
(slam_plus_plus_amd/sharding.py, the rule of the library's own splitter), one rank per GPU, one RCCL all-reduce of the
packed blocks of the reduced camera system per step; "scaling": "strong".  The N = 1 line carries the same two systems
on one GPU (`ba_c5`, `ba_1k_1m`), so the curve has its first point.  The pose-graph factorization does not shard (one
elimination tree): at N > 1 it runs as N independent replicas and is reported beside the headline (`pose_graph_replicas`).

One JSON line on stdout (rank 0).  `value` = algorithmic GFLOP/s of the whole job; `ms_per_step` =
wall-clock per Lambda solve; `roofline` = dominant kernel against HBM peak, timed with HIP events
on the solver's stream inside the timed region; `cpu_baseline` = the compiled reference's CHOLMOD
path (oracle/_ref/ref_harness) on this box's host cores, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the legs (bench_legs/): re-exported here, so that `import bench` keeps offering what tools/ and tests/ use
from bench_legs.common import *  # noqa: E402,F401,F403
from bench_legs.common import _KEEP, _DevPtr  # noqa: E402,F401
from bench_legs.c3 import *  # noqa: E402,F401,F403
from bench_legs.small import *  # noqa: E402,F401,F403
from bench_legs.ba import *  # noqa: E402,F401,F403
from bench_legs.line import *  # noqa: E402,F401,F403
from bench_legs.line import _r, _short_roofline, _leg_summary  # noqa: E402,F401
from bench_legs.scaling import *  # noqa: E402,F401,F403


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves (torch.distributed.run, one per
    GPU) as a child -- nothing in this process has touched the GPU -- and leave with its code.  Fewer than N devices: exit
    non-zero, never a silent one-GPU run."""
    import socket
    import torch
    one_device = os.environ.get("SLAMPP_BENCH_ONE_DEVICE") == "1"
    n_dev = torch.cuda.device_count()      # (may initialise HIP in THIS process: it therefore only ever spawns children below, never execs)
    if n_dev < (1 if one_device else args.gpus):
        log(f"bench.py --gpus {args.gpus}: {n_dev} HIP device(s) visible; refusing to report {args.gpus} GPUs from fewer")
        return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    log("bench.py: starting", args.gpus, "ranks:", " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all", "c3", "ba", "small"])
    ap.add_argument("--small-configs", default="C1,C2", help="which of BASELINE configs[0] / configs[1] run_small_configs takes")
    ap.add_argument("--ba-cams", type=int, default=1000)
    ap.add_argument("--ba-points", type=int, default=500_000, help="landmarks of the C4 system")
    ap.add_argument("--ba-steps", type=int, default=5)
    ap.add_argument("--poses", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--c3-solve-only", action="store_true", help="leave out the legs beside the C3 solve (assembly, marginals, host path): tools/profile_round.sh")
    ap.add_argument("--ba-solve-only", action="store_true", help="leave out the legs beside the solve (host path, marginals, incremental update, assembly): the counter passes of tools/profile_round.sh, whose per-kernel averages should be the solve's")
    ap.add_argument("--ba-legs", default="venice,band,uniform", help="which visibility models the C4 part runs, the first as `ba_schur` (profiling runs one at a time)")
    ap.add_argument("--c5-cams", type=int, default=2000)
    ap.add_argument("--c5-points", type=int, default=2_000_000, help="landmarks of the fixed system `--gpus N` shards (BASELINE config 5)")
    ap.add_argument("--c5-mode", default="band", choices=["band", "venice", "uniform"])
    ap.add_argument("--target-cams", type=int, default=1000)
    ap.add_argument("--target-points", type=int, default=1_000_000, help="the north star's 1k-camera / 1M-point system, reported beside C5")
    ap.add_argument("--full-json", default=None, help="where the full record goes (default gpurun_out/bench_full_n<N>.json); the stdout line is the compact one")
    ap.add_argument("--group-reps", type=int, default=3, help="warm solves of the in-library device group leg (--gpus N, N > 1)")
    ap.add_argument("--no-group-leg", action="store_true", help="leave out the one-process device group leg at N > 1")
    ap.add_argument("--no-cold-process", action="store_true", help="leave out the cold analyses in a process of their own (bench_legs/cold.py): analyze_ms_cold is then the in-process number")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} started as one of {world} ranks: the launcher's --nproc-per-node and --gpus must agree")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # development aid: SLAMPP_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 with the gloo backend, so that the
    # multi-rank control flow (rendezvous, block-list agreement, max-over-ranks timing) can be run on a 1-GPU box
    one_device = os.environ.get("SLAMPP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    legs = [m for m in args.ba_legs.split(",") if m]
    cold_res, cold_specs = None, {}
    if world == 1 and args.workload == "all" and not args.ba_solve_only and not args.c3_solve_only and not args.no_cold_process:
        # the cold analyses first, in a process of their own, while this one holds nothing yet (bench_legs/cold.py: beside a parent
        # that held the bench's systems and tensors the child measured C5 at 73-83 ms, on its own at 52-58)
        from bench_legs import cold
        cold_specs = {"own_ordering": f"chain:{args.poses}", "C1": "manhattan:3500", "C2": "sphere:50:50",
                      "ba_schur": f"ba:{args.ba_cams}:{args.ba_points}:{legs[0]}", "ba_schur_band": f"ba:{args.ba_cams}:{args.ba_points}:band",
                      "ba_schur_uniform_dense_S": f"ba:{args.ba_cams}:{args.ba_points}:uniform", "ba_schur_venice": f"ba:{args.ba_cams}:{args.ba_points}:venice",
                      "ba_c5": f"ba:{args.c5_cams}:{args.c5_points}:{args.c5_mode}", "ba_1k_1m": f"ba:{args.target_cams}:{args.target_points}:{args.c5_mode}"}
        wanted = {"own_ordering", "C1", "C2", "ba_schur", "ba_c5", "ba_1k_1m"} | {k_ for k_, m in (("ba_schur_band", "band"), ("ba_schur_uniform_dense_S", "uniform"), ("ba_schur_venice", "venice")) if m in legs[1:]}
        wanted -= {k_ for k_ in ("C1", "C2") if k_ not in args.small_configs.split(",")}
        try:
            cold_res = cold.run_in_subprocess(sorted(set(cold_specs[k_] for k_ in wanted)))
        except Exception as e:
            cold_res = {"error": str(e)[:200]}
    import torch
    if not one_device and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py --gpus {world}: {torch.cuda.device_count()} HIP device(s) visible")
    dist = None
    # SLAMPP_BENCH_FORCE_DIST=1 exercises the RCCL plumbing (process group + all-reduce callback) with one rank
    if world > 1 or os.environ.get("SLAMPP_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        torch.cuda.set_device(local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    out = None

    def promote(ba, scaling):
        return {"metric": "BA Schur solve GFLOP/s (algorithmic flops / wall-clock)", "value": ba["GFLOP/s"],
                "unit": "GFLOP/s", "n_gpus": world, "steps": ba["steps"], "warmup": 1, "ms_per_step": ba["ms_per_step"],
                "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": ba["workload"], "parallelism": "1 GPU" if world == 1 else
                           f"{world} landmark shards of one fixed system, one rank per GPU, RCCL all-reduce of the reduced camera system"},
                "roofline": ba.get("roofline"), "cpu_baseline": ba.get("cpu_baseline")}

    if world > 1:
        # N > 1: strong scaling of the fixed C5 system (BASELINE config 5), the north star's 1k x 1M beside it; the pose
        # graph (does not shard) as N replicas, reported but not the headline
        strong = [(key, c, p) for key, c, p in (("ba_c5", args.c5_cams, args.c5_points), ("ba_1k_1m", args.target_cams, args.target_points))
                  if args.workload in ("all", "ba")]
        for i, (key, c, p) in enumerate(strong):
            leg = run_ba(args, rank, world, local_rank, dist, mode=args.c5_mode, extras=False, cams=c, points=p,
                         label="C5" if key == "ba_c5" else "north-star target")
            if rank == 0:
                if out is None:
                    out = promote(leg, "strong")
                out[key] = leg
        if args.workload in ("all", "c3"):
            c3 = run_c3(args, rank, world, local_rank, dist)
            if rank == 0:
                if out is None:
                    out = c3
                else:
                    out["pose_graph_replicas"] = {k_: c3[k_] for k_ in ("metric", "value", "unit", "ms_per_step", "scaling", "config")}
    else:
        if args.workload in ("all", "c3"):
            out = run_c3(args, rank, world, local_rank, dist)
        if args.workload == "small":
            small = run_small_configs(args, local_rank)
            first = small[sorted(small)[-1]]
            out = {"metric": "Lambda solve ms (numeric factor + 2 substitutions)", "value": first["ms_per_solve"], "unit": "ms",
                   "n_gpus": 1, "steps": 20, "warmup": 1, "ms_per_step": first["ms_per_solve"], "higher_is_better": False,
                   "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": first["workload"]}, "roofline": first.get("roofline"), "other_configs": small}
        if args.workload in ("all", "ba"):
            # BASELINE config 4 is "Venice-style": that leg is `ba_schur`; band and uniform visibility (SURVEY.md section 8d:
            # "band (sparse S) or uniformly (dense S -- report both)") ride beside it
            first = legs[0]
            ba = run_ba(args, rank, world, local_rank, dist, mode=first, extras=not args.ba_solve_only)
            if out is None:   # BA only: promote it to the headline
                out = promote(ba, "weak")
            out["ba_schur"] = ba
            if args.workload == "all":
                out["other_configs"] = run_small_configs(args, local_rank)
            for key, mode in (("ba_schur_band", "band"), ("ba_schur_uniform_dense_S", "uniform"), ("ba_schur_venice", "venice")):
                if mode in legs[1:]:
                    out[key] = run_ba(args, rank, world, local_rank, dist, mode=mode, extras=False)
            if args.workload == "all" and not args.ba_solve_only:
                # the N = 1 points of the strong-scaling curves `bench.py --gpus N` reports
                out["ba_c5"] = run_ba(args, rank, world, local_rank, dist, mode=args.c5_mode, extras=False, cams=args.c5_cams,
                                      points=args.c5_points, label="C5")
                out["ba_1k_1m"] = run_ba(args, rank, world, local_rank, dist, mode=args.c5_mode, extras=False, cams=args.target_cams,
                                         points=args.target_points, label="north-star target")
                # ... spelled out: `--gpus N`, N > 1, reports THIS metric on THIS system as its `value` (the pose graph of the N = 1
                # headline does not shard), so the efficiency of the curve is value(N) / strong_scaling_n1.value, not / `value`
                out["strong_scaling_n1"] = {"metric": "BA Schur solve GFLOP/s (algorithmic flops / wall-clock)",
                                            "value": out["ba_c5"].get("GFLOP/s"), "unit": "GFLOP/s",
                                            "ms_per_step": out["ba_c5"].get("ms_per_step"),
                                            "workload": out["ba_c5"].get("workload"),
                                            "note": "the N = 1 point of the curve bench.py --gpus N (N > 1) reports as `value`"}
    if rank == 0 and world == 1 and out is not None and out.get("ba_c5"):
        # both strong-scaling systems, device-resident and from host arrays
        out["scaling_model"] = {key: scaling_model(out[key], out[key].get("n_values"), out[key].get("n_scalars"),
                                                   out[key].get("n_exchange_doubles"))
                                for key in ("ba_c5", "ba_1k_1m") if out.get(key)}
        # ... and what the other two visibility models say: the same model on the C4-size legs of this run
        # (1k cameras x 500k points, Venice-like and uniform) -- with a reduced camera system that is dense or nearly so the
        # replicated factorization is most of the step and the landmark shards stop mattering; band visibility is the one
        # whose reduced system is sparse
        for key, name in (("ba_schur", "c4_" + legs[0]), ("ba_schur_band", "c4_band"), ("ba_schur_uniform_dense_S", "c4_uniform"),
                          ("ba_schur_venice", "c4_venice")):
            if out.get(key) and out[key].get("phases_ms"):
                out["scaling_model"].setdefault(name, scaling_model(out[key], out[key].get("n_values"), out[key].get("n_scalars"),
                                                                    out[key].get("n_exchange_doubles")))
    if cold_res is not None and out is not None:
        # analyze_ms_cold of every leg from a process that holds nothing but that leg's system (bench_legs/cold.py says why; it ran
        # before this process touched the device); this process's own measurement stays beside it
        where = {k_: (out.get("other_configs", {}).get(k_) if k_ in ("C1", "C2") else out.get(k_)) for k_ in cold_specs}
        where = {k_: v for k_, v in where.items() if isinstance(v, dict) and v.get("analyze_ms_cold") is not None}
        if "error" in cold_res:
            out["analyze_ms_cold_process"] = cold_res
        else:
            for k_, leg in where.items():
                t = cold_res.get(cold_specs[k_])
                if t:
                    leg["analyze_ms_cold_in_bench_process"] = leg["analyze_ms_cold"]
                    leg["analyze_ms_cold"] = float(sorted(t)[len(t) // 2])
                    leg["analyze_ms_cold_all"] = t
    if rank == 0 and world > 1 and out is not None:
        out["rccl_ranks"] = dist.get_world_size()
        out["dist_backend"] = dist.get_backend()      # "nccl" is RCCL on ROCm; "gloo" only under SLAMPP_BENCH_ONE_DEVICE=1 (development)
    if dist is not None:
        dist.destroy_process_group()
    if rank != 0:
        return
    if world > 1 and out is not None and args.workload in ("all", "ba") and not args.no_group_leg:
        # the same fixed C5 system through ONE handle over the N devices (what SLAMPP_HIP_DEVICES gives an unchanged SLAM++
        # binary), after the process group is gone: the other ranks have left their devices
        time.sleep(1.0)
        try:
            out["device_group"] = device_group_leg(args, world, one_device, lam=_KEEP.get("C5"))
            out["exchange_in_library"] = out["device_group"]["exchange"]
            # the number the north star's ">= 4x at 8 GPUs" can be read on: a caller's host arrays through ONE handle over the
            # N devices against the same call on one device (PCIe inclusive); `value` stays the device-resident curve
            out["host_path_speedup_vs_single_device"] = out["device_group"].get("speedup_vs_single_device")
            out["north_star_4x_read_on"] = "host_path_speedup_vs_single_device (host arrays, N PCIe links); `value` is device-resident and capped by the replicated reduced solve (scaling_model in the N = 1 line)"
        except Exception as e:
            out["device_group"] = {"error": str(e)[:300]}
    if out is not None:
        try:    # RCCL prints its version banner through C stdio: get it out before the one JSON line, not after it
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        full_path = args.full_json or os.path.join("gpurun_out", f"bench_full_n{world}.json")
        try:
            os.makedirs(os.path.dirname(os.path.join(ROOT, full_path)) or ".", exist_ok=True)
            with open(os.path.join(ROOT, full_path), "w") as f:
                json.dump(out, f)
        except OSError as e:
            log("bench.py: could not write", full_path, e)
            full_path = None
        print(compact_line(out, full_path), flush=True)


if __name__ == "__main__":
    main()
