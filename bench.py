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

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (vendor figure quoted in SURVEY.md section 8d)
# what a loop of nothing but v_mfma_f64_16x16x4 sustains on this chip (tools/micro/mfma_f64_peak.hip, profiles/r05_mfma_f64_peak.txt:
# 105 clocks per instruction per SIMD at 2.4 GHz; the vector unit's v_fma_f64 63-70): printed beside `peak`, never instead of it
F64_MFMA_SUSTAINED_TFLOPS = 47.8

# Algorithmic work of the default C3 instance (synth.pose_chain(), seed 12345), counted by the
# reference's own CHOLMOD (AMD ordering) with oracle/_ref/ref_harness cholmod_phases in the build
# container: Common.lnz, Common.fl, nnz(triu(Lambda)).  SURVEY.md section 8d convention:
# factor flops = fl, solve flops = 4 lnz, factor bytes = 8 (nnz + lnz), solve bytes = 16 lnz + 32 n.
C3_REF = {"n": 600_000, "nnz_triu": 7_271_928, "lnz": 8_360_436, "fl": 122_411_332.0}


_KEEP = {}


def host_cores():
    from oracle import oracle_lib as O
    return O.host_cores()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_counts_c3(lam, stats):
    if lam.n_bcols == 100_000 and lam.n_blocks == 201_998:
        c = dict(C3_REF)
        c["source"] = "reference CHOLMOD/AMD counters (fl, lnz) on this instance"
    else:  # other sizes: our own ordering's counts (upper bound on the reference's)
        c = {"n": lam.n_scalars, "nnz_triu": stats["nnz_upper"], "lnz": stats["l_nnz"],
             "fl": stats["factor_flops"], "source": "own ordering"}
    c["flops"] = c["fl"] + 4.0 * c["lnz"]
    c["factor_bytes"] = 8.0 * (c["nnz_triu"] + c["lnz"])
    c["solve_bytes"] = 16.0 * c["lnz"] + 32.0 * c["n"]
    return c


def load_traffic(workload):
    """HBM traffic per launch from the committed PMC summary of this round (tools/profile_round.sh ->
    profiles/*_traffic.json): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB units,
    FETCH_SIZE doubled as the gfx950 note of MI355X_MICROARCH.md prescribes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_traffic.json")))
    if not files:
        return {}, None
    return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)


def kernel_traffic(traffic, needle):
    for name, v in traffic.items():
        if needle in name:
            return v["hbm_bytes_per_launch_corrected"]
    return None


def kernel_traffic_mean(traffic, needle):
    """Bytes per launch over every kernel whose name holds the needle, weighted by their launches (template variants of one
    kernel with different launch counts: the four- and eight-wave slice kernels of one step)."""
    parts = [(v["hbm_bytes_per_launch_corrected"], v["launches"]) for name, v in traffic.items() if needle in name]
    n_launches = sum(n for _, n in parts)
    return (sum(b * n for b, n in parts) / n_launches) if n_launches else None


def kernel_traffic_sum(traffic, needle):
    """Bytes per step of every kernel whose name holds the needle (template variants of one kernel: each runs once a step)."""
    parts = [v["hbm_bytes_per_launch_corrected"] for name, v in traffic.items() if needle in name]
    return sum(parts) if parts else None


def per_kernel_bytes_sparse(plan, n_bottom_stages=1):
    """Algorithmic bytes each kernel of the sparse path moves per step, from the plan (our own
    factor structure): factor = 8 (nnz of the Lambda blocks read + nnz of the L columns written),
    substitution = 8 nnz(L columns) + vectors, split into the bottom-stage launch and the rest."""
    dim = plan["dim"].astype(np.int64)
    lptr, lrow, asrc = plan["lptr"], plan["lrow"].astype(np.int64), plan["asrc"]
    n = len(dim)
    col_of = np.repeat(np.arange(n), np.diff(lptr))
    is_diag = np.zeros(len(lrow), dtype=bool)
    is_diag[lptr[:-1]] = True
    dj, di = dim[col_of], dim[lrow]
    blk_nnz = np.where(is_diag, dj * (dj + 1) // 2, di * dj)
    l_col = np.bincount(col_of, weights=blk_nnz, minlength=n)
    a_col = np.bincount(col_of, weights=np.where(asrc >= 0, blk_nnz, 0), minlength=n)
    def cols_of(s0, s1):
        m = np.zeros(n, dtype=bool)
        t0, t1 = plan["stage_ptr"][s0], plan["stage_ptr"][s1]
        m[plan["task_cols"][plan["task_ptr"][t0]:plan["task_ptr"][t1]]] = True
        return m
    n_stages = len(plan["stage_ptr"]) - 1
    leaves = cols_of(0, 1)                                         # stage 0: the leaf subtrees (lane-per-task kernel)
    wide = cols_of(1, n_bottom_stages) if n_bottom_stages > 1 else np.zeros(n, dtype=bool)   # wave-per-task kernel
    upper = cols_of(n_bottom_stages, n_stages) if n_bottom_stages < n_stages else np.zeros(n, dtype=bool)
    fac = 8.0 * (l_col + a_col) + 16.0 * dim     # + the fused forward substitution's vector traffic
    sub = 8.0 * l_col + 16.0 * dim               # a substitution reads the L column, reads + writes the vector
    return {"factor_leaves": float(fac[leaves].sum()), "factor_wide": float(fac[wide].sum()),
            "factor_upper": float(fac[upper].sum()), "forward": float(sub.sum()), "backward": float(sub.sum()),
            "leaf_cols": int(leaves.sum())}


def cpu_baseline_c3(lam, counts, x_gpu, budget_reps=12):
    """The compiled reference on this box's host, on the same system: (a) what its nonlinear solver pays per iteration
    with CLinearSolver_CholMod -- Solve_PosDef, whose tag is "basic": conversion, ordering and symbolic analysis re-run on
    every call (LinearSolver_CholMod.h:86-94) -- is `value`; (b) the numeric phases alone (cholmod_factorize +
    cholmod_solve, the like-for-like of the GPU's warm step) and (c) its fastest solver with a cached analysis, the native
    block Cholesky (CLinearSolver_UberBlock::Solve_PosDef_Blocky, second call) are reported beside it.  Also returns the
    rel-inf distance of the GPU's solution from the reference's."""
    from oracle import oracle_lib as O
    with tempfile.TemporaryDirectory() as td:
        if O.have_reference():
            path, xp = os.path.join(td, "c3.bin"), os.path.join(td, "x.bin")
            lam.save(path)
            t0 = time.perf_counter()
            r = O.reference_solve(path, "cholmod_auto", xp, reps=budget_reps)
            wall = time.perf_counter() - t0
            ms = float(np.median(r["times_ms"]))
            x_ref = np.fromfile(xp, dtype=np.float64)
            out = {"value": counts["flops"] / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "reference",
                   "ms_per_solve": ms, "sample": f"{budget_reps} x CLinearSolver_CholMod(CHOLMOD_AUTO, AMD)::Solve_PosDef "
                   f"on the same 100k-pose system (median; {wall:.1f} s of CPU incl. load); serial in the reference; built -O3 -march=x86-64-v3 "
                   "(the reference's own flag is -march=native; the binary has to run on another host)",
                   "x_gpu_vs_reference_rel_inf": float(np.abs(x_gpu - x_ref).max() / np.abs(x_ref).max())}
            try:
                ph = subprocess.run([O.REF_HARNESS, "cholmod_phases", path, "auto", "3"], capture_output=True, text=True, timeout=300,
                                    env=O.reference_env())
                reps = json.loads([l for l in ph.stdout.splitlines() if l.startswith("{")][-1])["reps"]
                med = {k: float(np.median([q[k] for q in reps])) for k in ("convert_ms", "analyze_ms", "factorize_ms", "solve_ms")}
                out["cholmod_phases_ms"] = med
                out["numeric_only_ms"] = med["factorize_ms"] + med["solve_ms"]
                xu = os.path.join(td, "x_ub.bin")
                ub = O.reference_solve(path, "uberblock", xu, reps=3)
                out["native_block_solver_ms"] = {"first_call": float(ub["times_ms"][0]), "warm": float(min(ub["times_ms"][1:]))}
                # "the acceptance report must print cond-proxy + inter-oracle spread beside our error"
                xs = [x_ref, np.fromfile(xu, dtype=np.float64)]
                xc = os.path.join(td, "x_cs.bin")
                if O.reference_solve(path, "csparse", xc, reps=1)["ok"]:
                    xs.append(np.fromfile(xc, dtype=np.float64))
                out["inter_oracle_spread"] = max(float(np.abs(a - b).max() / np.abs(b).max()) for a in xs for b in xs if a is not b)
                out["inter_oracle_solvers"] = "CHOLMOD (auto), native block Cholesky, CSparse" if len(xs) == 3 else "CHOLMOD (auto), native block Cholesky"
                out["cond_proxy"] = O.solve_sparse(lam)[2].get("cond_proxy")   # (max / min diagonal of R)^2 <= cond_2, natural order, CPU restatement
            except Exception as e:      # the headline baseline stands without the split
                out["phases_error"] = str(e)[:200]
            return out
    t0 = time.perf_counter()
    ok, _, _ = O.solve_sparse(lam)
    ms = (time.perf_counter() - t0) * 1e3
    return {"value": counts["flops"] / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "ms_per_solve": ms, "sample": "1 x oracle/slampp_oracle.c up-looking block Cholesky, natural order"}


def host_path_leg(make_solver, lam, reps=5):
    """What a caller with *host* arrays pays (cold = ordering + symbolic + upload + factor +
    solve + download, warm = the same with the analysis cached), through slampp_hip_factor_solve: the values move
    through the library's pinned staging in chunks, a few host threads ahead of the DMA engine.  PCIe-inclusive: never
    the headline `value`."""
    solver = make_solver()
    eta = lam.rhs.copy()
    t0 = time.perf_counter()
    ok = solver.Solve_PosDef(lam, eta)
    cold = (time.perf_counter() - t0) * 1e3
    warm, last = [], None
    for _ in range(reps):
        eta = lam.rhs.copy()
        t0 = time.perf_counter()
        ok = solver.Solve_PosDef_Blocky(lam, eta) and ok
        warm.append((time.perf_counter() - t0) * 1e3)
        last = solver.times.as_dict()
    return {"ok": bool(ok), "cold_ms": cold, "warm_host_ms": float(np.median(warm)), "warm_host_ms_min": float(min(warm)),
            "bytes_up": int(8 * (lam.values.shape[0] + lam.n_scalars)), "bytes_down": int(8 * lam.n_scalars),
            "last_call_ms": {k: last[k] for k in ("upload_ms", "factor_ms", "schur_ms", "download_ms", "total_ms")}}


def dropin_leg(lam, reps=5):
    """The C++ boundary itself: oracle/_ref/dropin_driver (the reference's headers + include/slam/LinearSolver_HIP.h, built in
    the build container) builds the system as a CUberBlockMatrix and times the reference's solver class and the HIP one in
    one process: gather of the pooled blocks into pinned staging (OpenMP, chunked, overlapped with the uploads) + solve +
    solution back."""
    drv = os.path.join(ROOT, "oracle", "_ref", "dropin_driver")
    if not os.path.exists(drv):
        return None
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "p.bin")
        lam.save(path)
        try:
            from oracle import oracle_lib as O
            r = subprocess.run([drv, "time", path, str(reps)], capture_output=True, text=True, timeout=900, env=O.reference_env())
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            return json.loads(line[-1]) if line else {"error": (r.stdout + r.stderr)[-300:]}
        except Exception as e:
            return {"error": str(e)[:200]}


def run_c3(args, rank, world, local_rank, dist):
    import torch
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP

    dev = torch.device(f"cuda:{local_rank}")
    lam = synth.pose_chain(n=args.poses)
    solver = CLinearSolver_HIP(device=local_rank)
    t0 = time.perf_counter()
    solver.SymbolicDecomposition_Blocky(lam)
    analyze_ms = (time.perf_counter() - t0) * 1e3
    stats = solver.stats()
    counts = algorithmic_counts_c3(lam, stats)
    vals = torch.from_numpy(lam.values).to(dev)
    rhs0 = torch.from_numpy(lam.rhs).to(dev)
    bufs = [rhs0.clone() for _ in range(args.steps + args.warmup)]
    torch.cuda.synchronize()
    for k in range(args.warmup):
        solver.factor_solve_device_async(vals.data_ptr(), bufs[k].data_ptr())
    if not solver.sync():
        raise SystemExit("warm-up solve failed: not positive definite")
    solver.set_option("profile", 3)      # one event pair in the timed region: around the leaf kernel, the roofline's
    solver.profile(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        solver.factor_solve_device_async(vals.data_ptr(), bufs[k].data_ptr())
    ok = solver.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not ok:
        raise SystemExit("solve failed: not positive definite")
    if rank != 0:
        return None
    # parity guard on the last timed solution: ||Lambda x - eta||_inf / ||eta||_inf
    x = bufs[-1].cpu().numpy()
    resid = float(np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max())
    ms_per_step = dt / args.steps * 1e3
    prof = solver.profile()          # timed region: one event pair, around the leaf kernel (option profile = 3)
    # the split of the table below comes from a few extra, untimed steps: every event pair between two kernels costs
    # microseconds of stream time, and the timed region carries only the one the roofline needs
    solver.set_option("profile", 2)
    solver.profile(reset=True)
    extra = [rhs0.clone() for _ in range(5)]
    for t_ in extra:
        solver.factor_solve_device_async(vals.data_ptr(), t_.data_ptr())
    solver.sync()
    prof_fine = solver.profile()
    solver.set_option("profile", 3)
    solver.profile(reset=True)
    prof = dict(prof_fine, **{k_: v_ for k_, v_ in prof.items() if k_ in ("factor_leaves",)})
    n_stages, n_bottom = stats["n_stages"], stats["n_bottom_stages"]
    kb = per_kernel_bytes_sparse(solver.plan(), n_bottom)
    # the separator stages are one launch each -- the tasks as panels in LDS, next to them the updates the NEXT stage's blocks
    # receive from further down (one half-workgroup per factor block) -- plus one launch of those updates for the first of them
    # (the first panel stage has an update launch of its own only when it sits above wide one-wave-per-column stages; right
    # above the lane-per-task leaves its tasks bring in their updates themselves)
    launches = {"factor_leaves": 1, "factor_wide": max(n_bottom - 1, 1), "factor_upper": max(n_stages - n_bottom, 1) + (1 if n_bottom > 1 else 0),
                "forward": n_stages, "backward": n_stages}
    names = {"factor_leaves": "factor_simt_kernel", "factor_wide": "factor_stage_kernel<D, 1, 8, 32, 48>",
             "factor_upper": "factor_panel_kernel (slices of the elimination tree, one launch per stage)", "forward": "forward_stage_kernel",
             "backward": "backward_stage_kernel (+ backward_simt_kernel for the leaf subtrees where there are many)"}
    needles = {"factor_leaves": "factor_simt_kernel", "factor_wide": ", 1, 8, 32, 48>", "factor_upper": "factor_panel_kernel",
               "forward": "forward_stage_kernel", "backward": "::backward_s"}   # as rocprofv3 spells the kernels (backward_stage_ / backward_simt_)
    traffic, traffic_file = load_traffic("c3")
    kernels = []
    for ph, (cnt, tot_ms) in prof.items():
        if ph not in kb or cnt == 0:
            continue
        per_step_ms = tot_ms / cnt
        kernels.append({"kernel": names[ph], "launches_per_step": launches[ph], "ms_per_step": per_step_ms,
                        "avg_launch_us": per_step_ms / launches[ph] * 1e3,
                        "algorithmic_bytes_per_launch": kb[ph] / launches[ph],
                        "achieved_GBs": kb[ph] / (per_step_ms * 1e-3) / 1e9,
                        "timed_in": "timed region" if ph == "factor_leaves" else "5 extra steps after it",
                        "hbm_traffic_bytes_per_launch": kernel_traffic_mean(traffic, needles[ph]) if isinstance(needles[ph], str) else
                        (lambda parts: (sum(parts) / len(parts)) if all(p is not None for p in parts) else None)(
                            [kernel_traffic(traffic, n_) for n_ in needles[ph]])})
    kernels.sort(key=lambda k: -k["ms_per_step"])
    gpu_ms = sum(k["ms_per_step"] for k in kernels)
    for k in kernels:
        k["share_of_step_time"] = k["ms_per_step"] / gpu_ms if gpu_ms > 0 else None
    # two roofline objects: `roofline` is for the kernel the step spends most of its time in (since round 3 the panel kernel of
    # the separator slices: a chain of launches of 0.1-4 MB each, bound by dependent latency, and priced as what it is);
    # `roofline_leaf_kernel` for the one that moves the step's bytes -- the leaf kernel reads nearly all of Lambda and writes
    # nearly all of L in ONE launch.  `roofline_whole_step` prices the step as a whole.
    def roofline_of(k):
        return {"bound": "hbm", "kernel": k["kernel"], "achieved": k["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": k["achieved_GBs"] / HBM_PEAK_GBS,
                "traffic": k["hbm_traffic_bytes_per_launch"], "traffic_source": traffic_file,
                "traffic_measured_in": "builder's rocprofv3 --pmc run of this command (replayed from the committed file, not measured in this run)",
                "avg_launch_us": k["avg_launch_us"], "launches_per_step": k["launches_per_step"],
                "algorithmic_bytes_per_launch": k["algorithmic_bytes_per_launch"], "share_of_step_time": k["share_of_step_time"],
                "timed_in": k["timed_in"]}
    dom = kernels[0]
    leaf = max(kernels, key=lambda k: k["algorithmic_bytes_per_launch"])
    out = {
        "metric": "Lambda solve GFLOP/s (algorithmic factor+solve flops / wall-clock), 100k-pose SE(3)",
        "value": counts["flops"] * world / (dt / args.steps) / 1e9, "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C3: synthetic {lam.n_bcols}-pose SE(3) chain + loop closures, 6x6 blocks, "
                               f"{lam.n_blocks} upper blocks, n={lam.n_scalars}; numeric factor + 2 substitutions per step "
                               "(symbolic analysis cached, inputs resident in HBM)",
                   "parallelism": "1 GPU" if world == 1 else f"{world} independent replicas (path does not shard)"},
        "solve_residual_rel_inf": resid,
        "algorithmic": {"flops_per_step": counts["flops"], "factor_bytes": counts["factor_bytes"],
                        "solve_bytes": counts["solve_bytes"], "source": counts["source"]},
        "own_ordering": {"l_nnz": stats["l_nnz"], "factor_flops": stats["factor_flops"], "n_stages": n_stages,
                         "n_tasks": stats["n_tasks"], "analyze_ms_cold": analyze_ms},
        "roofline": roofline_of(dom), "roofline_leaf_kernel": roofline_of(leaf),
        "kernels": kernels,
        "roofline_whole_step": {"bound": "hbm", "achieved": (counts["factor_bytes"] + counts["solve_bytes"]) / (dt / args.steps) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": (counts["factor_bytes"] + counts["solve_bytes"]) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                                "launches_per_step": int(sum(k["launches_per_step"] for k in kernels))},
    }
    if args.c3_solve_only:     # (the profiling passes: per-kernel averages of the solve alone)
        return out
    out["assembly"] = assembly_leg(solver, lam, dev)
    if world == 1:
        out["replicas_one_gpu"] = replicas_leg(lam, counts, dev, local_rank, torch)
    if world == 1:
        out["marginals"] = marginals_leg_c3(args, solver, lam, vals, dev, torch)
    if world == 1:
        out["host_path"] = host_path_leg(lambda: CLinearSolver_HIP(device=local_rank), lam)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c3(lam, counts, x)
        out["solve_x_vs_reference_rel_inf"] = out["cpu_baseline"].get("x_gpu_vs_reference_rel_inf")
        out["inter_oracle_spread"] = out["cpu_baseline"].get("inter_oracle_spread")
        out["cond_proxy"] = out["cpu_baseline"].get("cond_proxy")
        out["dropin_cpp"] = dropin_leg(lam)
    return out


def replicas_leg(lam, counts, dev, local_rank, torch, ks=(1, 2, 4, 8), steps=10):
    """SURVEY.md section 8(e), second row: pose graphs do not shard -- "replicas only (multiple independent problems /
    damping values per GPU)".  K independent solves of the SAME structure on ONE device, each on a handle (and HIP stream)
    of its own: what an LM loop trying K damping values at once, or K robots' graphs, would enqueue
    (/root/reference/include/slam/NonlinearSolver_Lambda_LM.h:967-1001, 1660-1676: the reference re-damps and re-solves one after
    the other).  A single C3 solve is a chain of 12 dependent launches that fills a fraction of the chip; K chains side by
    side is the throughput the device has for this workload.  Reported: aggregate GFLOP/s and the whole-step HBM fraction
    on SURVEY 8d's algorithmic bytes, per K."""
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
    k_max = max(ks)
    solvers = [CLinearSolver_HIP(device=local_rank) for _ in range(k_max)]
    for s_ in solvers:
        s_.SymbolicDecomposition_Blocky(lam)
    vals = [torch.from_numpy(lam.values).to(dev) for _ in range(k_max)]     # K value sets (damped copies, in the LM reading)
    torch.cuda.synchronize()
    for k_, v_ in enumerate(vals):
        solvers[k_].apply_damping_device_async(v_.data_ptr(), 1e-3 * k_, 0, lam.n_bcols)
        solvers[k_].sync()
    rhs0 = torch.from_numpy(lam.rhs).to(dev)
    out = {"workload": f"K concurrent solves of the C3 structure on one device, a handle and a stream each; K value sets (damping 1e-3 k)", "by_k": {}}
    bytes_step = counts["factor_bytes"] + counts["solve_bytes"]
    for K in ks:
        bufs = [[rhs0.clone() for _ in range(steps + 1)] for _ in range(K)]
        torch.cuda.synchronize()
        for k_ in range(K):
            solvers[k_].factor_solve_device_async(vals[k_].data_ptr(), bufs[k_][0].data_ptr())
        assert all(solvers[k_].sync() for k_ in range(K))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            for k_ in range(K):
                solvers[k_].factor_solve_device_async(vals[k_].data_ptr(), bufs[k_][i].data_ptr())
        ok = all(solvers[k_].sync() for k_ in range(K))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        x = bufs[K - 1][-1].cpu().numpy()
        out["by_k"][str(K)] = {"ok": bool(ok), "ms_per_round": dt * 1e3, "solves_per_s": K / dt,
                                "GFLOP/s": K * counts["flops"] / dt / 1e9, "hbm_frac_whole_step": K * bytes_step / dt / 1e9 / HBM_PEAK_GBS,
                                "finite": bool(np.isfinite(x).all())}
    out["speedup_k8_vs_k1"] = out["by_k"][str(k_max)]["solves_per_s"] / out["by_k"]["1"]["solves_per_s"] if "1" in out["by_k"] else None
    # ... and the same K value sets through ONE handle in ONE pass of launches (slampp_hip_factor_solve_batch_device_async):
    # the chain of dependent launches as long as for one system, every launch K times as wide
    del solvers[1:]
    solver = solvers[0]
    n_v, n_s = lam.values.shape[0] + lam.values.shape[0] % 2, lam.n_scalars + lam.n_scalars % 2
    vb = torch.zeros(k_max * n_v, dtype=torch.float64, device=dev)
    for k_ in range(k_max):
        vb[k_ * n_v:k_ * n_v + lam.values.shape[0]] = vals[k_]
    del vals
    out["batched"] = {}
    for K in ks:
        rb = [torch.zeros(K * n_s, dtype=torch.float64, device=dev) for _ in range(steps + 1)]
        for r_ in rb:
            for k_ in range(K):
                r_[k_ * n_s:k_ * n_s + lam.n_scalars] = rhs0
        torch.cuda.synchronize()
        solver.factor_solve_batch_device_async(K, vb.data_ptr(), n_v, rb[0].data_ptr(), n_s)
        ok = all(solver.sync_batch(K))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            solver.factor_solve_batch_device_async(K, vb.data_ptr(), n_v, rb[i].data_ptr(), n_s)
        ok = ok and all(solver.sync_batch(K))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # parity guard: the last member's solution against the residual of ITS system (damped by 1e-3 (K - 1))
        x = rb[-1][(K - 1) * n_s:(K - 1) * n_s + lam.n_scalars].cpu().numpy()
        resid = float(np.abs(lam.to_scipy() @ x + 1e-3 * (K - 1) * x - lam.rhs).max() / np.abs(lam.rhs).max())
        out["batched"][str(K)] = {"ok": bool(ok), "ms_per_round": dt * 1e3, "solves_per_s": K / dt, "GFLOP/s": K * counts["flops"] / dt / 1e9,
                                  "hbm_frac_whole_step": K * bytes_step / dt / 1e9 / HBM_PEAK_GBS, "last_member_resid": resid}
    return out


def marginals_leg_c3(args, solver, lam, vals, dev, torch):
    """Outside the timed region: the block diagonal of the covariance Lambda^-1 of the same pose graph (numeric
    factorization + sparse inverse subset on the factor's pattern + extraction), next to the reference's recipe for it
    (ordering, CholeskyOf_FBS, CMarginals::Calculate_DenseMarginals_Recurrent_FBS) on the host."""
    n, d = lam.n_bcols, int(lam.cumsum[1])
    cov = torch.empty(n * d * d, dtype=torch.float64, device=dev)
    lib, h = solver._lib, solver._h
    solver._check(lib.slampp_hip_marginals_device_async(h, vals.data_ptr(), cov.data_ptr()))
    if not solver.sync():
        return None
    solver.profile(reset=True)
    reps = 10
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        solver._check(lib.slampp_hip_marginals_device_async(h, vals.data_ptr(), cov.data_ptr()))
    ok = solver.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    prof = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items() if v[0]}
    solver.profile(reset=True)
    c_np = cov.cpu().numpy().reshape(n, d, d)
    err = 0.0
    for c in (n // 3, n - 1):   # column j of the covariance is the solution of Lambda x = e_j
        e = np.zeros(lam.n_scalars)
        e[d * c] = 1.0
        if not solver.Solve_PosDef_Blocky(lam, e):
            return None
        err = max(err, float(np.abs(e[d * c:d * c + d] - c_np[c][:, 0]).max() / np.abs(c_np[c][:, 0]).max()))
    out = {"workload": f"block diagonal of Lambda^-1: {n} blocks {d}x{d}", "ok": bool(ok), "ms_per_call": ms, "phases_ms": prof,
           "column_check_rel_inf": err}
    if not args.no_cpu_baseline:
        from oracle import oracle_lib as O
        if O.have_reference():
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "c3.bin")
                lam.save(path)
                t0 = time.perf_counter()
                r = subprocess.run([O.REF_HARNESS, "sparse_marginals", path, os.path.join(td, "m")], capture_output=True, text=True,
                                   timeout=900, env=O.reference_env())
                wall = time.perf_counter() - t0
            if '"ok": true' in r.stdout:
                out["cpu_baseline"] = {"value": wall * 1e3, "unit": "ms", "cores": 1, "kind": "reference",
                                       "sample": "block ordering, CholeskyOf_FBS and CMarginals::Calculate_DenseMarginals_Recurrent_FBS"
                                                 "(.., mpart_Diagonal) on the same system, incl. load"}
    return out


def assembly_leg(solver, lam, dev, reps=20, rd=None, column_vertex_first=False):
    """Outside the timed region: Lambda and eta of the same graph assembled on the device from synthetic per-edge
    Jacobians, written where the solver reads them.  HBM-bound: bytes in (J0, J1, Sigma^-1,
    error, weight per edge) + bytes out (Lambda values, eta) over the HIP-event time of the two kernels."""
    import torch
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLambdaAssembly_HIP
    col = np.repeat(np.arange(lam.n_bcols), np.diff(lam.bcol_ptr))
    off = lam.brow_idx != col
    v0, v1 = lam.brow_idx[off].astype(np.int64), col[off].astype(np.int64)
    if column_vertex_first:   # BA: vertex 0 of a projection edge is the landmark (EDGE_P2MC xyz_id cam_id), the later block column
        v0, v1 = v1, v0
    dims = np.diff(lam.cumsum)
    d = int(dims[0]) if rd is None else int(rd)
    es = synth.random_edge_set(dims, v0, v1, rd=d, seed=3)
    asm = CLambdaAssembly_HIP(solver, lam, v0, v1, d)
    bufs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (es.J0, es.J1, es.sigma_inv, es.err, es.weight)]
    values = torch.empty(lam.values.shape[0], dtype=torch.float64, device=dev)
    eta = torch.empty(lam.n_scalars, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    args = [t.data_ptr() for t in bufs] + [values.data_ptr(), eta.data_ptr(), es.unary_vertex, es.unary_factor, es.unary_error]
    asm.Refresh_Lambda_device(*args)
    solver.sync()
    solver.set_option("profile", 1)        # (level 3 keeps only the solve's own kernels)
    solver.profile(reset=True)
    for _ in range(reps):
        asm.Refresh_Lambda_device(*args)
    solver.sync()
    cnt, ms = solver.profile().get("assemble", (0, 0.0))
    ok = solver.factor_solve_device(values.data_ptr(), eta.data_ptr())     # the assembled system, solved where it lies
    n_bytes = 8 * (sum(int(np.prod(t.shape)) for t in bufs) + values.numel() + eta.numel())
    us = ms / max(cnt, 1) * 1e3
    return {"n_edges": int(v0.shape[0]), "us_per_assembly": us, "algorithmic_bytes": n_bytes,
            "achieved_GBs": n_bytes / (us * 1e-6) / 1e9 if us > 0 else None, "peak_GBs": HBM_PEAK_GBS,
            "assembled_system_solved": bool(ok)}


def run_small_configs(args, local_rank):
    """BASELINE.json configs[0] and configs[1] (parity-test cases, not the benchmark workload): the Manhattan3500 SE(2)
    and Sphere2500 SE(3) look-alikes, warm numeric factor + solve on the GPU next to the reference's CHOLMOD on the
    host.  Reported as extra objects of the JSON line; N = 1 only."""
    import torch
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
    from oracle import oracle_lib as O
    dev = torch.device(f"cuda:{local_rank}")
    out = {}
    which = [k_ for k_ in args.small_configs.split(",") if k_]
    for key, name, make in (("C1", "Manhattan3500 SE(2) look-alike, 3x3 blocks", lambda: synth.manhattan(3500)),
                            ("C2", "Sphere2500 SE(3) look-alike, 6x6 blocks", lambda: synth.sphere(50, 50))):
        if key not in which:
            continue
        lam = make()
        solver = CLinearSolver_HIP(device=local_rank)
        t0 = time.perf_counter()
        solver.SymbolicDecomposition_Blocky(lam)
        analyze_ms = (time.perf_counter() - t0) * 1e3
        vals = torch.from_numpy(lam.values).to(dev)
        reps = 20
        bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(reps + 1)]
        torch.cuda.synchronize()
        if not solver.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr()):
            raise SystemExit(f"{key}: not positive definite")
        t0 = time.perf_counter()
        for b in bufs[1:]:
            solver.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
        solver.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        x = bufs[-1].cpu().numpy()
        st = solver.stats()
        rec = {"workload": f"{name}, n={lam.n_scalars}", "ms_per_solve": ms, "analyze_ms_cold": analyze_ms,
               "dense_top_dim": st["schur_dim"], "n_stages": st["n_stages"],
               "solve_residual_rel_inf": float(np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max())}
        # the split of the step (five extra solves with every phase bracketed by events) and its roofline: the big separators
        # of a 2-D-like graph are factored as one dense matrix on the matrix cores (the "dense top": the flops of its columns
        # under our ordering, sum of squared column counts, against the fp64 MFMA peak); the block-by-block part below it is
        # a chain of small launches and is priced against HBM on its SURVEY 8d bytes
        solver.set_option("profile", 2)
        solver.profile(reset=True)
        extra = [torch.from_numpy(lam.rhs).to(dev) for _ in range(5)]
        for t_ in extra:
            solver.factor_solve_device_async(vals.data_ptr(), t_.data_ptr())
        solver.sync()
        prof = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items()}
        rec["phases_ms"] = prof
        plan = solver.plan()
        dim = plan["dim"].astype(np.int64)
        lptr, lrow, dpos = plan["lptr"], plan["lrow"].astype(np.int64), plan["dense_pos"]
        col_of = np.repeat(np.arange(len(dim)), np.diff(lptr))
        below = np.bincount(col_of, weights=dim[lrow], minlength=len(dim)) - dim      # scalar rows below the diagonal block
        t_ = np.arange(1, dim.max() + 1)
        col_flops = np.array([np.sum((below[j] + t_[:dim[j]]) ** 2) for j in range(len(dim))], dtype=np.float64)
        col_lnz = dim * (dim + 1) // 2 + dim * below
        top = dpos >= 0
        if "dense_chol" in prof and top.any():
            tf = float(col_flops[top].sum()) / (prof["dense_chol"] * 1e-3) / 1e12
            rec["roofline"] = {"bound": "mfma", "kernel": "dense top: tile-scheduled Cholesky on the matrix cores (tile_potrf / tile_trsm / tile_update, "
                               "or potrf_diag / trsm / syrk)", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / F64_MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_flops": float(col_flops[top].sum()),
                               "ms": prof["dense_chol"], "dense_top_dim": int(st["schur_dim"])}
        sparse_ms = sum(prof.get(k_, 0.0) for k_ in ("factor_leaves", "factor_wide", "factor_upper", "factor_rest"))
        if sparse_ms > 0:
            nbytes = 8.0 * (float(st["nnz_upper"]) + float(col_lnz[~top].sum()))
            rec["roofline_block_part"] = {"bound": "hbm", "kernel": "block-by-block elimination below the dense top (leaf subtrees + separator panels)",
                                          "achieved": nbytes / (sparse_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": nbytes / (sparse_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                          "algorithmic_bytes": nbytes, "ms": sparse_ms}
        if not args.no_cpu_baseline and O.have_reference():
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "p.bin")
                lam.save(path)
                r = O.reference_solve(path, "cholmod_auto", "-", reps=5)
            rec["reference_cholmod_ms"] = float(np.median(r["times_ms"]))
            # ... and the reference's fastest solver with its analysis cached (CLinearSolver_UberBlock::Solve_PosDef_Blocky,
            # calls after the first): the like-for-like of the GPU's warm step
            try:
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, "p.bin")
                    lam.save(path)
                    ub = O.reference_solve(path, "uberblock", "-", reps=5)
                rec["reference_native_block_solver_ms"] = {"first_call": float(ub["times_ms"][0]), "warm": float(min(ub["times_ms"][1:]))}
                rec["speedup_vs_reference_native_warm"] = float(min(ub["times_ms"][1:])) / ms
            except Exception as e:
                rec["reference_native_error"] = str(e)[:200]
        out[key] = rec
    return out


class _DevPtr:
    """Lets torch wrap a raw device pointer (the solver's [S | r] buffer) without copying."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def make_allreduce(dist, torch, dev):
    """slampp_hip_allreduce_fn over torch.distributed (backend nccl = RCCL over xGMI): sums the
    partial reduced camera systems in place, ordered on the solver's own HIP stream."""
    cache = {}

    def fn(ptr, count, stream):
        t = cache.get((ptr, count))
        if t is None:
            t = cache[(ptr, count)] = torch.as_tensor(_DevPtr(ptr, count), device=dev)
        with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
            dist.all_reduce(t)
        return 0
    return fn


def cpu_baseline_ba(lam, flops, x_gpu):
    from oracle import oracle_lib as O
    if not O.have_reference():
        return None
    with tempfile.TemporaryDirectory() as td:
        path, xp = os.path.join(td, "ba.bin"), os.path.join(td, "x.bin")
        lam.save(path)
        t0 = time.perf_counter()
        r = O.reference_solve(path, "schur", xp, reps=2, timeout=900)
        wall = time.perf_counter() - t0
        x_ref = np.fromfile(xp, dtype=np.float64)
    ms = float(r["times_ms"][-1])
    return {"value": flops / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "cores": host_cores(), "kind": "reference",
            "ms_per_solve": ms, "ms_first_call": float(r["times_ms"][0]),
            "x_gpu_vs_reference_rel_inf": float(np.abs(x_gpu - x_ref).max() / np.abs(x_ref).max()),
            "sample": f"2 x CLinearSolver_Schur<CholMod>::Solve_PosDef[_Blocky] on the same system (second call, ordering reused; "
                      f"{wall:.1f} s of CPU incl. load); OpenMP only in the block-diagonal inverse and one SpMV, dense LLT serial"}


def incremental_leg(lam, dev, local_rank, torch, share=0.01, reps=5, always=False):
    """Outside the timed region: option schur_incremental.  After a relinearization that moved `share` of the landmarks the
    reduced camera system is updated from the previous one (the reference's dog-leg solver does that from Omega = delta
    Lambda, NonlinearSolver_Lambda_DL.h:2301-) instead of rebuilt: the same solve both ways, same values."""
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    rng = np.random.default_rng(3)
    nc, n_pts = lam.n_matrix_cut, lam.n_bcols - lam.n_matrix_cut
    points = np.sort(rng.choice(n_pts, size=max(int(share * n_pts), 1), replace=False))
    off = lam.block_value_offsets()
    vals2 = lam.values.copy()
    for p_ in points:                                             # the moved landmarks: more curvature, scaled projections
        k0, k1 = int(lam.bcol_ptr[nc + p_]), int(lam.bcol_ptr[nc + p_ + 1])
        vals2[off[k0]:off[k1 - 1]] *= 0.9
        vals2[off[k1 - 1]:off[k1]] += 0.5 * np.eye(3).ravel()
    solver = CLinearSolver_Schur_HIP(device=local_rank, schur_incremental=2 if always else 1)   # 2: use the list however long it is
    solver.SymbolicDecomposition_Blocky(lam)
    v1, v2 = torch.from_numpy(lam.values).to(dev), torch.from_numpy(vals2).to(dev)
    rhs = torch.from_numpy(lam.rhs).to(dev)
    out = {}
    for name, use_list in (("full_rebuild_ms", False), ("update_ms", True)):
        ms, xs = [], None
        for _ in range(reps):
            b1, b2 = rhs.clone(), rhs.clone()
            solver.factor_solve_device(v1.data_ptr(), b1.data_ptr())           # the system before the relinearization
            if use_list:
                solver.Set_Changed_Landmarks(points)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ok = solver.factor_solve_device(v2.data_ptr(), b2.data_ptr())
            ms.append((time.perf_counter() - t0) * 1e3)
            xs = b2.cpu().numpy()
        out[name] = float(np.median(ms))
        out[name.replace("_ms", "_x")] = xs
    x_full, x_upd = out.pop("full_rebuild_x"), out.pop("update_x")
    out["update_vs_full_rel_inf"] = float(np.abs(x_upd - x_full).max() / np.abs(x_full).max())
    out["changed_landmarks"] = int(len(points))
    out["ok"] = bool(ok)
    out["note"] = ("option schur_incremental = 1: the list is used when it is the shorter way (up to 1/32 of the landmarks "
                   "with the landmark-major assembly); a longer list is answered with the full rebuild")
    return out


def marginals_leg(args, solver, lam, vals, dev, torch):
    """Block diagonal of the covariance on the bench's BA system, after the timed solves:
    the reduced system assembled and factored as for a solve, the blocks of its inverse the landmarks need taken from
    a sparse inverse subset on the factor's pattern -- or, `dense_inverse`, S inverted on the matrix cores (2 n^3 / 3
    flops) -- then gathered per landmark.  The reference's CSchurComplement_Marginals is run beside it on a bounded
    sample (--no-cpu-baseline skips it)."""
    nc, n_pts = lam.n_matrix_cut, lam.n_bcols - lam.n_matrix_cut
    cams = torch.empty(nc * 36, dtype=torch.float64, device=dev)
    pts = torch.empty(n_pts * 9, dtype=torch.float64, device=dev)
    def timed():
        solver.schur_marginals_device_async(vals.data_ptr(), cams.data_ptr(), pts.data_ptr())
        if not solver.sync():
            return None
        solver.profile(reset=True)
        reps = 3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            solver.schur_marginals_device_async(vals.data_ptr(), cams.data_ptr(), pts.data_ptr())
        ok_ = solver.sync()
        ms_ = (time.perf_counter() - t0) / reps * 1e3
        prof_ = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items() if k_.startswith("marginals")}
        solver.profile(reset=True)
        return ok_, ms_, prof_

    # first the way the library picks (the sparse inverse subset when the solves factor S by the sparse block path), then
    # with S inverted densely on the matrix cores: that one has the MFMA roofline
    n = 6.0 * nc
    solver.set_option("marginals_dense", 1)
    r_dense = timed()
    solver.set_option("marginals_dense", 0)
    r = timed()
    if r is None or r_dense is None:
        return None
    ok, ms, prof = r
    tf = 2.0 * n ** 3 / 3.0 / (r_dense[2]["marginals_inverse"] * 1e-3) / 1e12
    # a sampled check against the definition: column j of the covariance is the solution of Lambda x = e_j
    c_np, p_np = cams.cpu().numpy().reshape(nc, 6, 6), pts.cpu().numpy().reshape(n_pts, 3, 3)
    err = 0.0
    for (idx, blk, d, base) in ((nc // 3, c_np, 6, 0), (n_pts // 2, p_np, 3, 6 * nc)):
        e = np.zeros(lam.n_scalars)
        e[base + d * idx] = 1.0
        if not solver.Solve_PosDef_Blocky(lam, e):
            return None
        ref = e[base + d * idx: base + d * idx + d]
        err = max(err, float(np.abs(ref - blk[idx][:, 0]).max() / np.abs(ref).max()))
    out = {"workload": f"block diagonal of Lambda^-1: {nc} camera blocks 6x6 + {n_pts} landmark blocks 3x3", "ok": bool(ok),
           "ms_per_call": ms, "phases_ms": prof, "column_check_rel_inf": err,
           "dense_inverse": {"ok": bool(r_dense[0]), "ms_per_call": r_dense[1], "phases_ms": r_dense[2],
                             "roofline": {"bound": "mfma",
                                          "kernel": "inverse_level_kernel + inverse_lauum_kernel (inverse of S from its factor)",
                                          "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": tf / F64_MFMA_PEAK_TFLOPS, "traffic": None, "flops": 2.0 * n ** 3 / 3.0,
                                          "ms": r_dense[2]["marginals_inverse"]}}}
    if not args.no_cpu_baseline:
        from oracle import oracle_lib as O
        if O.have_reference():
            import subprocess
            sample = dataclasses_replace_points(lam, min(n_pts, 100_000))
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "ba.bin")
                sample.save(path)
                t0 = time.perf_counter()
                r = subprocess.run([O.REF_HARNESS, "schur_marginals", path, os.path.join(td, "m")], capture_output=True, text=True,
                                   timeout=900, env=O.reference_env())
                wall = time.perf_counter() - t0
            if '"ok": true' in r.stdout:
                out["cpu_baseline"] = {"value": wall * 1e3, "unit": "ms", "cores": host_cores(), "kind": "reference",
                                       "sample": f"CSchurComplement_Marginals::Schur_Marginals on the first "
                                                 f"{sample.n_bcols - nc} landmarks of the same system (all {nc} cameras), with the "
                                                 f"Schur complement and its Cholesky factor it needs, incl. load; OpenMP"}
    return out


def dataclasses_replace_points(lam, n_keep):
    """The same BA system cut down to its first n_keep landmarks (block columns are stored landmark by landmark)."""
    from slam_plus_plus_amd.synth import BlockSystem
    nc = lam.n_matrix_cut
    n = nc + n_keep
    nb = int(lam.bcol_ptr[n])
    off = lam.block_value_offsets()
    return BlockSystem(lam.cumsum[:n + 1].copy(), lam.bcol_ptr[:n + 1].copy(), lam.brow_idx[:nb].copy(),
                       lam.values[:off[nb]].copy(), lam.rhs[:int(lam.cumsum[n])].copy(), nc)


def shared_system(tag, make, rank, world, dist):
    """One fixed system for all ranks: rank 0 builds it (the generator is a minute of numpy at 2 M landmarks), the others
    map its arrays from /dev/shm and copy out only what their shard needs."""
    if world == 1:
        return make()
    from slam_plus_plus_amd.synth import BlockSystem
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(),
                        f"slampp_bench_{os.environ.get('MASTER_PORT', '0')}_{tag}")
    names = ("cumsum", "bcol_ptr", "brow_idx", "values", "rhs")
    lam = None
    if rank == 0:
        lam = make()
        for n_ in names:
            np.save(f"{base}_{n_}.npy", getattr(lam, n_))
        with open(f"{base}_cut.txt", "w") as f:
            f.write(str(int(lam.n_matrix_cut)))
    dist.barrier()
    if rank != 0:
        arrays = [np.load(f"{base}_{n_}.npy", mmap_mode="r") for n_ in names]
        lam = BlockSystem(*arrays, int(open(f"{base}_cut.txt").read()), tag)
    dist.barrier()
    if rank == 0:    # (the others hold their mappings open: the names can go)
        for n_ in names:
            os.unlink(f"{base}_{n_}.npy")
        os.unlink(f"{base}_cut.txt")
    return lam


def run_ba(args, rank, world, local_rank, dist, schur_sparse=-1, mode="band", extras=True, cams=None, points=None, label=None):
    """One BA system of `cams` cameras x `points` landmarks in total (default: C4, --ba-cams x --ba-points), solved
    through the Schur complement.  N > 1: the SAME system cut into N landmark shards, one per rank (strong scaling), the
    partial reduced camera systems summed by one RCCL all-reduce per step.
    schur_sparse: -1 = the library decides how to factor the reduced camera system (sparse block path when under 15 %
    of its blocks are nonzero), 0 = force the dense MFMA factorization."""
    import torch
    from slam_plus_plus_amd import synth, sharding
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP

    dev = torch.device(f"cuda:{local_rank}")
    k = 4
    n_cams, n_points_total = cams or args.ba_cams, points or args.ba_points
    lam_full = shared_system(f"ba_{n_cams}x{n_points_total}_{mode}",
                             lambda: synth.ba(n_cams, n_points_total, k=k, mode=mode, seed=777), rank, world, dist)
    if world > 1 and rank == 0 and label == "C5":
        _KEEP["C5"] = lam_full      # the device group leg solves the same system after the ranks are done
    if world > 1:
        lam, own = sharding.landmark_shard(lam_full, rank, world)   # A and eta_x as 1 / world on every rank: the sum is the system
    else:
        lam, own = lam_full, slice(int(lam_full.cumsum[lam_full.n_matrix_cut]), lam_full.n_scalars)
    solver = CLinearSolver_Schur_HIP(device=local_rank, schur_sparse=schur_sparse)
    t0 = time.perf_counter()
    solver.SymbolicDecomposition_Blocky(lam)
    analyze_ms = (time.perf_counter() - t0) * 1e3
    if dist is not None:
        solver.set_option("shard_rank", rank)
        solver.set_option("shard_world", world)
        solver.set_allreduce(make_allreduce(dist, torch, dev))
    st = solver.stats()
    vals = torch.from_numpy(lam.values).to(dev)
    steps, warmup = args.ba_steps, 1
    bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(steps + warmup)]
    torch.cuda.synchronize()
    for i in range(warmup):
        solver.factor_solve_device_async(vals.data_ptr(), bufs[i].data_ptr())
    if not solver.sync():
        raise SystemExit("BA warm-up solve failed")
    solver.set_option("profile", 3)      # event pairs only around the kernels the rooflines are about
    solver.profile(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        solver.factor_solve_device_async(vals.data_ptr(), bufs[i].data_ptr())
    ok = solver.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not ok:
        raise SystemExit("BA solve failed: not positive definite")
    prof = {k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items()}   # timed region: the roofline kernels' phases
    # the other phases from three extra, untimed steps with every phase bracketed (an event pair costs microseconds);
    # every rank runs them: the solve holds a collective
    solver.set_option("profile", 1)
    solver.profile(reset=True)
    extra_rhs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(3)]
    torch.cuda.synchronize()
    for t_ in extra_rhs:
        solver.factor_solve_device_async(vals.data_ptr(), t_.data_ptr())
    solver.sync()
    del extra_rhs
    prof = dict({k_: v[1] / max(v[0], 1) for k_, v in solver.profile().items()}, **prof)
    solver.profile(reset=True)
    red = solver.reduced_stats()
    x_local = bufs[-1].cpu().numpy()
    totals = np.array([st["n_points"], st["n_observations"], st["n_update_pairs"]], dtype=np.float64)
    phases_by_rank, parity = [prof], None
    if world > 1:
        # the whole system's counts, every rank's phases, and the parity guard of the sharded solution: the residual of the
        # FULL system, assembled from every rank's landmarks (rank 0 holds the full matrix: it built it)
        t = torch.from_numpy(totals).to(dev)
        dist.all_reduce(t)
        totals = t.cpu().numpy()
        phases_by_rank = [None] * world
        dist.all_gather_object(phases_by_rank, prof)
        pieces = [None] * world
        dist.all_gather_object(pieces, (own.start, own.stop, x_local[int(lam_full.cumsum[lam_full.n_matrix_cut]):]))
        if rank == 0:
            x_full = np.empty(lam_full.n_scalars)
            x_full[:int(lam_full.cumsum[lam_full.n_matrix_cut])] = x_local[:int(lam_full.cumsum[lam_full.n_matrix_cut])]
            for a_, b_, piece in pieces:
                x_full[a_:b_] = piece
            parity = float(np.abs(lam_full.to_scipy() @ x_full - lam_full.rhs).max() / np.abs(lam_full.rhs).max())
    if rank != 0:
        return None
    ms = dt / steps * 1e3
    n_values_total, n_scalars_total = int(lam_full.values.shape[0]), int(lam_full.n_scalars)
    dc_ = int(lam_full.cumsum[1] - lam_full.cumsum[0])
    n_pts, n_obs, n_pairs, N = int(totals[0]), int(totals[1]), int(totals[2]), st["schur_dim"]
    # per point with k observations 58 + 108 k + 216 k (k + 1) / 2 flops for the Schur
    # products, 2 flops per stored scalar of U for each of the 3 SpMV passes, n^3/3 + ... for the dense factor
    schur_flops = n_pts * 58.0 + 108.0 * n_obs + 216.0 * n_pairs + 3 * 2.0 * 18 * n_obs   # the whole system's (all shards)
    dense_flops = st["factor_flops"] + st["solve_flops"]
    b_dense = "dense_chol" in prof
    # the factorization of the reduced system is redundant on every rank: counted once -- n^3/3 for the dense one, the inner
    # plan's own count (sum of squared column counts under our ordering + 4 nnz(L)) for the sparse one
    reduced_flops = dense_flops if b_dense else red["factor_flops"] + red["solve_flops"]
    flops = schur_flops + reduced_flops
    out = {
        "workload": f"{label or ('C4' if (n_cams, n_points_total) == (1000, 500_000) else 'BA')}: BA {n_cams} cams x {n_points_total} points"
                    f"{'' if world == 1 else f' as {world} landmark shards (one fixed system: strong scaling)'}, "
                    f"{'2..30 (mean 5.3)' if mode == 'venice' else k} obs/point, {mode} visibility; Schur complement + "
                    f"{'dense (MFMA)' if b_dense else 'sparse block'} factorization of the reduced system, per step",
        "reduced_system": "dense" if b_dense else "sparse",
        "ms_per_step": ms, "points_per_s": n_pts / (dt / steps), "GFLOP/s": flops / (dt / steps) / 1e9,
        "n_gpus": world, "steps": steps, "schur_dim": N, "n_observations": n_obs, "analyze_ms_cold": analyze_ms,
        "phases_ms": prof, "n_camera_pair_blocks": st["l_blocks"], "n_contributions": n_pairs,
        "algorithmic_flops": {"schur": schur_flops, "reduced_system": reduced_flops},
        # what a caller's host arrays hold and what the ranks exchange (scaling_model): packed values, scalars, and the
        # nonzero camera-pair blocks of S + the reduced right-hand side (the whole lower triangle when S is dense)
        "n_values": n_values_total, "n_scalars": n_scalars_total,
        "n_exchange_doubles": (N * (N + 1) // 2 if b_dense else int(st["l_blocks"]) * dc_ * dc_) + N,
    }
    if world > 1:
        out["phases_ms_by_rank"] = phases_by_rank
        out["solve_residual_rel_inf"] = parity
        out["exchange"] = "torch.distributed all_reduce (RCCL) of the packed blocks of S + the reduced right-hand side, on the solver's stream"
    # the committed counter passes are of the C4 legs on one GPU: other sizes and shards have no measured traffic
    b_c4_single = world == 1 and (cams or args.ba_cams) == 1000 and (points or args.ba_points) == 500_000
    traffic, traffic_file = load_traffic("ba_" + mode) if b_c4_single else ({}, None)
    if b_c4_single and not traffic and mode == "band":
        traffic, traffic_file = load_traffic("ba")   # (rounds 1 and 2 named the band leg's file so)
    if "dense_chol" in prof:
        tf = st["factor_flops"] / (prof["dense_chol"] * 1e-3) / 1e12
        n_panels = (N + 1 + 63) // 64
        tr = None
        if traffic and all(any(k_ in n for n in traffic) for k_ in ("potrf_diag_kernel", "trsm_kernel")):
            # HBM bytes of the whole factorization = sum over its kernels of launches x bytes per launch (the updates ride in
            # the potrf_diag launches; syrk_kernel launches of its own only in the files of rounds 1 and 2)
            present = [k_ for k_ in ("potrf_diag_kernel", "trsm_kernel", "syrk_kernel") if any(k_ in n for n in traffic)]
            tr = sum((kernel_traffic(traffic, k_) or 0.0) * traffic[[n for n in traffic if k_ in n][0]]["launches"]
                     for k_ in present) / \
                max(traffic[[n for n in traffic if "potrf_diag_kernel" in n][0]]["launches"] / n_panels, 1)
        out["roofline"] = {"bound": "mfma", "kernel": "dense_cholesky (potrf_diag_kernel with the updates riding + trsm_kernel, one factorization)",
                           "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TFLOPS,
                           "sustained_matrix_rate_measured": F64_MFMA_SUSTAINED_TFLOPS, "traffic": tr, "traffic_source": traffic_file, "flops_per_factorization": st["factor_flops"],
                           "ms_per_factorization": prof["dense_chol"]}
    if "schur_tiles" in prof:
        # landmark-major assembly (schur_tiles.hip): every landmark's column of Lambda is read once -- 144 B per observation,
        # 72 B of C and 24 B of l per landmark --, C^-1 written (72 B), every block of S read and written once
        # (per launch = this rank's shard)
        nbytes = 144.0 * st["n_observations"] + (72.0 + 24.0 + 72.0) * st["n_points"] + 2 * 8.0 * 36 * st["l_blocks"]
        gb = nbytes / (prof["schur_tiles"] * 1e-3) / 1e9
        tr = None
        if traffic:
            parts = [kernel_traffic_sum(traffic, k_) for k_ in ("schur_run_kernel", "schur_tile_kernel", "schur_tile_reduce_kernel")]
            tr = sum(p for p in parts if p) or None
        out["roofline_schur_assembly"] = {
            "bound": "hbm", "kernel": "schur_run_kernel (+ schur_tile_kernel for landmarks outside runs, + schur_tile_reduce_kernel): "
                                      "S and r assembled landmark by landmark, one launch of each per step",
            "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS, "traffic": tr,
            "traffic_source": traffic_file,
            "traffic_measured_in": "builder's rocprofv3 --pmc run (replayed from the committed file)" if traffic_file else None,
            "algorithmic_bytes_per_step": nbytes, "ms_per_step": prof["schur_tiles"]}
        # ... and over EVERY assembly kernel of the step (run / tile / reduce kernels + what the contribution lists of the
        # landmarks in no run or tile cost: C^-1, W, gather, right-hand side), on the same algorithmic bytes: what the
        # reference's steps a13-a18 (LinearSolver_Schur.h:1743-1767) cost here as a whole
        asm_ms = sum(prof.get(k_, 0.0) for k_ in ("schur_tiles", "schur_gather", "schur_points", "schur_rhs", "schur_init"))
        tr_all = None
        if traffic:
            parts = [kernel_traffic_sum(traffic, k_) for k_ in ("schur_run_kernel", "schur_tile_kernel", "schur_tile_reduce_kernel", "schur_gather_S_kernel",
                                                                "schur_obs_W_kernel", "schur_rhs_kernel", "schur_point_inverse_kernel", "schur_scatter_A_kernel",
                                                                "schur_wide_kernel")]
            tr_all = sum(p for p in parts if p) or None
        out["roofline_schur_assembly_all"] = {
            "bound": "hbm", "kernel": "every kernel of the assembly of S and r (landmark-major runs / tiles / reduction + contribution lists)",
            "achieved": nbytes / (asm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (asm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": tr_all, "traffic_source": traffic_file, "algorithmic_bytes_per_step": nbytes, "ms_per_step": asm_ms}
        if not b_dense:
            out["roofline"] = out["roofline_schur_assembly_all"]
    if "schur_gather" in prof and prof["schur_gather"] > 0.02 and "schur_tiles" not in prof:
        gb = (288.0 * st["n_update_pairs"] + 2 * 8.0 * 36 * st["l_blocks"]) / (prof["schur_gather"] * 1e-3) / 1e9
        out["roofline_schur_gather"] = {"bound": "hbm", "kernel": "schur_gather_S_kernel", "achieved": gb, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                                        "traffic": kernel_traffic(traffic, "3, 8>" if mode != "uniform" else "3, 1>"),
                                        "traffic_source": traffic_file,
                                        "traffic_measured_in": "builder's rocprofv3 --pmc run (replayed from the committed file)",
                                        "ms_per_launch": prof["schur_gather"]}
        if not b_dense:   # then the gather is the dominant kernel of the step
            out["roofline"] = out["roofline_schur_gather"]
    if "reduced_sparse" in prof and red["l_nnz"] > 0:
        # the reduced camera system through the sparse block path (inner plan: its own nested dissection, a dense top on the
        # matrix cores where it has one).  Priced both ways: SURVEY 8d bytes (8 (nnz + lnz) for the factor, 16 lnz + 32 n for
        # the substitutions) against HBM, and the inner plan's flops against the fp64 MFMA peak -- with a dense top the
        # flops are what the phase is made of, without one it is a chain of small launches and neither roof is near
        r_bytes = 8.0 * (red["nnz_upper"] + red["l_nnz"]) + 16.0 * red["l_nnz"] + 32.0 * red["n_scalars"]
        r_flops = red["factor_flops"] + red["solve_flops"]
        t_s = prof["reduced_sparse"] * 1e-3
        b_top = red["schur_dim"] > 0
        out["roofline_reduced_sparse"] = {
            "bound": "mfma" if b_top else "hbm", "kernel": "reduced camera system: sparse block Cholesky + substitutions "
            f"({red['n_stages']} stages{', dense top of dimension %d on the matrix cores' % red['schur_dim'] if b_top else ''})",
            "achieved": (r_flops / t_s / 1e12) if b_top else (r_bytes / t_s / 1e9),
            "peak": F64_MFMA_PEAK_TFLOPS if b_top else HBM_PEAK_GBS, "unit": "TFLOP/s" if b_top else "GB/s",
            "frac": (r_flops / t_s / 1e12 / F64_MFMA_PEAK_TFLOPS) if b_top else (r_bytes / t_s / 1e9 / HBM_PEAK_GBS),
            "traffic": None, "algorithmic_flops": r_flops, "algorithmic_bytes": r_bytes, "ms_per_step": prof["reduced_sparse"],
            "l_nnz": red["l_nnz"], "dense_top_dim": red["schur_dim"]}
    if world == 1:
        x = bufs[-1].cpu().numpy()
        out["solve_residual_rel_inf"] = float(np.abs(lam.to_scipy() @ x - lam.rhs).max() / np.abs(lam.rhs).max())
        if not args.no_cpu_baseline and schur_sparse != 0 and extras:
            # the reference factors S densely whatever its structure: n^3/3 for it even where the GPU path counts none
            n3 = float(N) ** 3 / 3.0
            out["cpu_baseline"] = cpu_baseline_ba(lam, schur_flops + (dense_flops if b_dense else n3), x)
            out["solve_x_vs_reference_rel_inf"] = out["cpu_baseline"]["x_gpu_vs_reference_rel_inf"] if out["cpu_baseline"] else None
        if extras and schur_sparse != 0:
            out["host_path"] = host_path_leg(lambda: CLinearSolver_Schur_HIP(device=local_rank), lam, reps=3)
            if not args.no_cpu_baseline:
                out["dropin_cpp"] = dropin_leg(lam, reps=3)
        if extras and schur_sparse != 0:
            out["marginals"] = marginals_leg(args, solver, lam, vals, dev, torch)
            out["incremental_schur_update"] = incremental_leg(lam, dev, local_rank, torch, share=0.01)       # the list is used
            out["incremental_schur_update_10pct"] = incremental_leg(lam, dev, local_rank, torch, share=0.1)  # answered with a rebuild
            # Lambda of the same structure assembled on the device from 2-d projection residuals (one edge per observation)
            out["assembly"] = assembly_leg(solver, lam, dev, reps=5, rd=2, column_vertex_first=True)
    return out


def _r(v, sig=4):
    """Numbers of the compact line: four significant digits."""
    if isinstance(v, bool) or v is None or isinstance(v, (str, int)):
        return v
    if isinstance(v, float):
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k_: _r(x, sig) for k_, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _short_roofline(r):
    if not r:
        return None
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "share_of_step_time", "avg_launch_us", "launches_per_step")
    out = {k_: r[k_] for k_ in keep if k_ in r}
    if isinstance(out.get("kernel"), str) and len(out["kernel"]) > 90:
        out["kernel"] = out["kernel"][:87] + "..."
    return out


def _leg_summary(leg):
    """One BA / small-config leg of the full record as {ms_per_step, frac, bound, ...}."""
    if not leg:
        return None
    r = leg.get("roofline") or {}
    s = {"ms_per_step": leg.get("ms_per_step", leg.get("ms_per_solve")), "bound": r.get("bound"), "frac": r.get("frac")}
    if leg.get("roofline_reduced_sparse"):
        s["reduced_solve_ms"] = leg["roofline_reduced_sparse"].get("ms_per_step")
    if leg.get("roofline_schur_assembly_all"):
        s["assembly_frac_all_kernels"] = leg["roofline_schur_assembly_all"].get("frac")
    if leg.get("solve_residual_rel_inf") is not None:
        s["resid"] = leg["solve_residual_rel_inf"]
    if leg.get("cpu_baseline"):
        s["cpu_ref_ms"] = leg["cpu_baseline"].get("ms_per_solve")
        s["x_vs_ref"] = leg["cpu_baseline"].get("x_gpu_vs_reference_rel_inf")
    if leg.get("reference_cholmod_ms") is not None:
        s["cpu_ref_ms"] = leg["reference_cholmod_ms"]
    if leg.get("analyze_ms_cold") is not None:
        s["analyze_ms_cold"] = leg["analyze_ms_cold"]
    if leg.get("host_path"):
        s["host_warm_ms"] = leg["host_path"].get("warm_host_ms")
    if leg.get("dropin_cpp") and isinstance(leg["dropin_cpp"], dict) and "hip_warm_ms_median" in leg["dropin_cpp"]:
        s["dropin_warm_ms"] = leg["dropin_cpp"]["hip_warm_ms_median"]
    return s


COMPACT_LIMIT = 8000     # bytes: the driver's record keeps an 8 KB tail, and round 3's 21.5 KB line was not parsed


def compact_line(out, full_path):
    """The ONE stdout line: the contract's keys, `roofline`, `cpu_baseline`, what a caller with host arrays pays, and one
    {ms_per_step, frac} entry per other leg.  Everything else lives in the side file `full`."""
    line = {k_: out.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                      "scaling", "vs_baseline", "dtype", "data", "config")}
    line["roofline"] = _short_roofline(out.get("roofline"))
    for k_ in ("roofline_leaf_kernel", "roofline_whole_step"):
        if out.get(k_):
            line[k_] = _short_roofline(out[k_])
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k_: cb[k_] for k_ in ("value", "unit", "cores", "kind", "ms_per_solve", "numeric_only_ms") if k_ in cb}
        line["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:160]
        if "native_block_solver_ms" in cb:
            line["cpu_baseline"]["native_block_solver_warm_ms"] = cb["native_block_solver_ms"].get("warm")
            line["cpu_baseline"]["native_block_solver_first_call_ms"] = cb["native_block_solver_ms"].get("first_call")
    else:
        line["cpu_baseline"] = None
    for k_ in ("solve_residual_rel_inf", "solve_x_vs_reference_rel_inf", "inter_oracle_spread", "cond_proxy", "exchange", "rccl_ranks",
               "dist_backend", "host_path_speedup_vs_single_device", "north_star_4x_read_on"):
        if out.get(k_) is not None:
            line[k_] = out[k_]
    # SURVEY 8d's "warm" and "cold": a caller with host arrays / a CUberBlockMatrix (PCIe inclusive; never `value`)
    hp, dc = out.get("host_path"), out.get("dropin_cpp")
    if hp:
        line["ms_per_step_host_warm"] = hp.get("warm_host_ms")
        line["ms_cold"] = hp.get("cold_ms")
    if isinstance(dc, dict) and "hip_warm_ms_median" in dc:
        line["ms_per_step_dropin_warm"] = dc.get("hip_warm_ms_median")
        line["ms_dropin_first_call"] = dc.get("hip_cold_ms")
    if out.get("own_ordering"):
        line["analyze_ms_cold"] = out["own_ordering"].get("analyze_ms_cold")
    # three ratios against the reference on this box's host, each between like quantities (none of them is `value`, none is
    # a claim about kernel quality -- the roofline fraction is): numeric phases against numeric phases with the inputs where
    # each side keeps them; what a caller that swaps CLinearSolver_CholMod for CLinearSolver_HIP sees per iteration; and the
    # same caller against the reference's fastest solver with a cached analysis (its native block Cholesky)
    if cb and out.get("ms_per_step"):
        sp = {}
        if cb.get("numeric_only_ms"):
            sp["numeric_phases_device_resident"] = cb["numeric_only_ms"] / out["ms_per_step"]
        dropin = line.get("ms_per_step_dropin_warm") or line.get("ms_per_step_host_warm")
        if dropin and cb.get("ms_per_solve"):
            sp["dropin_caller_vs_cholmod_per_call"] = cb["ms_per_solve"] / dropin
        if dropin and (cb.get("native_block_solver_ms") or {}).get("warm"):
            sp["dropin_caller_vs_reference_best_warm"] = cb["native_block_solver_ms"]["warm"] / dropin
        if sp:
            line["speedup_vs_reference"] = sp
    legs = {}
    for key in ("ba_schur", "ba_schur_band", "ba_schur_uniform_dense_S", "ba_schur_venice", "ba_c5", "ba_1k_1m"):
        if out.get(key):
            legs[key] = _leg_summary(out[key])
    for key, rec in (out.get("other_configs") or {}).items():
        legs[key] = _leg_summary(rec)
    if out.get("replicas_one_gpu"):
        r1 = out["replicas_one_gpu"]
        legs["replicas_one_gpu"] = {k_: {"ms": v_["ms_per_round"], "GFLOP/s": v_["GFLOP/s"], "hbm_frac": v_["hbm_frac_whole_step"]}
                                    for k_, v_ in r1["by_k"].items()}
        if r1.get("batched"):
            legs["replicas_batched"] = {k_: {"ms": v_["ms_per_round"], "GFLOP/s": v_["GFLOP/s"], "hbm_frac": v_["hbm_frac_whole_step"],
                                             "resid": v_["last_member_resid"]} for k_, v_ in r1["batched"].items()}
    if out.get("pose_graph_replicas"):
        legs["pose_graph_replicas"] = {"ms_per_step": out["pose_graph_replicas"].get("ms_per_step"), "value": out["pose_graph_replicas"].get("value")}
    if out.get("device_group"):
        g = out["device_group"]
        legs["device_group"] = {k_: g.get(k_) for k_ in ("members", "exchange", "rccl_ranks", "warm_host_ms", "single_device_warm_host_ms",
                                                          "speedup_vs_single_device", "resid")}
    if legs:
        line["legs"] = legs
    if out.get("strong_scaling_n1"):
        line["strong_scaling_n1"] = {a: b for a, b in out["strong_scaling_n1"].items() if a not in ("workload", "note", "metric")}
    if out.get("scaling_model"):
        line["scaling_model"] = {key: ({a: b for a, b in m.items() if a in ("serial_ms", "sharded_ms", "device_resident", "host_arrays")}
                                       if not key.startswith("c4_") else        # (the C4-size legs: the 8-GPU figures only, the rest is in the full record)
                                       {"serial_ms": m["serial_ms"], "sharded_ms": m["sharded_ms"], "device_resident_at_8": m["device_resident"]["8"],
                                        "host_arrays_at_8": (m.get("host_arrays") or {}).get("8")})
                                 for key, m in out["scaling_model"].items() if isinstance(m, dict) and "device_resident" in m}
    line["full"] = full_path
    text = json.dumps(_r(line))
    if len(text) >= COMPACT_LIMIT:       # never again an unparseable headline: shed the optional parts, keep the contract
        for k_ in ("legs", "scaling_model", "roofline_whole_step", "roofline_leaf_kernel"):
            line.pop(k_, None)
            text = json.dumps(_r(line))
            if len(text) < COMPACT_LIMIT:
                break
    assert len(text) < COMPACT_LIMIT, len(text)
    return text


PCIE_GBS = 54.0       # one device's host link as measured on this pool (pinned H2D); spec 63
XGMI_LINK_GBS = 153.0  # one xGMI link, one direction (MI355X_MICROARCH / north star: 7 links per GPU)


def scaling_model(leg, n_values=None, n_scalars=None, n_exchange_doubles=None):
    """Amdahl model of the landmark-sharded solve from the one-GPU phases of the same system, for the two ways the north
    star's ">= 4x at 8 GPUs" can be read:

    device_resident  Lambda and eta already in HBM on every rank (what `value` measures): assembly and the landmarks'
                     back-substitution shard; the reduced camera system's solve is repeated on every rank (serial); the
                     exchange is one ring all-reduce of the packed blocks of S, 2 (N-1)/N of its bytes over one xGMI link
                     per neighbour.
    host_arrays      what a drop-in caller pays (the `device_group` leg measures it at N > 1): on top of the above every
                     member uploads its shard of the values and right-hand side and downloads its part of the solution
                     over its OWN PCIe link (bytes / N each; one link carries all of it at N = 1).

    Measured values replace the model where the driver runs N > 1; the model says what to expect and which reading of the
    target can hold: device-resident is capped by the serial reduced solve, the host path is not (its dominant term, the
    transfer, shards)."""
    ph = leg.get("phases_ms") or {}
    serial = sum(ph.get(k_, 0.0) for k_ in ("reduced_sparse", "dense_chol", "dense_solve", "schur_init"))
    total = leg["ms_per_step"]
    sharded = max(total - serial, 0.0)
    out = {"workload": leg.get("workload"), "serial_ms": serial, "sharded_ms": sharded}
    ex_bytes = 8.0 * n_exchange_doubles if n_exchange_doubles else 0.0

    def allreduce_ms(n_):
        return 2.0 * (n_ - 1) / n_ * ex_bytes / (XGMI_LINK_GBS * 1e9) * 1e3 if n_ > 1 else 0.0

    out["device_resident"] = {str(n_): total / (serial + sharded / n_ + allreduce_ms(n_)) for n_ in (2, 4, 8)}
    out["allreduce_ms"] = {str(n_): allreduce_ms(n_) for n_ in (2, 4, 8)}
    if n_values and n_scalars:
        xfer = (8.0 * n_values + 2 * 8.0 * n_scalars) / (PCIE_GBS * 1e9) * 1e3     # values and eta up, the solution down
        out["host_transfer_ms_one_link"] = xfer
        out["host_arrays"] = {str(n_): (xfer + total) / (xfer / n_ + serial + sharded / n_ + allreduce_ms(n_)) for n_ in (2, 4, 8)}
    out["predicted_speedup"] = out["device_resident"]   # (the key earlier rounds printed)
    out["note"] = ("serial = reduced camera system (every rank factors the same S); sharded = Schur assembly + landmark "
                   "back-substitution; the >= 4x at 8 GPUs of the north star is reachable on the host-array path (transfers over 8 "
                   "PCIe links), not device-resident")
    return out


def device_group_leg(args, n_members, one_device, lam=None):
    """The path a drop-in SLAM++ binary takes with SLAMPP_HIP_DEVICES=0,..,N-1: ONE process, one handle made by
    slampp_hip_create_multi over N devices, host arrays in, solution out (PCIe inclusive).  Next to it the same call on a
    one-device handle.  Reference counterpart: one caller thread, NonlinearSolver_Lambda_LM.h:1543-1552."""
    from slam_plus_plus_amd import synth
    from slam_plus_plus_amd.hip_solver import CLinearSolver_Schur_HIP
    if lam is None:
        lam = synth.ba(args.c5_cams, args.c5_points, k=4, mode=args.c5_mode, seed=777)
    devices = [0] * n_members if one_device else list(range(n_members))
    out = {"workload": f"C5: BA {args.c5_cams} cams x {args.c5_points} points through slampp_hip_create_multi({devices}), host arrays in and out",
           "devices": devices}
    keep = {}
    for name, devs in (("single", [devices[0]]), ("group", devices)):
        solver = CLinearSolver_Schur_HIP(device=devs[0]) if name == "single" else CLinearSolver_Schur_HIP(devices=devs)
        r = host_path_leg(lambda: solver, lam, reps=args.group_reps)
        eta = lam.rhs.copy()
        ok = solver.Solve_PosDef_Blocky(lam, eta)
        r["resid"] = float(np.abs(lam.to_scipy() @ eta - lam.rhs).max() / np.abs(lam.rhs).max()) if ok else None
        if name == "group":
            info = solver.group_info()
            r.update(members=info["members"], exchange=info["exchange"])
        keep[name] = r
        del solver
    g = keep["group"]
    ex = g.get("exchange") or ""
    out.update(members=g.get("members"), exchange=ex, rccl_ranks=(g.get("members") if ex.startswith("rccl") else 0),
               warm_host_ms=g["warm_host_ms"], cold_ms=g["cold_ms"], resid=g["resid"], last_call_ms=g["last_call_ms"],
               single_device_warm_host_ms=keep["single"]["warm_host_ms"], single_device_cold_ms=keep["single"]["cold_ms"],
               speedup_vs_single_device=keep["single"]["warm_host_ms"] / g["warm_host_ms"], ok=bool(g["ok"]))
    return out


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
    legs = [m for m in args.ba_legs.split(",") if m]

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
