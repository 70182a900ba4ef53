"""bench.py's stdout contract: ONE compact JSON line (under the driver's 8 KB window), the full record in a side file."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import _KEEP, _DevPtr  # noqa: F401
from .c3 import *  # noqa: F401,F403
from .small import *  # noqa: F401,F403
from .ba import *  # noqa: F401,F403

__all__ = ['_r', '_short_roofline', '_leg_summary', 'COMPACT_LIMIT', 'compact_line']


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
    if r.get("name"):   # (BA legs: the roofline of the phase the step spends most of its time in)
        s["of"] = r["name"].replace("roofline_", "")
        s["share"] = r.get("share_of_step")
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
