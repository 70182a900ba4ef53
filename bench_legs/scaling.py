"""What N GPUs can buy a landmark-sharded BA solve, from the one-GPU phases (Amdahl model, both readings of the north star's
">= 4x at 8 GPUs"), and the in-library device group leg (one handle over N devices, host arrays in and out)."""
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
from .line import *  # noqa: F401,F403
from .line import _r, _short_roofline, _leg_summary  # noqa: F401

__all__ = ['PCIE_GBS', 'XGMI_LINK_GBS', 'scaling_model', 'device_group_leg']


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
