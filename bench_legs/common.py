"""What every leg of bench.py shares: the peaks the rooflines are priced against, the committed counter passes, the two host paths
(host arrays through the C ABI, a CUberBlockMatrix through include/slam/LinearSolver_HIP.h), the all-reduce callback of the N > 1 legs."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

__all__ = ['ROOT', 'HBM_PEAK_GBS', 'F64_MFMA_PEAK_TFLOPS', 'F64_MFMA_SUSTAINED_TFLOPS', 'C3_REF', '_KEEP', 'host_cores', 'log', 'timed_cold_analysis', 'load_traffic', 'kernel_traffic', 'kernel_traffic_mean', 'kernel_traffic_sum', 'host_path_leg', 'dropin_leg', '_DevPtr', 'make_allreduce', 'dataclasses_replace_points', 'shared_system']


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # the repository
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (vendor figure quoted in SURVEY.md section 8d)
# what a loop of nothing but v_mfma_f64_16x16x4 sustains on this chip (tools/micro/mfma_f64_peak.hip, profiles/r06_mfma_f64_peak.txt:
# 47-49 TFLOP/s with 2-5 waves resident per SIMD = 101-105 clocks per instruction per SIMD, of which the matrix pipe is busy
# 64 (SQ_VALU_MFMA_BUSY_CYCLES); v_mfma_f64_4x4x4_4b sustains 73 = 93 % of the peak, from operands in registers --
# profiles/r06_mfma_4x4x4_experiment.txt is what it did inside the update kernel; the vector unit's v_fma_f64 63-70):
# printed beside `peak`, never instead of it
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
def timed_cold_analysis(solver, lam):
    """analyze_ms_cold: set_structure + analyze of a fresh handle, on an idle device, in a process whose runtime is up.
    (i) The first handle of a process pays the runtime's first uses -- streams, first copies, code objects: 25 ms and more
    (profiles/r06_first_launch_cost.txt) -- which are no part of any analysis: a throwaway handle takes them, once.
    (ii) A handle destroyed a moment ago leaves the driver releasing its memory, and the first copy on the next handle's
    stream then completes 4-12 ms late (seen as C1 / C2 at 18-27 instead of 9-10 ms): the device is
    given 30 ms to settle before the clock starts."""
    import torch
    if not _KEEP.get("runtime_warm"):
        from slam_plus_plus_amd import synth
        from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
        warm = CLinearSolver_HIP(device=torch.cuda.current_device())
        warm.SymbolicDecomposition_Blocky(synth.pose_chain(n=64))
        del warm
        _KEEP["runtime_warm"] = True
    torch.cuda.synchronize()
    time.sleep(0.03)
    t0 = time.perf_counter()
    solver.SymbolicDecomposition_Blocky(lam)
    return (time.perf_counter() - t0) * 1e3
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
