"""Host-side mirror of the reference's linear-solver interface over the C ABI of libslampp_hip.so.

The reference binds a linear solver as a duck-typed template parameter
(/root/reference/include/slam/LinearSolverTags.h:38-135); the C++ binding is
include/slam/LinearSolver_HIP.h.  This module is the same surface for Python callers and for the
parity tests: same method names, same argument meaning (``eta`` is overwritten with the solution),
same error behaviour (``False`` = not positive definite, ``MemoryError`` = std::bad_alloc,
``RuntimeError`` = std::runtime_error) as

* ``CLinearSolver_CholMod`` / ``CLinearSolver_UberBlock``  (LinearSolver_CholMod.h:148-214,
  LinearSolver_UberBlock.h:256-426)  ->  :class:`CLinearSolver_HIP`
* ``CLinearSolver_Schur``  (LinearSolver_Schur.h:1423-1935)  ->  :class:`CLinearSolver_Schur_HIP`
* ``CNonlinearSolver_Lambda::Refresh_Lambda`` (NonlinearSolver_Lambda_Base.h:1634-1688) for one edge set
  ->  :class:`CLambdaAssembly_HIP`

There is no CPU fallback: if the shared library is missing, or there is no GPU, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:  # (optional: only the structure keys of this mirror use it)
    import xxhash as _xxhash
except ImportError:
    _xxhash = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslampp_hip.so")

OK, NOT_POSDEF = 0, 1
ERR_INVALID, ERR_ALLOC, ERR_DEVICE, ERR_UNSUPPORTED = -1, -2, -3, -4
MODE_SPARSE, MODE_SCHUR = 0, 1


def _hash_array(a: np.ndarray) -> int:
    """Hash of an index array's bytes: xxh3 over the buffer where the module is there (6 ms for C5's 10^7 block rows),
    Python's hash of a copy otherwise (50 ms)."""
    a = np.ascontiguousarray(a)
    if _xxhash is not None:
        return _xxhash.xxh3_64_intdigest(a)
    return hash(a.tobytes())


class Times(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "order_ms", "symbolic_ms", "upload_ms", "factor_ms", "solve_ms", "download_ms",
        "schur_ms", "reduce_ms", "cholsol_ms", "backsubst_ms", "total_ms")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [("n_bcols", C.c_int64), ("n_blocks_upper", C.c_int64), ("n_scalars", C.c_int64),
                ("nnz_upper", C.c_int64), ("l_blocks", C.c_int64), ("l_nnz", C.c_int64),
                ("factor_flops", C.c_double), ("solve_flops", C.c_double),
                ("n_stages", C.c_int64), ("n_tasks", C.c_int64), ("etree_height", C.c_int64),
                ("n_update_pairs", C.c_int64), ("n_cams", C.c_int64), ("n_points", C.c_int64),
                ("n_observations", C.c_int64), ("schur_dim", C.c_int64), ("device_bytes", C.c_int64),
                ("n_bottom_stages", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PlanView(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_bcols", "l_blocks", "n_pairs", "n_row_entries",
                                         "n_stages", "n_tasks", "n_task_cols", "l_values")] + [
        ("p_perm", C.c_void_p), ("p_dim", C.c_void_p), ("p_lptr", C.c_void_p), ("p_lrow", C.c_void_p),
        ("p_loff", C.c_void_p), ("p_asrc", C.c_void_p), ("p_atrans", C.c_void_p), ("p_pptr", C.c_void_p),
        ("p_pa", C.c_void_p), ("p_pb", C.c_void_p), ("p_rptr", C.c_void_p), ("p_rblk", C.c_void_p),
        ("p_stage_ptr", C.c_void_p), ("p_task_ptr", C.c_void_p), ("p_task_cols", C.c_void_p),
        ("p_dense_pos", C.c_void_p), ("dense_dim", C.c_int64)]


class ShardView(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_bcols", "n_blocks", "n_camera_values", "n_value_begin", "n_value_end", "n_camera_scalars", "n_scalar_begin",
        "n_scalar_end", "n_point_begin", "n_point_end")] + [
        ("p_bcol_cumsum", C.c_void_p), ("p_bcol_ptr", C.c_void_p), ("p_brow_idx", C.c_void_p)]


class EdgeSetPtrs(C.Structure):
    _fields_ = [("p_assembly", C.c_void_p), ("p_J0_dev", C.c_void_p), ("p_J1_dev", C.c_void_p), ("p_sigma_inv_dev", C.c_void_p),
                ("p_error_dev", C.c_void_p), ("p_weight_dev", C.c_void_p)]


class PhaseTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("n_count", C.c_int64), ("f_total_ms", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

# every symbol include/slampp_hip.h declares: (restype, argtypes)
_P = C.c_void_p
ABI = {
    "slampp_hip_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "slampp_hip_create_multi": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_int), C.c_int]),
    "slampp_hip_group_info": (C.c_int, [_P, C.POINTER(C.c_int), _P, C.c_int, C.POINTER(C.c_char_p)]),
    "slampp_hip_group_exchange_count": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "slampp_hip_group_exchange_selftest": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int64, C.c_char_p, C.c_int]),
    "slampp_hip_landmark_shard": (C.c_int, [C.c_int64, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.POINTER(ShardView)]),
    "slampp_hip_destroy": (None, [_P]),
    "slampp_hip_free_memory": (C.c_int, [_P]),
    "slampp_hip_last_error": (C.c_char_p, [_P]),
    "slampp_hip_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "slampp_hip_set_structure": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "slampp_hip_analyze": (C.c_int, [_P, C.c_int, C.c_int64]),
    "slampp_hip_factor_solve": (C.c_int, [_P, _P, _P, C.POINTER(Times)]),
    "slampp_hip_factor_solve_device": (C.c_int, [_P, _P, _P, C.POINTER(Times)]),
    "slampp_hip_host_staging": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "slampp_hip_upload_values_async": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "slampp_hip_solve_again": (C.c_int, [_P, _P]),
    "slampp_hip_factorize": (C.c_int, [_P, _P, _P]),
    "slampp_hip_factor_structure": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P, _P, _P, _P, _P]),
    "slampp_hip_schur_set_changed_points": (C.c_int, [_P, _P, C.c_int64]),
    "slampp_hip_solve_marginal_poses": (C.c_int, [_P, _P, _P]),
    "slampp_hip_solve_marginal_poses_device_async": (C.c_int, [_P, _P, _P]),
    "slampp_hip_apply_damping_device_async": (C.c_int, [_P, _P, C.c_double, C.c_int64, C.c_int64]),
    "slampp_hip_marginals": (C.c_int, [_P, _P, _P]),
    "slampp_hip_marginals_device_async": (C.c_int, [_P, _P, _P]),
    "slampp_hip_schur_marginals": (C.c_int, [_P, _P, _P, _P]),
    "slampp_hip_schur_marginals_device_async": (C.c_int, [_P, _P, _P, _P]),
    "slampp_hip_factor_solve_device_async": (C.c_int, [_P, _P, _P]),
    "slampp_hip_sync": (C.c_int, [_P]),
    "slampp_hip_factor_solve_batch_device_async": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.c_int64]),
    "slampp_hip_sync_batch": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int]),
    "slampp_hip_stream": (_P, [_P]),
    "slampp_hip_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "slampp_hip_get_reduced_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "slampp_hip_get_profile": (C.c_int, [_P, C.POINTER(PhaseTime), C.c_int, C.POINTER(C.c_int), C.c_int]),
    "slampp_hip_get_profile_reference_names": (C.c_int, [_P, C.POINTER(PhaseTime), C.c_int, C.POINTER(C.c_int)]),
    "slampp_hip_set_allreduce": (C.c_int, [_P, ALLREDUCE_FN, _P]),
    "slampp_hip_assembly_create": (C.c_int, [_P, C.POINTER(_P), C.c_int64, _P, _P, C.c_int]),
    "slampp_hip_assembly_destroy": (None, [_P]),
    "slampp_hip_assemble_device_async": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, _P, _P, _P, C.c_int]),
    "slampp_hip_assemble_sets_device_async": (C.c_int, [C.POINTER(EdgeSetPtrs), C.c_int, C.c_int64, _P, _P, _P, _P, C.c_int]),
    "slampp_hip_get_plan": (C.c_int, [_P, C.POINTER(PlanView)]),
    "slampp_hip_plan_create": (C.c_int, [C.POINTER(_P), C.c_int64, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "slampp_hip_plan_get": (C.c_int, [_P, C.POINTER(PlanView), C.POINTER(Stats)]),
    "slampp_hip_plan_destroy": (None, [_P]),
}

_lib = None


def load_library() -> C.CDLL:
    """Loads libslampp_hip.so (built in-tree by ``__graft_entry__.build()``); raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        try:
            # Python callers get their device memory from torch, which bundles its own HIP runtime: load that one
            # first, so that this library binds to it too -- two HIP runtimes in one process do not share the GPU
            # (whichever initialises second reports "no HIP GPUs")
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(lib, name)   # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


_PLAN_FIELDS = [("perm", "p_perm", np.int32, "n_bcols", 0), ("dim", "p_dim", np.int32, "n_bcols", 0),
                ("lptr", "p_lptr", np.int64, "n_bcols", 1), ("lrow", "p_lrow", np.int32, "l_blocks", 0),
                ("loff", "p_loff", np.int64, "l_blocks", 0), ("asrc", "p_asrc", np.int64, "l_blocks", 0),
                ("atrans", "p_atrans", np.int32, "l_blocks", 0), ("pptr", "p_pptr", np.int64, "l_blocks", 1),
                ("pa", "p_pa", np.int32, "n_pairs", 0), ("pb", "p_pb", np.int32, "n_pairs", 0),
                ("rptr", "p_rptr", np.int64, "n_bcols", 1), ("rblk", "p_rblk", np.int32, "n_row_entries", 0),
                ("stage_ptr", "p_stage_ptr", np.int32, "n_stages", 1),
                ("task_ptr", "p_task_ptr", np.int64, "n_tasks", 1),
                ("task_cols", "p_task_cols", np.int32, "n_task_cols", 0),
                ("dense_pos", "p_dense_pos", np.int32, "n_bcols", 0)]


def _fetch_plan(getter) -> dict:
    v = PlanView()
    getter(v)                                   # sizes
    out = {}
    for name, field, dt, size_field, extra in _PLAN_FIELDS:
        arr = np.zeros(getattr(v, size_field) + extra, dtype=dt)
        out[name] = arr
        setattr(v, field, arr.ctypes.data)
    getter(v)                                   # contents
    for f in ("n_bcols", "l_blocks", "n_pairs", "n_row_entries", "n_stages", "n_tasks", "l_values", "dense_dim"):
        out[f] = getattr(v, f)
    return out


def group_exchange_selftest(devices, exchange: int = 0, count: int = 1 << 16):
    """(status, name of the exchange used): the all-reduce of a multi-device handle by itself (slampp_hip_group_exchange_selftest)."""
    lib = load_library()
    ids = (C.c_int * len(devices))(*[int(d) for d in devices])
    buf = C.create_string_buffer(256)
    rc = lib.slampp_hip_group_exchange_selftest(ids, len(devices), int(exchange), int(count), buf, 256)
    return rc, buf.value.decode()


def landmark_shard_structure(lam, rank: int, world: int) -> dict:
    """The library's own shard splitter (slampp_hip_landmark_shard; host code, no GPU): structure arrays and value /
    right-hand side ranges of the landmark shard ``rank`` of ``world`` holds."""
    lib = load_library()
    cs = np.ascontiguousarray(lam.cumsum, dtype=np.int64)
    bp = np.ascontiguousarray(lam.bcol_ptr, dtype=np.int64)
    br = np.ascontiguousarray(lam.brow_idx, dtype=np.int32)
    v = ShardView()
    args = (lam.n_bcols, _ptr(cs), _ptr(bp), _ptr(br), int(lam.n_matrix_cut), int(rank), int(world))
    if lib.slampp_hip_landmark_shard(*args, C.byref(v)) != OK:
        raise ValueError("slampp_hip_landmark_shard refused the arguments")
    out = {"cumsum": np.zeros(v.n_bcols + 1, dtype=np.int64), "bcol_ptr": np.zeros(v.n_bcols + 1, dtype=np.int64),
           "brow_idx": np.zeros(v.n_blocks, dtype=np.int32)}
    v.p_bcol_cumsum, v.p_bcol_ptr, v.p_brow_idx = (out[k].ctypes.data for k in ("cumsum", "bcol_ptr", "brow_idx"))
    if lib.slampp_hip_landmark_shard(*args, C.byref(v)) != OK:
        raise ValueError("slampp_hip_landmark_shard refused the arguments")
    for k, _ in ShardView._fields_[:10]:
        out[k] = int(getattr(v, k))
    return out


def host_plan(lam, leaf_size: int = 0, subtree_size: int = 0, dense_top_nb: int = -1):
    """Ordering + symbolic analysis + schedule on the host only (no GPU): (plan dict, stats dict).
    ``dense_top_nb`` < 0 = the library default, 0 = no dense top."""
    lib = load_library()
    h = C.c_void_p()
    cs = np.ascontiguousarray(lam.cumsum, dtype=np.int64)
    bp = np.ascontiguousarray(lam.bcol_ptr, dtype=np.int64)
    br = np.ascontiguousarray(lam.brow_idx, dtype=np.int32)
    rc = lib.slampp_hip_plan_create(C.byref(h), lam.n_bcols, _ptr(cs), _ptr(bp), _ptr(br),
                                    leaf_size, subtree_size, dense_top_nb)
    if rc != OK:
        raise ValueError(f"slampp_hip_plan_create failed ({rc})")
    try:
        st = Stats()

        def getter(v):
            if lib.slampp_hip_plan_get(h, C.byref(v), C.byref(st)) != OK:
                raise RuntimeError("slampp_hip_plan_get failed")
        plan = _fetch_plan(getter)
        return plan, st.as_dict()
    finally:
        lib.slampp_hip_plan_destroy(h)


class _SolverBase:
    """Shared plumbing: handle life cycle, structure caching, status -> exception mapping."""

    _mode = MODE_SPARSE

    def __init__(self, device: int = 0, devices=None, **options):
        """``devices``: a list of HIP device ordinals -- one handle over several GPUs of this process
        (slampp_hip_create_multi): a BA system is cut into landmark shards, one per listed device, and the reduced camera
        system is summed inside the library (RCCL, or peer pointers); pose graphs run on ``devices[0]``."""
        self._lib = load_library()
        self._h = C.c_void_p()
        self._devices = None if devices is None else [int(d) for d in devices]
        if self._devices is not None:
            if not self._devices:
                raise ValueError("devices must name at least one device")
            ids = (C.c_int * len(self._devices))(*self._devices)
            rc = self._lib.slampp_hip_create_multi(C.byref(self._h), ids, len(self._devices))
            device = self._devices[0]
        else:
            rc = self._lib.slampp_hip_create(C.byref(self._h), int(device))
        if rc == ERR_INVALID:
            self._h = None
            raise ValueError(f"slampp_hip_create_multi(devices={self._devices}) refused the device list" if self._devices is not None
                             else f"slampp_hip_create(device={device}) refused the device ordinal")
        if rc != OK:
            self._h = None
            raise RuntimeError(f"slampp_hip_create(device={device}) failed ({rc}): no usable HIP device")
        self._device = int(device)
        self._options = dict(options)
        for k, v in options.items():
            self._check(self._lib.slampp_hip_set_option(self._h, k.encode(), int(v)))
        self._structure_key = None
        self._analyzed = False
        self._keepalive = None
        self.times = Times()

    # copies do not carry state, as in the reference (LinearSolver_CholMod.h:163-167,
    # NonlinearSolver_Base.h:400,438), but they do keep the configuration
    def __copy__(self):
        return type(self)(**self._ctor_args())

    def _ctor_args(self):
        return dict(self._options, device=self._device, devices=self._devices)

    def group_info(self) -> dict:
        """Members in use, their landmark ranges and the exchange of a handle made with ``devices=[...]``
        (``{"members": 0, ...}`` while the handle is not solving with shards)."""
        n = C.c_int(0)
        name = C.c_char_p()
        bounds = np.zeros(17, dtype=np.int64)
        self._check(self._lib.slampp_hip_group_info(self._h, C.byref(n), _ptr(bounds), 16, C.byref(name)))
        return {"members": n.value, "point_bounds": bounds[:n.value + 1].tolist() if n.value else [],
                "exchange": (name.value or b"").decode()}

    def exchange_count(self) -> int:
        """Exchanges the members of a device group have gone into (slampp_hip_group_exchange_count): all of them or none."""
        n = C.c_int64(0)
        self._check(self._lib.slampp_hip_group_exchange_count(self._h, C.byref(n)))
        return int(n.value)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.slampp_hip_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _error(self) -> str:
        return (self._lib.slampp_hip_last_error(self._h) or b"").decode()

    def _check(self, rc: int) -> bool:
        if rc == OK:
            return True
        if rc == NOT_POSDEF:
            return False
        if rc == ERR_ALLOC:
            raise MemoryError(self._error())
        if rc == ERR_INVALID:
            raise ValueError(self._error())
        if rc == ERR_UNSUPPORTED:
            raise NotImplementedError(self._error())
        raise RuntimeError(self._error())

    def Free_Memory(self) -> None:
        self._check(self._lib.slampp_hip_free_memory(self._h))
        self._structure_key = None
        self._analyzed = False

    def Clear_SymbolicDecomposition(self) -> None:
        """LinearSolverTags.h:112-120: the block structure is about to change."""
        self._structure_key = None
        self._analyzed = False

    def _key(self, lam):
        """Identity of the block structure.  The same array objects as last time (what an iteration loop passes) are not
        hashed again -- 10^6 indices per call --, but they are not trusted blindly either: a caller that edits the index
        arrays in place (same objects, same shapes) would otherwise have the old analysis applied to a new pattern, a wrong
        answer without an error.  So the fast path compares a strided sample of each array (1024 entries + both ends)
        with what was there when the key was taken; a system that carries a ``structure_version`` counter is keyed on that
        as well.  An edit that misses every sampled entry still needs ``Clear_SymbolicDecomposition()``, as in the
        reference (LinearSolverTags.h:112-120)."""
        arrays = (lam.bcol_ptr, lam.brow_idx, lam.cumsum)
        ident = tuple(id(a) for a in arrays) + (getattr(lam, "structure_version", None),)
        if getattr(self, "_key_ident", None) == ident and all(a is b for a, b in zip(self._key_arrays, arrays)) and \
                all(np.array_equal(self._sample(a), s) for a, s in zip(arrays, self._key_samples)):
            return self._key_value
        key = (lam.n_bcols, lam.n_blocks, int(lam.cumsum[-1]),
               _hash_array(lam.bcol_ptr), _hash_array(lam.brow_idx), _hash_array(lam.cumsum))
        self._key_ident, self._key_arrays, self._key_value = ident, arrays, key
        self._key_samples = tuple(self._sample(a).copy() for a in arrays)
        return key

    @staticmethod
    def _sample(a: np.ndarray) -> np.ndarray:
        n = a.shape[0]
        if n <= 2048:
            return a
        return np.concatenate([a[::max(n // 1024, 1)], a[-2:]])

    def _n_matrix_cut(self, lam) -> int:
        return 0

    def SymbolicDecomposition_Blocky(self, lam) -> bool:
        cs = np.ascontiguousarray(lam.cumsum, dtype=np.int64)
        bp = np.ascontiguousarray(lam.bcol_ptr, dtype=np.int64)
        br = np.ascontiguousarray(lam.brow_idx, dtype=np.int32)
        self._check(self._lib.slampp_hip_set_structure(self._h, lam.n_bcols, _ptr(cs), _ptr(bp), _ptr(br)))
        self._check(self._lib.slampp_hip_analyze(self._h, self._mode, self._n_matrix_cut(lam)))
        self._structure_key = self._key(lam)
        self._n_values = int(lam.values.shape[0])
        self._analyzed = True
        return True

    def Solve_PosDef_Blocky(self, lam, eta: np.ndarray) -> bool:
        """Reuses ordering / symbolic analysis while the block structure is unchanged
        (LinearSolverTags.h:130-134).  ``eta``: rhs on entry, solution on return."""
        if eta.dtype != np.float64 or not eta.flags.c_contiguous or eta.shape != (lam.n_scalars,):
            raise ValueError("eta must be a contiguous float64 vector of the system's dimension")
        if not self._analyzed or self._structure_key != self._key(lam):
            self._check(self._lib.slampp_hip_set_option(self._h, b"staging_ahead", 1))   # host arrays: the staging beside the analysis
            self.SymbolicDecomposition_Blocky(lam)
        vals = np.ascontiguousarray(lam.values, dtype=np.float64)
        if vals.shape != (self._n_values,):
            raise ValueError("lam.values does not match the block structure")
        return self._check(self._lib.slampp_hip_factor_solve(self._h, _ptr(vals), _ptr(eta), C.byref(self.times)))

    def Solve_PosDef(self, lam, eta: np.ndarray) -> bool:
        """Cold solve: ordering + symbolic + numeric (LinearSolver_CholMod.cpp:264-358)."""
        self.Clear_SymbolicDecomposition()
        return self.Solve_PosDef_Blocky(lam, eta)

    def Solve_Again(self, eta: np.ndarray) -> bool:
        """Another right-hand side with the factor the last solve left behind (cholmod_solve on a kept factor,
        LinearSolver_CholMod.cpp:322-347)."""
        if self._structure_key is None or not self._analyzed:
            raise ValueError("Solve_Again: there is no factorization")
        n_scalars = self._structure_key[2]
        if eta.dtype != np.float64 or not eta.flags.c_contiguous or eta.shape != (n_scalars,):
            raise ValueError("eta must be a contiguous float64 vector of the system's dimension")
        return self._check(self._lib.slampp_hip_solve_again(self._h, _ptr(eta)))

    def host_staging(self):
        """(values, rhs): numpy views of the library's pinned staging for the current structure -- filling these and
        passing them to Solve_PosDef_Blocky spares the transfers a staging pass (slampp_hip_host_staging)."""
        pv, pr = C.c_void_p(), C.c_void_p()
        self._check(self._lib.slampp_hip_host_staging(self._h, C.byref(pv), C.byref(pr)))
        st = Stats()
        self._check(self._lib.slampp_hip_get_stats(self._h, C.byref(st)))
        n_values = int(self._n_values)
        values = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_double)), shape=(n_values,))
        rhs = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_double)), shape=(int(st.n_scalars),))
        return values, rhs

    def factor_structure(self) -> dict:
        """Block structure of the factor slampp_hip_factorize hands back, over the caller's block columns
        (slampp_hip_factor_structure): ``perm``, ``dim``, ``lptr``, ``lrow``, ``loff``, ``l_values``."""
        n, nb, nv = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self._lib.slampp_hip_factor_structure(self._h, C.byref(n), C.byref(nb), C.byref(nv), None, None, None, None, None))
        out = {"perm": np.zeros(n.value, dtype=np.int32), "dim": np.zeros(n.value, dtype=np.int32),
               "lptr": np.zeros(n.value + 1, dtype=np.int64), "lrow": np.zeros(nb.value, dtype=np.int32),
               "loff": np.zeros(nb.value, dtype=np.int64)}
        self._check(self._lib.slampp_hip_factor_structure(self._h, C.byref(n), C.byref(nb), C.byref(nv), _ptr(out["perm"]), _ptr(out["dim"]),
                                                          _ptr(out["lptr"]), _ptr(out["lrow"]), _ptr(out["loff"])))
        out["l_values"] = int(nv.value)
        return out

    def factorize(self, lam):
        """Numeric factor only (sparse mode): ``(ok, structure, l_values)`` -- the lower factor of the permuted Lambda in
        block-CSC over the caller's block columns (``structure['lptr']``, ``['lrow']``, ``['loff']``, ``['perm']``, ``['dim']``:
        factor_structure()); a dense top's columns and the pieces of block columns wider than 8 come back in that layout too."""
        if not self._analyzed or self._structure_key != self._key(lam):
            self.SymbolicDecomposition_Blocky(lam)
        plan = self.factor_structure()
        vals = np.ascontiguousarray(lam.values, dtype=np.float64)
        out = np.zeros(int(plan["l_values"]))
        ok = self._check(self._lib.slampp_hip_factorize(self._h, _ptr(vals), _ptr(out)))
        return ok, plan, out

    def stats(self) -> dict:
        st = Stats()
        self._check(self._lib.slampp_hip_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def reduced_stats(self) -> dict:
        """Schur mode, after a solve: stats of the inner sparse solver of the reduced camera system (all zero: dense)."""
        st = Stats()
        self._check(self._lib.slampp_hip_get_reduced_stats(self._h, C.byref(st)))
        return st.as_dict()

    def set_option(self, name: str, value: int) -> None:
        self._check(self._lib.slampp_hip_set_option(self._h, name.encode(), int(value)))

    def profile(self, reset: bool = False) -> dict:
        """{phase: (count, total_ms)} measured with HIP events on the solver's stream (option profile=1)."""
        buf = (PhaseTime * 32)()
        n = C.c_int(0)
        self._check(self._lib.slampp_hip_get_profile(self._h, buf, 32, C.byref(n), int(reset)))
        return {buf[i].name.decode(): (int(buf[i].n_count), float(buf[i].f_total_ms)) for i in range(min(n.value, 32))}

    def profile_reference_names(self) -> dict:
        """The same totals under the names the reference prints with __SCHUR_PROFILING (LinearSolver_Schur.h:1895-1912) /
        CHOLMOD's numeric phases: {name: (count, total_ms)}, in the reference's order (slampp_hip_get_profile_reference_names)."""
        buf = (PhaseTime * 16)()
        n = C.c_int(0)
        self._check(self._lib.slampp_hip_get_profile_reference_names(self._h, buf, 16, C.byref(n)))
        return {buf[i].name.decode(): (int(buf[i].n_count), float(buf[i].f_total_ms)) for i in range(min(n.value, 16))}

    def plan(self) -> dict:
        def getter(v):
            self._check(self._lib.slampp_hip_get_plan(self._h, C.byref(v)))
        return _fetch_plan(getter)

    # ---- device-resident entry points (bench.py; torch only provides the memory) ----
    def factor_solve_device(self, values_ptr: int, rhs_ptr: int) -> bool:
        return self._check(self._lib.slampp_hip_factor_solve_device(self._h, values_ptr, rhs_ptr, C.byref(self.times)))

    def factor_solve_device_async(self, values_ptr: int, rhs_ptr: int) -> None:
        self._check(self._lib.slampp_hip_factor_solve_device_async(self._h, values_ptr, rhs_ptr))

    def factor_solve_batch_device_async(self, n_batch: int, values_ptr: int, values_stride: int, rhs_ptr: int, rhs_stride: int) -> None:
        """K value sets of the analyzed structure in one pass of launches (slampp_hip_factor_solve_batch_device_async): member k
        reads values_ptr + 8 k values_stride and overwrites rhs_ptr + 8 k rhs_stride (strides in doubles)."""
        self._check(self._lib.slampp_hip_factor_solve_batch_device_async(self._h, int(n_batch), values_ptr, int(values_stride), rhs_ptr, int(rhs_stride)))

    def sync_batch(self, n_batch: int):
        """Waits for the batches enqueued so far; one bool per member: False = that member's matrix is not positive definite."""
        st = (C.c_int * int(n_batch))()
        self._check(self._lib.slampp_hip_sync_batch(self._h, st, int(n_batch)))
        return [v == 0 for v in st]

    def sync(self) -> bool:
        return self._check(self._lib.slampp_hip_sync(self._h))

    def apply_damping_device_async(self, values_ptr: int, f_alpha: float, n_first_vertex: int, n_last_vertex: int) -> None:
        """ApplyDamping of the reference's LM solver (NonlinearSolver_Lambda_LM.h:228-239) on device-resident values."""
        self._check(self._lib.slampp_hip_apply_damping_device_async(self._h, values_ptr, float(f_alpha), int(n_first_vertex),
                                                                    int(n_last_vertex)))

    def stream(self) -> int:
        return int(self._lib.slampp_hip_stream(self._h) or 0)

    def set_allreduce(self, fn) -> None:
        """fn(dev_ptr:int, count:int, stream:int) -> int; kept alive by the solver."""
        if fn is None:
            self._keepalive = None
            self._check(self._lib.slampp_hip_set_allreduce(self._h, C.cast(None, ALLREDUCE_FN), None))
            return

        def tramp(_ctx, p, n, s):
            try:
                return int(fn(int(p or 0), int(n), int(s or 0)))
            except Exception:      # never unwind through C
                return 1
        self._keepalive = ALLREDUCE_FN(tramp)
        self._check(self._lib.slampp_hip_set_allreduce(self._h, self._keepalive, None))


class CLinearSolver_HIP(_SolverBase):
    """Sparse block Cholesky on the GPU; stands where CLinearSolver_CholMod / _CSparse / _UberBlock do."""

    _Tag = "CBlockwiseLinearSolverTag"   # LinearSolverTags.h:54
    _mode = MODE_SPARSE


def _block_diagonal_marginals(self, lam):
    """Block diagonal of the covariance Lambda^-1 ([n, d, d]), as CMarginals::Calculate_DenseMarginals_Recurrent_FBS
    (.., mpart_Diagonal) gives the reference's nonlinear solvers (NonlinearSolver_Lambda.h:700-760): factorization and a
    sparse inverse subset on the factor's pattern (below a dense inverse of the plan's dense top, if it has one).  One block
    size (3, 6 or 7) -> an array; a mix of block sizes up to 8 (no dense top) -> a list of blocks."""
    if not self._analyzed or self._structure_key != self._key(lam):
        self.SymbolicDecomposition_Blocky(lam)
    dims = np.diff(lam.cumsum)
    vals = np.ascontiguousarray(lam.values, dtype=np.float64)
    if np.all(dims == dims[0]):
        d = int(dims[0])
        out = np.empty((len(dims), d, d), dtype=np.float64)
        if not self._check(self._lib.slampp_hip_marginals(self._h, _ptr(vals), _ptr(out))):
            raise ArithmeticError("Marginals: the system is not positive definite")
        return out.transpose(0, 2, 1).copy()      # blocks come column-major
    # mixed block sizes: a list of d_c x d_c blocks
    flat = np.empty(int((dims.astype(np.int64) ** 2).sum()), dtype=np.float64)
    if not self._check(self._lib.slampp_hip_marginals(self._h, _ptr(vals), _ptr(flat))):
        raise ArithmeticError("Marginals: the system is not positive definite")
    off = np.concatenate([[0], np.cumsum(dims.astype(np.int64) ** 2)])
    return [flat[off[c]:off[c + 1]].reshape(int(dims[c]), int(dims[c])).T.copy() for c in range(len(dims))]


CLinearSolver_HIP.Marginals = _block_diagonal_marginals


class CLinearSolver_Schur_HIP(_SolverBase):
    """Schur-complement solver for BA systems; same public surface as CLinearSolver_Schur
    (LinearSolver_Schur.h:1472-1553, 1623).  The guided ordering (cameras = wide block columns first,
    landmarks last; LinearSolver_Schur.cpp:771-838) is computed here from the block widths."""

    _Tag = "CBlockwiseLinearSolverTag"
    _mode = MODE_SCHUR

    def __init__(self, base_solver=None, device: int = 0, devices=None, **options):
        # the reference's constructor takes (and ignores) a base solver instance
        # (LinearSolver_Schur.h:1472-1474); accepted for signature parity
        super().__init__(device=device, devices=devices, **options)
        self._cut_override = None

    def _n_matrix_cut(self, lam) -> int:
        if self._cut_override is not None:
            return int(self._cut_override)
        if getattr(lam, "n_matrix_cut", 0):
            return int(lam.n_matrix_cut)
        dims = np.diff(lam.cumsum)
        wide = dims == dims.max()
        n_cut = int(np.count_nonzero(wide))
        if n_cut == 0 or n_cut == len(dims) or not np.all(wide[:n_cut]):
            # no landmark part (the reference hands such a system to its base solver, LinearSolver_Schur.h:1635-1638), or
            # cameras and landmarks interleaved (the C++ header permutes them; this mirror does not): cut 0 sends the
            # whole of Lambda through the sparse block path, which gives the same solution
            return 0
        return n_cut

    def SymbolicDecomposition_Blocky(self, lam, b_force_guided_ordering: bool = False) -> bool:
        return super().SymbolicDecomposition_Blocky(lam)

    def Set_Changed_Landmarks(self, landmark_indices) -> None:
        """The next Solve_PosDef_Blocky updates the reduced camera system of the previous one instead of rebuilding it
        (option ``schur_incremental=1``; the reference's dog-leg solver does this from Omega = delta Lambda,
        NonlinearSolver_Lambda_DL.h:2301-): ``landmark_indices`` = the landmarks (0-based among the landmark block
        columns) whose blocks changed since the previous call; camera blocks and eta may change freely."""
        idx = np.unique(np.asarray(landmark_indices, dtype=np.int64))
        self._check(self._lib.slampp_hip_schur_set_changed_points(self._h, _ptr(idx) if idx.size else None, int(idx.size)))

    def Solve_PosDef_Blocky_MarginalPoses(self, lam, eta: np.ndarray) -> bool:
        """LinearSolver_Schur.h:1956-2143: only the landmarks are solved for (dl = C^-1 eta_l), the pose part of
        ``eta`` is zeroed."""
        if eta.dtype != np.float64 or not eta.flags.c_contiguous or eta.shape != (lam.n_scalars,):
            raise ValueError("eta must be a contiguous float64 vector of the system's dimension")
        if not self._analyzed or self._structure_key != self._key(lam):
            self.SymbolicDecomposition_Blocky(lam, True)
        vals = np.ascontiguousarray(lam.values, dtype=np.float64)
        return self._check(self._lib.slampp_hip_solve_marginal_poses(self._h, _ptr(vals), _ptr(eta)))

    def Schur_Marginals(self, lam, b_do_cam_marginals: bool = True):
        """Block diagonal of the covariance Lambda^-1, as CSchurComplement_Marginals::Schur_Marginals returns it
        (BAMarginals.h:579-806): (camera blocks [nc, dc, dc] or None, landmark blocks [np, dp, dp]); the reduced
        camera system is assembled, factored and inverted on the device.  Raises if Lambda is not positive definite."""
        if not self._analyzed or self._structure_key != self._key(lam):
            self.SymbolicDecomposition_Blocky(lam, True)
        nc = self._n_matrix_cut(lam)
        dims = np.diff(lam.cumsum)
        dc, dp = int(dims[0]), int(dims[nc])
        cams = np.empty((nc, dc, dc), dtype=np.float64) if b_do_cam_marginals else None
        pts = np.empty((len(dims) - nc, dp, dp), dtype=np.float64)
        vals = np.ascontiguousarray(lam.values, dtype=np.float64)
        ok = self._check(self._lib.slampp_hip_schur_marginals(self._h, _ptr(vals), _ptr(cams) if cams is not None else None,
                                                              _ptr(pts)))
        if not ok:
            raise ArithmeticError("Schur_Marginals: the system is not positive definite")
        # blocks are column-major and symmetric: the C-order view is the same matrix
        return cams, pts

    def schur_marginals_device_async(self, values_ptr: int, cam_cov_ptr: int, point_cov_ptr: int) -> None:
        self._check(self._lib.slampp_hip_schur_marginals_device_async(self._h, values_ptr, cam_cov_ptr or None,
                                                                      point_cov_ptr or None))


class CLambdaAssembly_HIP:
    """Lambda and eta assembled on the device from per-edge Jacobians, written where the solver reads them.

    Stands where the reference computes every edge's Hessian blocks on the host and sums them with its
    reduction plan (BaseTypes_Binary.h:759-840, NonlinearSolver_Lambda_Base.h:1634-1688), for one homogeneous
    set of binary edges.  ``lam`` supplies the block structure (it must hold block (min, max) of every edge)."""

    def __init__(self, solver: _SolverBase, lam, v0: np.ndarray, v1: np.ndarray, n_residual_dim: int):
        self._solver = solver          # keeps the solver (and its stream) alive
        self._lib = solver._lib
        self._a = C.c_void_p()
        if not solver._analyzed or solver._structure_key != solver._key(lam):
            solver.SymbolicDecomposition_Blocky(lam)
        v0 = np.ascontiguousarray(v0, dtype=np.int64)
        v1 = np.ascontiguousarray(v1, dtype=np.int64)
        if v0.shape != v1.shape or v0.ndim != 1:
            raise ValueError("v0 and v1 must be vectors of one length")
        solver._check(self._lib.slampp_hip_assembly_create(solver._h, C.byref(self._a), v0.shape[0], _ptr(v0), _ptr(v1),
                                                           int(n_residual_dim)))

    def __del__(self):
        try:
            if getattr(self, "_a", None):
                self._lib.slampp_hip_assembly_destroy(self._a)
                self._a = None
        except Exception:
            pass

    def Refresh_Lambda_device(self, J0_ptr: int, J1_ptr: int, sigma_inv_ptr: int, error_ptr: int, weight_ptr: int,
                              values_ptr: int, eta_ptr: int, unary_vertex: int = 0, unary_factor=None,
                              unary_error=None, accumulate: bool = False) -> None:
        """Enqueues the assembly on the solver's stream (device pointers as integers; ``weight_ptr`` 0 = no
        robust weights).  ``unary_factor`` is the d x d factor U of the anchor (column-major) and
        ``unary_error`` its error vector, both host arrays or None."""
        uf = None if unary_factor is None else np.ascontiguousarray(unary_factor, dtype=np.float64)
        ue = None if unary_error is None else np.ascontiguousarray(unary_error, dtype=np.float64)
        self._solver._check(self._lib.slampp_hip_assemble_device_async(
            self._a, J0_ptr, J1_ptr, sigma_inv_ptr, error_ptr, weight_ptr or None, int(unary_vertex),
            None if uf is None else _ptr(uf), None if ue is None else _ptr(ue), values_ptr, eta_ptr, int(bool(accumulate))))


def Refresh_Lambda_sets_device(assemblies, pointer_sets, values_ptr: int, eta_ptr: int, unary_vertex: int = 0, unary_factor=None,
                               unary_error=None, accumulate: bool = False) -> None:
    """Every edge type of one Lambda in one call (slampp_hip_assemble_sets_device_async): ``assemblies`` a list of
    CLambdaAssembly_HIP of the same solver, ``pointer_sets`` their (J0, J1, sigma_inv, error, weight) device pointers."""
    sets = (EdgeSetPtrs * len(assemblies))()
    for i, (a, ptrs) in enumerate(zip(assemblies, pointer_sets)):
        sets[i].p_assembly = a._a
        sets[i].p_J0_dev, sets[i].p_J1_dev, sets[i].p_sigma_inv_dev, sets[i].p_error_dev = [int(p) for p in ptrs[:4]]
        sets[i].p_weight_dev = int(ptrs[4]) if len(ptrs) > 4 and ptrs[4] else None
    uf = None if unary_factor is None else np.ascontiguousarray(unary_factor, dtype=np.float64)
    ue = None if unary_error is None else np.ascontiguousarray(unary_error, dtype=np.float64)
    solver = assemblies[0]._solver
    solver._check(solver._lib.slampp_hip_assemble_sets_device_async(
        sets, len(assemblies), int(unary_vertex), None if uf is None else _ptr(uf), None if ue is None else _ptr(ue),
        values_ptr, eta_ptr, int(bool(accumulate))))
