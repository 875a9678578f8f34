"""ctypes binding of libplspm_hip.so (C-ABI: include/plspm_hip.h).  No torch, no fallback.

The library is built in-tree by ``make -C plspm-python_amd/csrc`` (or ``__graft_entry__.build()``) into
``plspm/_lib/libplspm_hip.so``.  If it is missing, or no HIP device is visible, the estimator raises
``NativeBackendError`` -- there is deliberately no NumPy path behind this module.
"""
import atexit
import ctypes
import os
import time
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLSPM_HIP_LIB", os.path.join(_HERE, "_lib", "libplspm_hip.so"))
ABI_VERSION = 4

STATUS_OK, STATUS_NOT_CONVERGED, STATUS_SINGULAR, STATUS_NONFINITE = 0, 1, 2, 3
KERNELS = {"resample": 0, "gram": 1, "solver": 2, "scores": 3, "pack": 4, "reduce": 5}
EXPORTS = ["plspm_abi_version", "plspm_device_count", "plspm_last_error", "plspm_model_create", "plspm_model_destroy", "plspm_model_set_nonmetric", "plspm_model_set_categorical", "plspm_model_set_missing", "plspm_model_attach_second_stage", "plspm_model_set_incomplete_rows",
           "plspm_upload", "plspm_effect_pairs", "plspm_row_width", "plspm_row_stride", "plspm_fit", "plspm_bootstrap", "plspm_bootstrap_device", "plspm_bootstrap_summary",
           "plspm_sync", "plspm_stream", "plspm_bootstrap_indices", "plspm_profile_enable", "plspm_profile_read", "plspm_profile_reset",
           "plspm_model_set_option", "plspm_model_get_option", "plspm_bootstrap_moments", "plspm_nonmetric_criteria", "plspm_bootstrap_fetch", "plspm_bootstrap_store", "plspm_rccl_unique_id", "plspm_comm_create", "plspm_comm_destroy",
           "plspm_comm_size", "plspm_comm_uses_rccl", "plspm_group_create", "plspm_group_destroy", "plspm_group_last_error", "plspm_group_size",
           "plspm_group_shard", "plspm_group_bootstrap", "plspm_group_sync", "plspm_group_records", "plspm_group_summary", "plspm_group_rows", "plspm_group_adopt", "plspm_bootstrap_prepare",
           "plspm_group_barrier", "plspm_group_max", "plspm_group_enqueue_times", "plspm_release_cached_memory",
           "plspm_op_inner_weights", "plspm_op_outer_weights", "plspm_op_outer_weights_nonmetric", "plspm_gram_tile_plan",
           "plspm_comm_create_ex", "plspm_comm_split", "plspm_comm_transport", "plspm_comm_max_channels", "plspm_group_set_option", "plspm_group_plan", "plspm_chunk_plan"]
UNIQUE_ID_BYTES = 128


class NativeBackendError(RuntimeError):
    pass


class _FitResult(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("weights", "loadings", "crossloadings", "path_coef", "r2", "lv_cov", "total", "direct",
                                               "indirect", "scores", "cov", "mean", "sign", "iterations", "status")]


_lib = None
# live native objects, closed in dependency order (groups -> communicators -> handles) by an atexit hook: at interpreter shutdown
# __del__ runs in arbitrary order and possibly after the HIP / RCCL runtimes have torn themselves down
_live_models, _live_comms, _live_groups = weakref.WeakSet(), weakref.WeakSet(), weakref.WeakSet()


def _shutdown():
    for pool in (_live_groups, _live_comms, _live_models):
        for obj in list(pool):
            try:
                obj.close()
            except Exception:
                pass


atexit.register(_shutdown)


def load():
    """Load the shared library once and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeBackendError("libplspm_hip.so not found at %s: build it with `make -C plspm-python_amd/csrc` "
                                 "(there is no CPU fallback)" % LIB_PATH)
    # multi-process RCCL / device-memory sharing on this platform needs dmabuf IPC (the host driver has no legacy IPC: without it
    # hipIpcGetMemHandle fails with "invalid argument" inside ncclCommInitRank); the HSA runtime reads the variable when it starts, i.e.
    # at the first HIP call behind this load -- a launcher that scrubbed the environment must not cost the job its collective
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, u64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
    lib.plspm_abi_version.restype = ctypes.c_int
    lib.plspm_device_count.restype = ctypes.c_int
    lib.plspm_last_error.restype = ctypes.c_char_p
    lib.plspm_last_error.argtypes = [vp]
    lib.plspm_model_create.restype = vp
    lib.plspm_model_create.argtypes = [i32, i32, vp, vp, vp, i32, i32, i32, dbl, i32]
    lib.plspm_model_destroy.restype = None
    lib.plspm_model_destroy.argtypes = [vp]
    lib.plspm_model_set_nonmetric.argtypes = [vp, i32]
    lib.plspm_model_set_categorical.argtypes = [vp, i32, vp, vp]
    lib.plspm_model_set_missing.argtypes = [vp, i32, vp]
    lib.plspm_model_attach_second_stage.argtypes = [vp, vp, vp]
    lib.plspm_model_set_incomplete_rows.argtypes = [vp, i32, vp, vp, i32]
    lib.plspm_upload.argtypes = [vp, vp, i64, i32, i32, vp]
    lib.plspm_effect_pairs.restype = i32
    lib.plspm_effect_pairs.argtypes = [vp, vp, vp]
    lib.plspm_row_width.restype = i32
    lib.plspm_row_width.argtypes = [vp]
    lib.plspm_row_stride.restype = i32
    lib.plspm_row_stride.argtypes = [vp]
    lib.plspm_fit.argtypes = [vp, ctypes.POINTER(_FitResult)]
    lib.plspm_bootstrap.argtypes = [vp, i64, u64, i64, vp, vp, vp, vp]
    lib.plspm_bootstrap_device.argtypes = [vp, i64, u64, i64, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    lib.plspm_bootstrap_summary.argtypes = [vp, vp, i64, i32, vp, vp, ctypes.POINTER(i64)]
    lib.plspm_sync.argtypes = [vp]
    lib.plspm_stream.restype = vp
    lib.plspm_stream.argtypes = [vp]
    lib.plspm_bootstrap_indices.argtypes = [u64, i64, i64, vp]
    lib.plspm_profile_enable.argtypes = [vp, i32]
    lib.plspm_profile_read.argtypes = [vp, i32, ctypes.POINTER(dbl), ctypes.POINTER(i64)]
    lib.plspm_profile_reset.argtypes = [vp]
    lib.plspm_model_set_option.argtypes = [vp, ctypes.c_char_p, i32]
    lib.plspm_model_get_option.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(i32)]
    lib.plspm_model_get_option.restype = ctypes.c_int
    lib.plspm_bootstrap_fetch.argtypes = [vp, i64, i64, vp, vp, vp]
    lib.plspm_bootstrap_moments.argtypes = [vp, i64, u64, i64, vp, vp]
    lib.plspm_nonmetric_criteria.argtypes = [vp, i64, vp]
    lib.plspm_bootstrap_store.argtypes = [vp, vp, i64]
    lib.plspm_bootstrap_prepare.argtypes = [vp]
    lib.plspm_rccl_unique_id.argtypes = [vp]
    lib.plspm_comm_create.restype = vp
    lib.plspm_comm_create.argtypes = [vp, i32, i32, i32, vp]
    lib.plspm_comm_create_ex.restype = vp
    lib.plspm_comm_create_ex.argtypes = [vp, i32, i32, i32, vp, i32, i32]
    lib.plspm_comm_split.restype = vp
    lib.plspm_comm_split.argtypes = [vp, i32]
    lib.plspm_comm_transport.restype = i32
    lib.plspm_comm_transport.argtypes = [vp]
    lib.plspm_comm_max_channels.restype = i32
    lib.plspm_comm_max_channels.argtypes = [vp]
    lib.plspm_group_set_option.argtypes = [vp, ctypes.c_char_p, i32]
    lib.plspm_group_plan.argtypes = [vp, i64, ctypes.POINTER(i32), vp, vp]
    lib.plspm_chunk_plan.argtypes = [i64, i64, i32, i32, i64, vp]
    lib.plspm_comm_destroy.restype = None
    lib.plspm_comm_destroy.argtypes = [vp]
    lib.plspm_comm_size.restype = i32
    lib.plspm_comm_size.argtypes = [vp]
    lib.plspm_comm_uses_rccl.restype = i32
    lib.plspm_comm_uses_rccl.argtypes = [vp]
    lib.plspm_group_create.restype = vp
    lib.plspm_group_create.argtypes = [vp, vp]
    lib.plspm_group_destroy.restype = None
    lib.plspm_group_destroy.argtypes = [vp]
    lib.plspm_group_last_error.restype = ctypes.c_char_p
    lib.plspm_group_last_error.argtypes = [vp]
    lib.plspm_group_size.restype = i32
    lib.plspm_group_size.argtypes = [vp]
    lib.plspm_group_shard.argtypes = [vp, i64, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.plspm_group_bootstrap.argtypes = [vp, i64, u64, i64]
    lib.plspm_group_sync.argtypes = [vp]
    lib.plspm_group_records.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i32)]
    lib.plspm_group_summary.argtypes = [vp, vp, vp, ctypes.POINTER(i64)]
    lib.plspm_group_rows.argtypes = [vp, vp, vp, vp]
    lib.plspm_group_adopt.argtypes = [vp]
    lib.plspm_group_barrier.argtypes = [vp]
    lib.plspm_group_max.argtypes = [vp, ctypes.POINTER(dbl)]
    lib.plspm_group_enqueue_times.argtypes = [vp, ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    lib.plspm_op_inner_weights.argtypes = [i32, i32, i32, vp, vp, i64, vp]
    lib.plspm_op_outer_weights.argtypes = [i32, i32, vp, vp, i64, i32, vp]
    lib.plspm_op_outer_weights_nonmetric.argtypes = [i32, i32, vp, vp, vp, i64, i32, dbl, vp, vp]
    if lib.plspm_abi_version() != ABI_VERSION:
        raise NativeBackendError("libplspm_hip.so ABI %d != expected %d" % (lib.plspm_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def device_count():
    return load().plspm_device_count()


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def bootstrap_indices(seed, rep, n):
    """Host mirror of the on-device resampling stream (Philox4x32-10 keyed by (seed, replicate))."""
    idx = np.empty(n, dtype=np.int32)
    rc = load().plspm_bootstrap_indices(seed, rep, n, _ptr(idx))
    if rc:
        raise NativeBackendError("plspm_bootstrap_indices failed (%d)" % rc)
    return idx


def i8_tile_plan(count_tiles, pair_tiles, cus=256, mix=True):
    """Host mirror of the int8 Gram's tile-row cut (plspm_gram_tile_plan): (launch of the 320-replicate kernel taken?, tall rows, short rows)."""
    tall, shrt = ctypes.c_int32(0), ctypes.c_int32(0)
    lib = load()
    lib.plspm_gram_tile_plan.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    rc = lib.plspm_gram_tile_plan(count_tiles, pair_tiles, cus, 1 if mix else 0, ctypes.byref(tall), ctypes.byref(shrt))
    if rc not in (0, 1):
        raise NativeBackendError("plspm_gram_tile_plan failed (%d)" % rc)
    return bool(rc), tall.value, shrt.value


def chunk_plan(B, bytes_per_unit, chunks=0, ratio_pct=60, align=0):
    """Host mirror of the sub-batch planner (plspm_chunk_plan): the sizes of the sub-batches ONE call of B units runs as."""
    lib = load()
    parts = np.zeros(8, dtype=np.int64)
    n = lib.plspm_chunk_plan(B, bytes_per_unit, chunks, ratio_pct, align, _ptr(parts))
    if n < 1:
        raise NativeBackendError("plspm_chunk_plan failed (%d)" % n)
    return [int(v) for v in parts[:n]]


class NativeModel:
    """One compiled model on one GPU (an opaque ``plspm_model_t*``)."""

    def __init__(self, block_offset, path, modes, scheme, scaled, max_iter, tol, device_id=0, nonmetric=False, categorical=None, missing=None):
        lib = load()
        if lib.plspm_device_count() <= 0:
            raise NativeBackendError("no HIP device visible: the MI355X backend has no CPU fallback")
        self._lib = lib
        self.block_offset = np.ascontiguousarray(block_offset, dtype=np.int32)
        self.L = len(self.block_offset) - 1
        self.P = int(self.block_offset[-1])
        path = np.ascontiguousarray(path, dtype=np.uint8)
        modes = np.ascontiguousarray(modes, dtype=np.int32)
        self._h = lib.plspm_model_create(self.P, self.L, _ptr(self.block_offset), _ptr(path), _ptr(modes), int(scheme), int(bool(scaled)),
                                         int(max_iter), float(tol), int(device_id))
        if not self._h:
            raise NativeBackendError("plspm_model_create: " + lib.plspm_last_error(None).decode())
        if nonmetric:
            self._check(lib.plspm_model_set_nonmetric(self._h, 1), "plspm_model_set_nonmetric")
        self.P_out = self.P             # per-MV outputs: logical MVs (== device columns unless categorical)
        self.n_upload_cols = self.P     # device columns of the resident matrix (+ missing indicators, set below)
        if categorical is not None:     # (mv_off, mv_kind): device columns are indicator-augmented, see plspm_model_set_categorical
            mv_off = np.ascontiguousarray(categorical[0], dtype=np.int32)
            mv_kind = np.ascontiguousarray(categorical[1], dtype=np.int32)
            self.P_out = len(mv_kind)
            self._check(lib.plspm_model_set_categorical(self._h, self.P_out, _ptr(mv_off), _ptr(mv_kind)), "plspm_model_set_categorical")
        if missing is not None:         # ind_of [P]: upload column of every incomplete data column's 0/1 missing indicator (else -1)
            ind_of = np.ascontiguousarray(missing, dtype=np.int32)
            self._check(lib.plspm_model_set_missing(self._h, int((ind_of >= 0).sum()), _ptr(ind_of)), "plspm_model_set_missing")
            self.n_upload_cols = self.P + int((ind_of >= 0).sum())
        self.n_eff = lib.plspm_effect_pairs(self._h, None, None)
        ef = np.zeros(max(self.n_eff, 1), dtype=np.int32)
        et = np.zeros(max(self.n_eff, 1), dtype=np.int32)
        lib.plspm_effect_pairs(self._h, _ptr(ef), _ptr(et))
        self.eff_from, self.eff_to = ef[:self.n_eff], et[:self.n_eff]
        self.row_width = lib.plspm_row_width(self._h)
        self.row_stride = lib.plspm_row_stride(self._h)     # device rows: [row | status | iterations]
        self.N = 0
        self.last_B = 0
        self.device_id = int(device_id)
        _live_models.add(self)

    def set_incomplete_rows(self, rows, present, raw_scale=False):
        """Non-metric data with missing values (plspm_model_set_incomplete_rows), after ``upload``: ``rows`` ascending row numbers,
        ``present`` [K, P] booleans in device column order."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        present = np.ascontiguousarray(present, dtype=np.uint8)
        if present.shape != (len(rows), self.P):
            raise ValueError("present must have shape (K, P)")
        self._check(self._lib.plspm_model_set_incomplete_rows(self._h, len(rows), _ptr(rows), _ptr(present), int(bool(raw_scale))),
                    "plspm_model_set_incomplete_rows")

    def attach_second_stage(self, second, lv_first):
        """Two-stage HOC bootstrap (plspm_model_attach_second_stage): ``second`` is the data-less stage-2 handle; afterwards
        this handle's bootstrap rows / effect pairs / row width are the second stage's."""
        lv_first = np.ascontiguousarray(lv_first, dtype=np.int32)
        self._check(self._lib.plspm_model_attach_second_stage(self._h, second._h, _ptr(lv_first)), "plspm_model_attach_second_stage")
        self._second = second                       # keeps the stage-2 handle alive as long as this one
        self.n_eff, self.eff_from, self.eff_to = second.n_eff, second.eff_from, second.eff_to
        self.row_width = self._lib.plspm_row_width(self._h)
        self.row_stride = self._lib.plspm_row_stride(self._h)

    def _check(self, rc, what):
        if rc:
            raise NativeBackendError("%s failed (%d): %s" % (what, rc, self._lib.plspm_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.plspm_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, X, col_index=None):
        """X: 2-D float64 array (C- or F-contiguous, used in place); col_index[p] = source column of device column p."""
        X = np.asarray(X)
        if X.dtype != np.float64 or X.ndim != 2:
            X = np.ascontiguousarray(X, dtype=np.float64)
        if X.flags.c_contiguous:
            layout = 0
        elif X.flags.f_contiguous:
            layout = 1
        else:
            X, layout = np.ascontiguousarray(X), 0
        ci = None if col_index is None else np.ascontiguousarray(col_index, dtype=np.int32)
        self._check(self._lib.plspm_upload(self._h, _ptr(X), X.shape[0], X.shape[1], layout, _ptr(ci)), "plspm_upload")
        self.N = X.shape[0]

    def fit(self, want_scores=True, want_cov=False, scores_out=None):
        """``scores_out``: a caller-owned [N, L] float64 C-contiguous buffer for the scores (as ``bootstrap(out=...)``: nothing is allocated -- or first
        touched -- per call; a fresh 160 MB array costs more in page faults than its copy over the link)."""
        P, L, ne = self.P_out, self.L, self.n_eff
        if scores_out is not None and (scores_out.shape != (self.N, L) or scores_out.dtype != np.float64 or not scores_out.flags.c_contiguous):
            raise ValueError("scores_out = [N, L] float64, C-contiguous")
        out = dict(weights=np.empty(P), loadings=np.empty(P), crossloadings=np.empty((P, L)), path_coef=np.empty((L, L)), r2=np.empty(L),
                   lv_cov=np.empty((L, L)), total=np.empty(ne), direct=np.empty(ne), indirect=np.empty(ne),
                   scores=(scores_out if scores_out is not None else np.empty((self.N, L))) if want_scores else None, cov=np.empty((P, P)) if want_cov else None, mean=np.empty(P),
                   sign=np.empty(L, dtype=np.int8), iterations=np.zeros(1, dtype=np.int32), status=np.full(1, -1, dtype=np.int32))
        res = _FitResult(**{k: (v.ctypes.data if v is not None and v.size else None) for k, v in out.items()})
        self._check(self._lib.plspm_fit(self._h, ctypes.byref(res)), "plspm_fit")
        out["iterations"] = int(out["iterations"][0])
        out["status"] = int(out["status"][0])
        return out

    def bootstrap(self, B, seed=0, rep_offset=0, idx=None, out=None):
        """Returns (rows [B, R], status [B], iters [B]) on the host."""
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            if idx.shape != (B, self.N):
                raise ValueError("idx must have shape (B, N)")
        if out is None:
            rows, status, iters = np.empty((B, self.row_width)), np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
        else:                                      # caller-owned buffers (the C-ABI's contract): nothing is allocated -- or first touched -- per call
            rows, status, iters = out
            if rows.shape != (B, self.row_width) or rows.dtype != np.float64 or not rows.flags.c_contiguous or status.shape != (B,) or iters.shape != (B,) \
                    or status.dtype != np.int32 or iters.dtype != np.int32:
                raise ValueError("out = (rows [B, row_width] float64, status [B] int32, iters [B] int32), C-contiguous")
        self._check(self._lib.plspm_bootstrap(self._h, B, seed, rep_offset, _ptr(idx), _ptr(rows), _ptr(status), _ptr(iters)), "plspm_bootstrap")
        self.last_B = B
        return rows, status, iters

    def bootstrap_moments(self, B, seed=0, rep_offset=0, idx=None):
        """Test seam (plspm_bootstrap_moments): the replicates' moment matrices [B, C, C] of the uploaded columns + the ones column,
        from the Gram path the ``gram_path`` option selects; no solver."""
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            if idx.shape != (B, self.N):
                raise ValueError("idx must have shape (B, N)")
        C = self.n_upload_cols + 1
        out = np.empty((B, C, C))
        self._check(self._lib.plspm_bootstrap_moments(self._h, B, seed, rep_offset, _ptr(idx), _ptr(out)), "plspm_bootstrap_moments")
        return out

    def nonmetric_criteria(self, B):
        """Test seam (plspm_nonmetric_criteria): the stop-rule value each of the first B problems of the last non-metric run was decided on."""
        out = np.empty(B)
        self._check(self._lib.plspm_nonmetric_criteria(self._h, B, _ptr(out)), "plspm_nonmetric_criteria")
        return out

    def prepare_bootstrap(self):
        """Enqueue the per-data-set preparation of a later bootstrap (digit planes of the int8 Gram) beside whatever runs next."""
        self._check(self._lib.plspm_bootstrap_prepare(self._h), "plspm_bootstrap_prepare")

    def bootstrap_device(self, B, seed=0, rep_offset=0):
        """Enqueue B replicates; returns raw device pointers (rows, status, iters) owned by the handle."""
        d_out, d_st, d_it = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.plspm_bootstrap_device(self._h, B, seed, rep_offset, None, ctypes.byref(d_out), ctypes.byref(d_st), ctypes.byref(d_it)),
                    "plspm_bootstrap_device")
        self.last_B = B
        return d_out.value, d_st.value, d_it.value

    def summary(self, B, original, d_rows=None, stride=0):
        """Device-side _create_summary of the last bootstrap on this handle (or of the device records at ``d_rows``).
        Returns ([R, 6] array: original, mean, std.error, perc.025, perc.975, t stat.; number of OK replicates)."""
        original = np.ascontiguousarray(original, dtype=np.float64)
        if original.shape != (self.row_width,):
            raise ValueError("original must have row_width entries")
        out = np.empty((self.row_width, 6))
        used = ctypes.c_int64(0)
        self._check(self._lib.plspm_bootstrap_summary(self._h, d_rows, B, stride, _ptr(original), _ptr(out), ctypes.byref(used)), "plspm_bootstrap_summary")
        return out, used.value

    def fetch(self, first=0, count=None):
        """Host copy of replicates [first, first + count) of the LAST bootstrap / bootstrap_device / store on this handle (their
        records are still in HBM): (rows [count, R], status, iters)."""
        if count is None:
            count = self.last_B - first
        rows = np.empty((count, self.row_width))
        status = np.empty(count, dtype=np.int32)
        iters = np.empty(count, dtype=np.int32)
        self._check(self._lib.plspm_bootstrap_fetch(self._h, first, count, _ptr(rows), _ptr(status), _ptr(iters)), "plspm_bootstrap_fetch")
        return rows, status, iters

    def store(self, records):
        """Put merged host records [B, row_stride] (device layout: row | status | iterations) into the handle, e.g. shards that a
        host-side transport gathered; ``summary`` / ``fetch`` then work on them."""
        records = np.ascontiguousarray(records, dtype=np.float64)
        if records.ndim != 2 or records.shape[1] != self.row_stride:
            raise ValueError("records must have shape (B, row_stride)")
        self._check(self._lib.plspm_bootstrap_store(self._h, _ptr(records), records.shape[0]), "plspm_bootstrap_store")
        self.last_B = records.shape[0]

    def set_option(self, key, value):
        """Launch-geometry option of the handle (include/plspm_hip.h, plspm_model_set_option)."""
        self._check(self._lib.plspm_model_set_option(self._h, key.encode(), int(value)), "plspm_model_set_option")

    def get_option(self, key):
        """Current value of an option; ``"last_gram_path"``: 1 = fp64 MFMA Gram, 2 = int8 digit-plane Gram (last bootstrap call)."""
        v = ctypes.c_int32(0)
        self._check(self._lib.plspm_model_get_option(self._h, key.encode(), ctypes.byref(v)), "plspm_model_get_option")
        return v.value

    def stream_ptr(self):
        """The handle's hipStream_t as an integer."""
        return int(self._lib.plspm_stream(self._h) or 0)

    def sync(self):
        self._check(self._lib.plspm_sync(self._h), "plspm_sync")

    def profile(self, on=True, only=None):
        """HIP-event timing of the handle's kernels: every kernel, or (``only="gram"`` ...) just one of them."""
        self._lib.plspm_profile_enable(self._h, (2 + KERNELS[only]) if (on and only is not None) else int(bool(on)))

    def profile_reset(self):
        self._lib.plspm_profile_reset(self._h)

    def profile_read(self, kernel):
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        self._check(self._lib.plspm_profile_read(self._h, KERNELS[kernel], ctypes.byref(ms), ctypes.byref(n)), "plspm_profile_read")
        return ms.value, n.value


def op_inner_weights(scheme_code, path, y, device_id=0):
    """Scheme operator on the device (plspm_op_inner_weights): path [L, L] 0/1, y [N, L] scores -> E [L, L]."""
    lib = load()
    path = np.ascontiguousarray(path, dtype=np.uint8)
    y = np.ascontiguousarray(y, dtype=np.float64)
    L = path.shape[0]
    if path.shape != (L, L) or y.ndim != 2 or y.shape[1] != L:
        raise ValueError("path must be L x L and y N x L")
    E = np.empty((L, L))
    rc = lib.plspm_op_inner_weights(int(device_id), int(scheme_code), L, _ptr(path), _ptr(y), y.shape[0], _ptr(E))
    if rc:
        raise NativeBackendError("plspm_op_inner_weights failed (%d): %s" % (rc, lib.plspm_last_error(None).decode()))
    return E


def op_outer_weights(mode_code, Xk, z, device_id=0):
    """Mode operator on the device (plspm_op_outer_weights): Xk [N, k], z [N] -> w [k]."""
    lib = load()
    Xk = np.ascontiguousarray(Xk, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1)
    if Xk.ndim != 2 or z.shape[0] != Xk.shape[0]:
        raise ValueError("Xk must be N x k and z of length N")
    w = np.empty(Xk.shape[1])
    rc = lib.plspm_op_outer_weights(int(device_id), int(mode_code), _ptr(Xk), _ptr(z), Xk.shape[0], Xk.shape[1], _ptr(w))
    if rc:
        raise NativeBackendError("plspm_op_outer_weights failed (%d): %s" % (rc, lib.plspm_last_error(None).decode()))
    return w


def op_outer_weights_nonmetric(mode_code, Xk, present, z, correction, device_id=0):
    """Non-metric Mode operator on the device (plspm_op_outer_weights_nonmetric): Xk [N, k] (NaN = missing), present [N, k] 0/1 or None,
    z [N] -> (w [k], Y [N])."""
    lib = load()
    Xk = np.ascontiguousarray(Xk, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    if Xk.ndim != 2 or z.shape != (Xk.shape[0],):
        raise ValueError("Xk must be [N, k] and z [N]")
    mask = None
    if present is not None:
        mask = np.ascontiguousarray(np.asarray(present) != 0, dtype=np.uint8)
        if mask.shape != Xk.shape:
            raise ValueError("present must have the shape of Xk")
    w, Y = np.empty(Xk.shape[1]), np.empty(Xk.shape[0])
    rc = lib.plspm_op_outer_weights_nonmetric(int(device_id), int(mode_code), _ptr(Xk), _ptr(mask), _ptr(z), Xk.shape[0], Xk.shape[1], float(correction), _ptr(w), _ptr(Y))
    if rc:
        raise NativeBackendError("plspm_op_outer_weights_nonmetric failed (%d): %s" % (rc, lib.plspm_last_error(None).decode()))
    return w, Y


def release_cached_memory():
    """Hand the library's cached device / pinned-host blocks back to the HIP runtime."""
    load().plspm_release_cached_memory()


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 of a one-process-per-GPU job hands them to every rank)."""
    lib = load()
    buf = (ctypes.c_uint8 * UNIQUE_ID_BYTES)()
    rc = lib.plspm_rccl_unique_id(buf)
    if rc:
        raise NativeBackendError("plspm_rccl_unique_id failed (%d): %s" % (rc, lib.plspm_group_last_error(None).decode()))
    return bytes(buf)


class NativeComm:
    """The RCCL communicators of this process's ranks (an opaque ``plspm_comm_t*``): every rank of a single-process multi-GPU job
    (``NativeComm(devices)``) or one rank of a one-process-per-GPU job (``NativeComm([device], nranks, rank, unique_id)``).
    Expensive to create (librccl load + ncclCommInit*); create once, bind to groups one after the other."""

    TRANSPORTS = {"auto": 0, "rccl": 1, "copy": 2}

    def __init__(self, devices, nranks=None, first_rank=0, unique_id=None, transport="auto", max_channels=0, _handle=None):
        """``transport``: "auto" (RCCL between distinct devices), "rccl", or "copy" (single-process jobs: device-to-device copies on
        peer-mapped buffers -- the SDMA engines move the records, no kernel of the exchange takes a CU).  ``max_channels`` > 0 caps the
        workgroups RCCL runs for this communicator's collectives (ncclConfig_t.maxCTAs)."""
        if _handle is None and transport not in self.TRANSPORTS:
            raise ValueError("transport must be one of %s (got %r; PLSPM_TRANSPORT sets the default)" % (", ".join(sorted(self.TRANSPORTS)), transport))
        lib = load()
        self._lib = lib
        self.devices = [int(d) for d in devices]
        self.nranks = len(self.devices) if nranks is None else int(nranks)
        self.first_rank = int(first_rank)
        if _handle is not None:
            self._h = _handle
        else:
            dev = np.ascontiguousarray(self.devices, dtype=np.int32)
            uid = None
            if unique_id is not None:
                if len(unique_id) != UNIQUE_ID_BYTES:
                    raise ValueError("unique_id must have %d bytes" % UNIQUE_ID_BYTES)
                uid = (ctypes.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
            t0 = time.perf_counter()
            self._h = lib.plspm_comm_create_ex(_ptr(dev), len(self.devices), self.nranks, self.first_rank, uid, self.TRANSPORTS[transport], int(max_channels))
            self.create_s = time.perf_counter() - t0
            if not self._h:
                raise NativeBackendError("plspm_comm_create: " + lib.plspm_group_last_error(None).decode())
        self.uses_rccl = bool(lib.plspm_comm_uses_rccl(self._h))
        self.transport = {1: "rccl", 2: "copy-engines", 3: "device-copies"}.get(lib.plspm_comm_transport(self._h), "?")
        self.max_channels = int(lib.plspm_comm_max_channels(self._h))
        self._bound = None              # weak reference to the group this communicator currently serves (one at a time)
        _live_comms.add(self)

    def split(self, max_channels):
        """A second communicator over the same ranks with its own channel cap (ncclCommSplit: collective over this one, every rank calls it)."""
        t0 = time.perf_counter()
        h = self._lib.plspm_comm_split(self._h, int(max_channels))
        if not h:
            raise NativeBackendError("plspm_comm_split: " + self._lib.plspm_group_last_error(None).decode())
        twin = NativeComm(self.devices, self.nranks, self.first_rank, _handle=h)
        twin.create_s = time.perf_counter() - t0
        return twin

    def busy(self):
        """True while a live group is bound to this communicator (plspm_group_create refuses a second one)."""
        g = self._bound() if self._bound is not None else None
        return g is not None and bool(getattr(g, "_h", None))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.plspm_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeGroup:
    """Replicate shards over the handles of a communicator + ONE all-gather (an opaque ``plspm_group_t*``).  ``models[i]`` lives on
    ``comm.devices[i]`` and holds the same model and data as every other handle of the job."""

    def __init__(self, comm, models):
        lib = load()
        self._lib = lib
        self.comm, self.models = comm, list(models)
        if len(self.models) != len(comm.devices):
            raise ValueError("one handle per local rank of the communicator")
        arr = (ctypes.c_void_p * len(self.models))(*[m._h for m in self.models])
        self._h = lib.plspm_group_create(comm._h, arr)
        if not self._h:
            raise NativeBackendError("plspm_group_create: " + lib.plspm_group_last_error(None).decode())
        self.nranks = lib.plspm_group_size(self._h)
        self.first_rank = int(comm.first_rank)        # rank of local handle 0 (0: this process holds the rank that summarises)
        self.row_width, self.row_stride = self.models[0].row_width, self.models[0].row_stride
        self.last_B = 0
        comm._bound = weakref.ref(self)
        _live_groups.add(self)

    def _check(self, rc, what):
        if rc:
            raise NativeBackendError("%s failed (%d): %s" % (what, rc, self._lib.plspm_group_last_error(self._h).decode()))

    def close(self):
        # (a communicator that was destroyed first only RELEASED this group -- streams, buffers, its hold on the handles --; the
        # struct is still ours to free, and calls on it in between report PLSPM_E_STATE)
        if getattr(self, "_h", None):
            self._lib.plspm_group_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, B, rank):
        first, count = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._lib.plspm_group_shard(self._h, B, rank, ctypes.byref(first), ctypes.byref(count)), "plspm_group_shard")
        return first.value, count.value

    def set_option(self, key, value):
        """"chunks" 0 automatic | 1 .. 8 sub-batches per call; "chunk_ratio" percent (include/plspm_hip.h, plspm_group_bootstrap)."""
        self._check(self._lib.plspm_group_set_option(self._h, key.encode(), int(value)), "plspm_group_set_option")

    def plan(self, B):
        """The sub-batches a call of B replicates runs as: [(first, count), ...] in replicate-id order."""
        n = ctypes.c_int32(0)
        first, count = np.zeros(8, dtype=np.int64), np.zeros(8, dtype=np.int64)
        self._check(self._lib.plspm_group_plan(self._h, B, ctypes.byref(n), _ptr(first), _ptr(count)), "plspm_group_plan")
        return [(int(first[k]), int(count[k])) for k in range(n.value)]

    def bootstrap(self, B, seed=0, rep_offset=0):
        """Enqueue B replicates over the group + the all-gather (no host synchronisation for metric models)."""
        self._check(self._lib.plspm_group_bootstrap(self._h, B, seed, rep_offset), "plspm_group_bootstrap")
        self.last_B = B

    def sync(self):
        self._check(self._lib.plspm_group_sync(self._h), "plspm_group_sync")

    def barrier(self):
        self._check(self._lib.plspm_group_barrier(self._h), "plspm_group_barrier")

    def max(self, value):
        v = ctypes.c_double(float(value))
        self._check(self._lib.plspm_group_max(self._h, ctypes.byref(v)), "plspm_group_max")
        return v.value

    def enqueue_times(self):
        """(shards_ms, exchange_ms): host time of the last bootstrap call's two enqueue phases on this process."""
        a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._check(self._lib.plspm_group_enqueue_times(self._h, ctypes.byref(a), ctypes.byref(b)), "plspm_group_enqueue_times")
        return a.value, b.value

    def records(self, local=0):
        """(device pointer, n_records, stride) of the gathered records of the last bootstrap on local handle ``local``."""
        ptr, n, stride = ctypes.c_void_p(), ctypes.c_int64(0), ctypes.c_int32(0)
        self._check(self._lib.plspm_group_records(self._h, local, ctypes.byref(ptr), ctypes.byref(n), ctypes.byref(stride)), "plspm_group_records")
        return ptr.value, n.value, stride.value

    def summary(self, original):
        original = np.ascontiguousarray(original, dtype=np.float64)
        if original.shape != (self.row_width,):
            raise ValueError("original must have row_width entries")
        out = np.empty((self.row_width, 6))
        used = ctypes.c_int64(0)
        self._check(self._lib.plspm_group_summary(self._h, _ptr(original), _ptr(out), ctypes.byref(used)), "plspm_group_summary")
        return out, used.value

    def adopt(self):
        """Move the last bootstrap's records into ``models[0]`` (device-to-device); the group may be closed afterwards."""
        self._check(self._lib.plspm_group_adopt(self._h), "plspm_group_adopt")

    def rows(self):
        """(rows [B, R], status, iters) of the last bootstrap in replicate-id order, on the host."""
        B = self.last_B
        rows = np.empty((B, self.row_width))
        status = np.empty(B, dtype=np.int32)
        iters = np.empty(B, dtype=np.int32)
        self._check(self._lib.plspm_group_rows(self._h, _ptr(rows), _ptr(status), _ptr(iters)), "plspm_group_rows")
        return rows, status, iters
