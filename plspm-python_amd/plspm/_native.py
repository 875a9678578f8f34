"""ctypes binding of libplspm_hip.so (C-ABI: include/plspm_hip.h).  No torch, no fallback.

The library is built in-tree by ``make -C plspm-python_amd/csrc`` (or ``__graft_entry__.build()``) into
``plspm/_lib/libplspm_hip.so``.  If it is missing, or no HIP device is visible, the estimator raises
``NativeBackendError`` -- there is deliberately no NumPy path behind this module.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLSPM_HIP_LIB", os.path.join(_HERE, "_lib", "libplspm_hip.so"))
ABI_VERSION = 1

STATUS_OK, STATUS_NOT_CONVERGED, STATUS_SINGULAR, STATUS_NONFINITE = 0, 1, 2, 3
KERNELS = {"resample": 0, "gram": 1, "solver": 2, "scores": 3, "pack": 4, "reduce": 5}
EXPORTS = ["plspm_abi_version", "plspm_device_count", "plspm_last_error", "plspm_model_create", "plspm_model_destroy", "plspm_model_set_nonmetric", "plspm_model_set_categorical", "plspm_model_set_missing", "plspm_model_attach_second_stage", "plspm_model_set_incomplete_rows",
           "plspm_upload", "plspm_effect_pairs", "plspm_row_width", "plspm_row_stride", "plspm_fit", "plspm_bootstrap", "plspm_bootstrap_device", "plspm_bootstrap_summary",
           "plspm_sync", "plspm_stream", "plspm_bootstrap_indices", "plspm_profile_enable", "plspm_profile_read", "plspm_profile_reset"]


class NativeBackendError(RuntimeError):
    pass


class _FitResult(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("weights", "loadings", "crossloadings", "path_coef", "r2", "lv_cov", "total", "direct",
                                               "indirect", "scores", "cov", "mean", "sign", "iterations", "status")]


_lib = None


def load():
    """Load the shared library once and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeBackendError("libplspm_hip.so not found at %s: build it with `make -C plspm-python_amd/csrc` "
                                 "(there is no CPU fallback)" % LIB_PATH)
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
        if categorical is not None:     # (mv_off, mv_kind): device columns are indicator-augmented, see plspm_model_set_categorical
            mv_off = np.ascontiguousarray(categorical[0], dtype=np.int32)
            mv_kind = np.ascontiguousarray(categorical[1], dtype=np.int32)
            self.P_out = len(mv_kind)
            self._check(lib.plspm_model_set_categorical(self._h, self.P_out, _ptr(mv_off), _ptr(mv_kind)), "plspm_model_set_categorical")
        if missing is not None:         # ind_of [P]: upload column of every incomplete data column's 0/1 missing indicator (else -1)
            ind_of = np.ascontiguousarray(missing, dtype=np.int32)
            self._check(lib.plspm_model_set_missing(self._h, int((ind_of >= 0).sum()), _ptr(ind_of)), "plspm_model_set_missing")
        self.n_eff = lib.plspm_effect_pairs(self._h, None, None)
        ef = np.zeros(max(self.n_eff, 1), dtype=np.int32)
        et = np.zeros(max(self.n_eff, 1), dtype=np.int32)
        lib.plspm_effect_pairs(self._h, _ptr(ef), _ptr(et))
        self.eff_from, self.eff_to = ef[:self.n_eff], et[:self.n_eff]
        self.row_width = lib.plspm_row_width(self._h)
        self.row_stride = lib.plspm_row_stride(self._h)     # device rows: [row | status | iterations]
        self.N = 0

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

    def fit(self, want_scores=True, want_cov=False):
        P, L, ne = self.P_out, self.L, self.n_eff
        out = dict(weights=np.empty(P), loadings=np.empty(P), crossloadings=np.empty((P, L)), path_coef=np.empty((L, L)), r2=np.empty(L),
                   lv_cov=np.empty((L, L)), total=np.empty(ne), direct=np.empty(ne), indirect=np.empty(ne),
                   scores=np.empty((self.N, L)) if want_scores else None, cov=np.empty((P, P)) if want_cov else None, mean=np.empty(P),
                   sign=np.empty(L, dtype=np.int8), iterations=np.zeros(1, dtype=np.int32), status=np.full(1, -1, dtype=np.int32))
        res = _FitResult(**{k: (v.ctypes.data if v is not None and v.size else None) for k, v in out.items()})
        self._check(self._lib.plspm_fit(self._h, ctypes.byref(res)), "plspm_fit")
        out["iterations"] = int(out["iterations"][0])
        out["status"] = int(out["status"][0])
        return out

    def bootstrap(self, B, seed=0, rep_offset=0, idx=None):
        """Returns (rows [B, R], status [B], iters [B]) on the host."""
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            if idx.shape != (B, self.N):
                raise ValueError("idx must have shape (B, N)")
        rows = np.empty((B, self.row_width))
        status = np.empty(B, dtype=np.int32)
        iters = np.empty(B, dtype=np.int32)
        self._check(self._lib.plspm_bootstrap(self._h, B, seed, rep_offset, _ptr(idx), _ptr(rows), _ptr(status), _ptr(iters)), "plspm_bootstrap")
        return rows, status, iters

    def bootstrap_device(self, B, seed=0, rep_offset=0):
        """Enqueue B replicates; returns raw device pointers (rows, status, iters) owned by the handle."""
        d_out, d_st, d_it = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.plspm_bootstrap_device(self._h, B, seed, rep_offset, None, ctypes.byref(d_out), ctypes.byref(d_st), ctypes.byref(d_it)),
                    "plspm_bootstrap_device")
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

    def stream_ptr(self):
        """The handle's hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        return int(self._lib.plspm_stream(self._h) or 0)

    def sync(self):
        self._check(self._lib.plspm_sync(self._h), "plspm_sync")

    def profile(self, on=True):
        self._lib.plspm_profile_enable(self._h, int(on))

    def profile_reset(self):
        self._lib.plspm_profile_reset(self._h)

    def profile_read(self, kernel):
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        self._check(self._lib.plspm_profile_read(self._h, KERNELS[kernel], ctypes.byref(ms), ctypes.byref(n)), "plspm_profile_read")
        return ms.value, n.value
