"""A/B timing of two builds of libplspm_hip.so on the headline workload (10k x 60 x 6, 5000 replicates), using only the
entry points every build has.  usage: python tools/ab_lib.py <lib.so> [<lib.so> ...]"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, "tools")
import synthetic as orc  # noqa: E402  (workload generator: data only)

vp, i32, i64, u64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
C = orc.satisfaction_C()
X, blocks = orc.synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
order = np.concatenate(blocks).astype(np.int32)
modes = np.zeros(6, dtype=np.int32)
path = np.ascontiguousarray(C.astype(np.uint8))
ptr = lambda a: a.ctypes.data_as(vp)
for lib_path in sys.argv[1:]:
    lib = ctypes.CDLL(lib_path)
    lib.plspm_model_create.restype = vp
    lib.plspm_model_create.argtypes = [i32, i32, vp, vp, vp, i32, i32, i32, dbl, i32]
    lib.plspm_upload.argtypes = [vp, vp, i64, i32, i32, vp]
    lib.plspm_bootstrap_device.argtypes = [vp, i64, u64, i64, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    lib.plspm_sync.argtypes = [vp]
    lib.plspm_profile_enable.argtypes = [vp, i32]
    lib.plspm_profile_read.argtypes = [vp, i32, ctypes.POINTER(dbl), ctypes.POINTER(i64)]
    lib.plspm_profile_reset.argtypes = [vp]
    h = lib.plspm_model_create(60, 6, ptr(boff), ptr(path), ptr(modes), 2, 1, 100, 1e-6, 0)
    assert lib.plspm_upload(h, ptr(X), 10000, 60, 0, ptr(order)) == 0
    a, b, c = vp(), vp(), vp()
    for _ in range(5):
        lib.plspm_bootstrap_device(h, 5000, 1, 0, None, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    lib.plspm_sync(h)
    t0 = time.perf_counter()
    for s in range(50):
        lib.plspm_bootstrap_device(h, 5000, 1, 5000 * s, None, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    lib.plspm_sync(h)
    wall = (time.perf_counter() - t0) / 50 * 1e3
    lib.plspm_profile_enable(h, 1)
    for s in range(20):
        lib.plspm_bootstrap_device(h, 5000, 1, 5000 * s, None, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    lib.plspm_sync(h)
    out = {}
    for name, k in (("resample", 0), ("gram", 1), ("solver", 2)):
        ms, n = dbl(0), i64(0)
        lib.plspm_profile_read(h, k, ctypes.byref(ms), ctypes.byref(n))
        out[name] = round(ms.value / max(n.value, 1), 4)
    print(lib_path.split("/")[-1], "ms/step %.4f" % wall, out, flush=True)
