"""GPU test of the thread-safety the reference documents: `Estimator` is "designed to be threadsafe" by cloning its config per call
(plspm/estimator.py:25,30-31); the C-ABI's counterpart is "re-entrant across handles" (SURVEY.md 8(b)).  libplspm_hip.so keeps process-global
state -- the caching allocator and stream pool (csrc/plspm_hip.hip), the record-download crew (csrc/plspm_bootstrap.hip) -- so two host threads
driving two handles on ONE device at the same time must get exactly the rows, iteration counts and summaries each gets alone.  ctypes releases
the GIL around every library call: the calls below really overlap."""
import threading

import numpy as np
import pytest

import plspm_oracle as orc
from test_gpu_parity import SCHEME_ID

pytestmark = pytest.mark.gpu


def _handle(X, blocks, C, modes, scheme, nonmetric=False):
    from plspm import _native
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    md = np.array([0 if m == "A" else 1 for m in modes], dtype=np.int32)
    nm = _native.NativeModel(boff, C.astype(np.uint8), md, SCHEME_ID[scheme], True, 100, 1e-6 if not nonmetric else 1e-7, 0, nonmetric=nonmetric)
    nm.upload(X)
    return nm


def _workloads():
    Xa, ba = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)                  # the headline model: int8 Gram + wave solver, sub-batched download
    Xb, bb = orc.synth(1500, orc.chain_C(5), 4, seed=7)                          # Mode B / centroid on the fp64 Gram route
    Xc, bc = orc.synth(3000, orc.satisfaction_C(), 5, seed=3)                    # Scale.NUM (host read-backs inside the call)
    return [dict(X=Xa, blocks=ba, C=orc.satisfaction_C(), modes="AAAAAA", scheme="path", B=5000, seed=11),
            dict(X=Xb, blocks=bb, C=orc.chain_C(5), modes="BBBBB", scheme="centroid", B=700, seed=12),
            dict(X=Xc, blocks=bc, C=orc.satisfaction_C(), modes="AAAAAA", scheme="factorial", B=1200, seed=13, nonmetric=True)]


def _one_pass(w):
    """What one Plspm(bootstrap=True) does on the device side, on a fresh handle: upload, fit, replicates to the host, replicates kept in HBM + summary."""
    nm = _handle(w["X"], w["blocks"], w["C"], w["modes"], w["scheme"], w.get("nonmetric", False))
    fit = nm.fit(want_scores=True)
    rows, status, iters = nm.bootstrap(w["B"], seed=w["seed"])
    nm.bootstrap_device(w["B"], seed=w["seed"])
    summ, used = nm.summary(w["B"], np.ones(nm.row_width))
    return dict(weights=fit["weights"], scores=fit["scores"], fit_iters=int(fit["iterations"]),
                rows=rows, status=status, iters=iters, summary=summ, used=int(used))


def _same(a, b, tag):
    for k in ("weights", "scores", "rows", "status", "iters", "summary"):
        assert np.array_equal(a[k], b[k], equal_nan=True), "%s: %s differs between the concurrent and the serial run" % (tag, k)
    assert a["fit_iters"] == b["fit_iters"] and a["used"] == b["used"], tag


def test_two_threads_two_handles_one_device_bit_identical_to_serial():
    work = _workloads()
    serial = [_one_pass(w) for w in work]
    for s, w in zip(serial, work):
        assert s["used"] > 0.9 * w["B"]
    rounds = 4
    results = {}
    errors = []
    start = threading.Barrier(len(work))

    def runner(i):
        try:
            start.wait()
            results[i] = [_one_pass(work[i]) for _ in range(rounds)]
        except Exception as exc:                               # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=runner, args=(i,)) for i in range(len(work))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(len(work)):
        for r in range(rounds):
            _same(results[i][r], serial[i], "workload %d round %d" % (i, r))


def test_two_threads_sharing_workload_shapes_and_pools():
    """Two threads on the SAME model shape (their handles draw same-size blocks from the allocator's size classes and streams from the same pool,
    handles created and destroyed while the other thread's kernels run)."""
    w = _workloads()[0]
    w = dict(w, B=2000)
    serial = _one_pass(w)
    out, errors = {}, []

    def runner(i):
        try:
            out[i] = [_one_pass(w) for _ in range(5)]
        except Exception as exc:                               # noqa: BLE001
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=runner, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in out:
        for r, res in enumerate(out[i]):
            _same(res, serial, "thread %d round %d" % (i, r))


def test_plspm_objects_from_two_threads():
    """The public API from two threads (the reference's documented use): Plspm(..., bootstrap=True) on different data sets concurrently gives each
    thread the frames it gets alone."""
    import pandas as pd
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme

    def make(seed, n):
        X, blocks = orc.synth(n, orc.satisfaction_C(), 4, seed=seed)
        cols = ["%s%d" % (lv.lower(), k) for lv in orc.SAT_LVS for k in range(4)]
        frame = pd.DataFrame(X, columns=cols)
        structure = c.Structure()
        for frm, to in orc.SAT_EDGES:
            structure.add_path([frm], [to])

        def run():
            cfg = c.Config(structure.path(), scaled=True)
            for lv in orc.SAT_LVS:
                cfg.add_lv_with_columns_named(lv, Mode.A, frame, lv.lower())
            m = Plspm(frame, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=600, processes=1, seed=seed)
            return m.outer_model(), m.inner_summary(), m.bootstrap().weights(), m.bootstrap().paths()
        return run

    runs = [make(1, 2000), make(2, 900)]
    serial = [r() for r in runs]
    got, errors = {}, []

    def runner(i):
        try:
            got[i] = [runs[i]() for _ in range(3)]
        except Exception as exc:                               # noqa: BLE001
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=runner, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for res in got[i]:
            for a, b in zip(res, serial[i]):
                pd.testing.assert_frame_equal(a, b, check_exact=True)


def test_categorical_and_hoc_models_from_three_threads():
    """The host-synchronising solvers under concurrency: an ordinal model (one-launch wave step + verification read-backs), a higher order construct on ordinal items (two
    handle pairs' worth of launches and read-backs per call) and a metric model, each Plspm(..., bootstrap=True) on its own thread at the same time, three rounds -- every
    thread gets the frames and replicates it gets alone."""
    import os
    import pandas as pd
    import plspm.config as c
    from helpers import GOLDEN
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    import fuzz_cases as fc
    mobi = pd.read_csv(os.path.join(GOLDEN, "ref_data", "mobi.csv"), index_col=0).astype(float)

    def hoc():
        structure = c.Structure()
        structure.add_path(["Expectation", "Quality"], ["Satisfaction"])
        structure.add_path(["Satisfaction"], ["Complaints", "Loyalty"])
        config = c.Config(structure.path(), default_scale=Scale.ORD)
        config.add_higher_order("Satisfaction", Mode.A, ["Image", "Value"])
        for lv, prefix in (("Expectation", "CUEX"), ("Quality", "PERQ"), ("Loyalty", "CUSL"), ("Image", "IMAG"), ("Complaints", "CUSCO"), ("Value", "PERV")):
            config.add_lv_with_columns_named(lv, Mode.A, mobi, prefix)
        return Plspm(mobi, config, Scheme.PATH, 100, 1e-7, bootstrap=True, bootstrap_iterations=400, seed=3)

    def frame_model(X, model, scale):
        lvs = ["L%d" % l for l in range(model.L)]
        df = pd.DataFrame(X, columns=["x%d" % p for p in range(X.shape[1])])

        def run():
            cfg = c.Config(pd.DataFrame(np.asarray(model.C, dtype=int), index=lvs, columns=lvs), scaled=True, default_scale=scale)
            for l in range(model.L):
                cfg.add_lv(lvs[l], Mode.A, *[c.MV("x%d" % p) for p in model.blocks[l]])
            return Plspm(df, cfg, Scheme.PATH, 100, 1e-6, bootstrap=True, bootstrap_iterations=600, seed=5)
        return run
    Xc, mc = fc.make_cat_big_case(12)                          # six LVs, 23 items of up to eight categories, all Mode A
    Xm, mm, _ = fc.make_case(8)
    runs = [hoc, frame_model(Xc, mc, Scale.ORD), frame_model(Xm, orc.Model(mm.blocks, mm.C, "A" * mm.L, "path", True), None)]

    def digest(m):
        b = m.bootstrap()
        return [m.outer_model().select_dtypes(include=[np.number]).values, m.path_coefficients().values, b.replicates(), b.status(), b.weights().values]
    serial = [digest(r()) for r in runs]
    got, errors = {}, []
    start = threading.Barrier(len(runs))

    def runner(i):
        try:
            start.wait()
            got[i] = [digest(runs[i]()) for _ in range(3)]
        except Exception as exc:                               # noqa: BLE001
            errors.append((i, repr(exc)))
    threads = [threading.Thread(target=runner, args=(i,)) for i in range(len(runs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in got:
        for r, res in enumerate(got[i]):
            for a, b in zip(res, serial[i]):
                assert np.array_equal(a, b, equal_nan=True), "thread %d round %d differs from the serial run" % (i, r)
