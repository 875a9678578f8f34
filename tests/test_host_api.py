"""CPU tests of the host side: the drop-in model-specification API (mirrors reference tests/test_config.py and
tests/test_estimator-free parts), the model compiler, the C-ABI surface of libplspm_hip.so (loads, exports every symbol
include/plspm_hip.h declares -- no compute calls without a GPU), and that the product fails loudly without a GPU."""
import os
import re

import numpy as np
import pandas as pd
import pandas.testing as pdt
import pytest

import plspm.config as c
from plspm import _native
from plspm._compile import compile_model
from plspm.bootstrap import _create_summary
from plspm.mode import Mode
from plspm.scale import Scale
from plspm.scheme import Scheme
from helpers import GOLDEN, load, satisfaction_frame, SAT_PREFIX

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def russa():
    return pd.read_csv(os.path.join(GOLDEN, "ref_data", "russa.csv"), index_col=0)


def russa_path():
    lvs = ["AGRI", "IND", "POLINS"]
    return pd.DataFrame([[0, 0, 0], [0, 0, 0], [1, 1, 0]], index=lvs, columns=lvs)


# ------------------------------------------------------------------ Config / Structure (reference tests/test_config.py)
def test_config_rejects_bad_path_matrix():
    with pytest.raises(TypeError):
        c.Config("hello")
    with pytest.raises(ValueError):
        c.Config(pd.DataFrame([[0, 0, 0]]))
    with pytest.raises(ValueError):
        c.Config(pd.DataFrame([[1, 1], [1, 1]]))
    with pytest.raises(ValueError):
        c.Config(pd.DataFrame([[1, 0], [2, 1]]))
    with pytest.raises(ValueError):
        c.Config(pd.DataFrame([[1, 0], [1, 1]], index=["A", "B"], columns=["C", "D"]))


def test_config_lv_and_mv_consistency():
    config = c.Config(russa_path())
    with pytest.raises(ValueError):
        config.add_lv("POO", Mode.A, c.MV("test"))
    config.add_lv("AGRI", Mode.A, c.MV("gini"), c.MV("farm"), c.MV("rent"))
    assert config.mode("AGRI") == Mode.A and config.mvs("AGRI") == ["gini", "farm", "rent"]
    config.add_lv("IND", Mode.A, c.MV("gnpr"))
    with pytest.raises(ValueError):                      # POLINS missing
        config.filter(russa())
    bad = c.Config(russa_path())
    bad.add_lv("AGRI", Mode.A, c.MV("gini"), c.MV("farm"), c.MV("poo"))
    bad.add_lv("IND", Mode.A, c.MV("gnpr")); bad.add_lv("POLINS", Mode.A, c.MV("ecks"))
    with pytest.raises(ValueError):                      # column not in the data
        bad.filter(russa())


def test_config_filters_columns_in_add_lv_order_and_rejects_non_numeric():
    config = c.Config(russa_path())
    config.add_lv("POLINS", Mode.A)
    config.add_lv("AGRI", Mode.A, c.MV("gini"), c.MV("farm"), c.MV("rent"))
    config.add_lv("IND", Mode.A)
    assert list(config.filter(russa())) == ["gini", "farm", "rent"]
    data = russa(); data["gini"] = data["gini"].astype(str)
    with pytest.raises(ValueError):
        config.filter(data)


def test_scale_promotion_rules():
    def full(default, **kw):
        cfg = c.Config(russa_path(), default_scale=default, **kw)
        return cfg
    cfg = full(None)
    cfg.add_lv("POLINS", Mode.A, c.MV("ecks"), c.MV("death"), c.MV("demo"), c.MV("inst"))
    cfg.add_lv("AGRI", Mode.A, c.MV("gini", Scale.NUM), c.MV("farm"), c.MV("rent"))
    cfg.add_lv("IND", Mode.A, c.MV("gnpr"), c.MV("labo"))
    with pytest.raises(TypeError):
        cfg.treat(cfg.filter(russa()))
    cfg = full(Scale.RAW)
    cfg.add_lv("POLINS", Mode.A, c.MV("ecks"), c.MV("death"), c.MV("demo"), c.MV("inst"))
    cfg.add_lv("AGRI", Mode.A, c.MV("gini"), c.MV("farm"), c.MV("rent"))
    cfg.add_lv("IND", Mode.A, c.MV("gnpr"), c.MV("labo"))
    cfg.treat(cfg.filter(russa()))
    assert not cfg.scaled()
    cfg = full(Scale.RAW)
    cfg.add_lv("POLINS", Mode.A, c.MV("ecks", Scale.NUM), c.MV("death"), c.MV("demo"), c.MV("inst"))
    cfg.add_lv("AGRI", Mode.A, c.MV("gini"), c.MV("farm"), c.MV("rent"))
    cfg.add_lv("IND", Mode.A, c.MV("gnpr"), c.MV("labo"))
    cfg.treat(cfg.filter(russa()))
    assert cfg.scaled() and all(cfg.scale(mv) == Scale.NUM for mv in ["gini", "farm", "rent"])
    cfg = full(Scale.RAW, scaled=False)
    cfg.add_lv("AGRI", Mode.A, c.MV("gini", Scale.NUM), c.MV("farm", Scale.ORD), c.MV("rent"))
    cfg.add_lv("IND", Mode.A, c.MV("gnpr"), c.MV("labo"))
    cfg.add_lv("POLINS", Mode.A, c.MV("ecks"), c.MV("death"), c.MV("demo"), c.MV("inst"))
    cfg.filter(russa())
    assert not cfg.scaled() and cfg.scale("farm") == Scale.ORD and not cfg.metric()


def test_structure_topological_order_matches_reference():
    lvs = ["MANDRILL", "BONOBO", "APE", "GOAT", "CATFISH"]
    expected = pd.DataFrame([[0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [1, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 1, 0, 0]], index=lvs, columns=lvs)
    s = c.Structure()
    s.add_path(source=["BONOBO", "MANDRILL"], target=["APE"])
    s.add_path(source=["APE"], target=["CATFISH", "GOAT"])
    pdt.assert_frame_equal(expected, s.path())
    pdt.assert_frame_equal(expected, s.path())                  # repeatable (the reference's second call raises)
    rebuilt = c.Structure(expected).path()                      # same edges; order follows the row-major edge list
    assert sorted(rebuilt.index) == sorted(lvs)
    pdt.assert_frame_equal(rebuilt.loc[lvs, lvs], expected)
    with pytest.raises(ValueError):
        c.Structure().add_path(["A", "B"], ["C", "D"])
    cyc = c.Structure(); cyc.add_path(["A"], ["B"]); cyc.add_path(["B"], ["A"])
    with pytest.raises(ValueError):
        cyc.path()


def test_satisfaction_structure_order():
    s = c.Structure()
    s.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); s.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    s.add_path(["QUAL"], ["VAL", "SAT"]); s.add_path(["VAL"], ["SAT"]); s.add_path(["SAT"], ["LOY"])
    import plspm_oracle as orc
    p = s.path()
    assert list(p) == orc.SAT_LVS and np.array_equal(p.values, orc.satisfaction_C())


def test_duplicate_names_rejected():
    s = c.Structure(); s.add_path(source=["BONOBO"], target=["APE"])
    cfg = c.Config(s.path())
    cfg.add_lv("BONOBO", Mode.A, c.MV("a"), c.MV("b"))
    with pytest.raises(ValueError):
        cfg.add_lv("APE", Mode.A, c.MV("a"), c.MV("b"))
    with pytest.raises(ValueError):
        c.Config(s.path()).add_lv("BONOBO", Mode.A, c.MV("BONOBO"))


def test_metric_treat_is_one_global_scalar():
    sat = satisfaction_frame()
    s = c.Structure(); s.add_path(["IMAG"], ["EXPE"])
    cfg = c.Config(s.path(), scaled=True)
    cfg.add_lv_with_columns_named("IMAG", Mode.A, sat, "imag"); cfg.add_lv_with_columns_named("EXPE", Mode.A, sat, "expe")
    f = cfg.filter(sat)
    t = cfg.treat(f)
    n = f.shape[0]
    g = np.std(f.values.reshape(-1), ddof=1) * np.sqrt((n - 1) / n)
    np.testing.assert_allclose(t.values, (f.values - f.values.mean(axis=0)) / g, rtol=1e-13)
    assert t.std().nunique() > 1                                   # columns keep different spreads


# ------------------------------------------------------------------ model compiler
def test_compile_model_orders_and_maps():
    sat = satisfaction_frame()
    s = c.Structure()
    s.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); s.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    s.add_path(["QUAL"], ["VAL", "SAT"]); s.add_path(["VAL"], ["SAT"]); s.add_path(["SAT"], ["LOY"])
    cfg = c.Config(s.path(), scaled=False)
    for lv, mode in (("IMAG", Mode.A), ("EXPE", Mode.B), ("VAL", Mode.A), ("QUAL", Mode.B), ("SAT", Mode.A), ("LOY", Mode.A)):
        cfg.add_lv_with_columns_named(lv, mode, sat, SAT_PREFIX[lv])
    f = cfg.filter(sat)
    cm = compile_model(cfg, cfg.path(), list(f.columns))
    assert cm.lvs == ["IMAG", "EXPE", "QUAL", "VAL", "SAT", "LOY"]
    assert cm.dev_mvs[:5] == ["imag%d" % i for i in range(1, 6)] and cm.dev_mvs[10:15] == ["qual%d" % i for i in range(1, 6)]
    assert list(f.columns)[10:14] == ["val%d" % i for i in range(1, 5)]            # data order is add_lv order (VAL before QUAL)
    assert [list(f.columns)[i] for i in cm.col_index] == cm.dev_mvs
    assert [cm.dev_mvs[i] for i in cm.inv_index] == list(f.columns)
    assert cm.block_offset.tolist() == [0, 5, 10, 15, 19, 23, 27] and cm.modes.tolist() == [0, 1, 1, 0, 0, 0]
    assert cm.endogenous() == ["EXPE", "QUAL", "VAL", "SAT", "LOY"]
    assert Scheme.CENTROID.value.code == 0 and Scheme.FACTORIAL.value.code == 1 and Scheme.PATH.value.code == 2


# ------------------------------------------------------------------ summaries (host side of the bootstrap)
def test_create_summary_matches_reference_golden():
    g = load("g6_summary")
    frame = _create_summary(pd.DataFrame(g["samples"], columns=list("abcde")), pd.Series(g["original"], index=list("abcde")))
    assert list(frame.columns) == list(g["columns"])
    np.testing.assert_allclose(frame.values, g["summary"], rtol=1e-12)


# ------------------------------------------------------------------ C-ABI surface
def test_library_exports_every_declared_symbol():
    lib = _native.load()
    # production ABI + the test seams (include/plspm_hip_test.h); together they are exactly what the library exports
    header = open(os.path.join(ROOT, "include", "plspm_hip.h")).read() + open(os.path.join(ROOT, "include", "plspm_hip_test.h")).read()
    declared = sorted(set(re.findall(r"\b(plspm_[a-z_]+)\s*\(", header)) - {"plspm_model", "plspm_fit_result"})
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), "libplspm_hip.so does not export " + name
    assert sorted(_native.EXPORTS) == declared
    # ... and nothing else: `nm -D` of the library shows the headers' symbols only (csrc/exports.map; no internal seam leaks)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in nm.splitlines() if line.split()[-2] in "TDBWV")
    assert exported == declared, sorted(set(exported) ^ set(declared))
    assert lib.plspm_abi_version() == 4


def test_sub_batch_planner_covers_the_call_and_shrinks_geometrically():
    """plspm_chunk_plan (host arithmetic; plspm_bootstrap "boot_chunks", plspm_group_bootstrap "chunks"): ONE call as sub-batches whose
    transfer hides under the next sub-batch's kernels.  The parts cover B exactly, fall by about the ratio, are multiples of the alignment
    (64: whole count tiles of the int8 Gram; the library passes the replicates of one ROUND of the device -- 1,280 for the headline model:
    a part that ends inside a round pays for the whole round) but for the last, and small calls -- below 2 MiB of results -- stay in one piece."""
    rec = 158 * 8
    assert _native.chunk_plan(5000, rec) == [2560, 1536, 904]                   # 6.3 MB of records, three parts of whole count tiles
    assert _native.chunk_plan(5000, rec, align=1280) == [2560, 1280, 1160]      # ... of whole rounds: the headline batch as the library cuts it
    assert _native.chunk_plan(40000, rec, align=1280) == [20480, 12800, 6720]
    assert _native.chunk_plan(2500, rec, align=1280) == [1280, 1220]
    assert _native.chunk_plan(5000, rec, chunks=1) == [5000]
    assert _native.chunk_plan(1000, rec) == [1000]                               # 1.26 MB: nothing worth hiding
    assert len(_native.chunk_plan(1024, rec, chunks=4)) <= 2                     # no part below 512 units asked for
    for align in (0, 64, 320, 1280):
        a = align or 64
        for B in (1, 63, 64, 65, 511, 1664, 1700, 4999, 5000, 40000, 123457, 1 << 20):
            for chunks in (0, 1, 2, 3, 5, 8):
                for ratio in (10, 50, 60, 100):
                    parts = _native.chunk_plan(B, rec, chunks, ratio, align)
                    assert sum(parts) == B and all(p >= 1 for p in parts), (B, chunks, ratio, parts)
                    assert 1 <= len(parts) <= max(1, chunks if chunks else 3)
                    assert all(p % a == 0 for p in parts[:-1])
                    if len(parts) > 1:
                        assert min(parts) >= 64
                        assert all(parts[k + 1] <= parts[k] + a for k in range(len(parts) - 2)), parts        # falling sizes (the last takes the remainder)
    # 100 %: equal parts; a steeper ratio makes the first part larger
    eq = _native.chunk_plan(6144, rec, 3, 100)
    assert eq == [2048, 2048, 2048]
    assert _native.chunk_plan(6144, rec, 3, 30)[0] > _native.chunk_plan(6144, rec, 3, 60)[0] > eq[0]
    with pytest.raises(_native.NativeBackendError):
        _native.chunk_plan(0, rec)


def test_host_rng_mirror_properties():
    a = _native.bootstrap_indices(7, 0, 1000)
    b = _native.bootstrap_indices(7, 0, 1000)
    c2 = _native.bootstrap_indices(7, 1, 1000)
    d = _native.bootstrap_indices(8, 0, 1000)
    assert np.array_equal(a, b) and not np.array_equal(a, c2) and not np.array_equal(a, d)
    assert a.min() >= 0 and a.max() < 1000
    big = _native.bootstrap_indices(1, 5, 200000)
    frac_unique = len(np.unique(big)) / 200000.0
    assert abs(frac_unique - (1 - np.exp(-1))) < 0.01                 # ~63.2 % distinct rows per resample
    assert abs(big.mean() / 200000.0 - 0.5) < 0.01


def test_tile_row_cut_of_the_int8_gram_covers_the_batch_and_balances_the_last_round():
    """plspm_gram_tile_plan (host arithmetic of the six-plane int8 Gram): rows of 20 and 16 count tiles cover the batch
    with less than one row of padding, the headline shape (313 count tiles x 60 pair tiles, 256 CUs) takes three tall + one short
    tile per CU, batches that fill whole rounds of the 256-replicate kernel keep that kernel, and a cut is never modelled slower than
    tall rows alone.  The cut never shows in a result (GPU test: bit-identical matrices for every cut)."""
    def span(a, b, ntx, cus=256):                       # the model restated: list scheduling per XCD, tall tiles first
        worst = 0.0
        ta, tb = a * ntx, b * ntx
        pa, pb = (ta + 7) // 8, (tb + 7) // 8
        for x in range(8):
            na, nb = max(0, min(pa, ta - x * pa)), max(0, min(pb, tb - x * pb))
            load = [0.0] * (cus // 8)
            for t in range(na + nb):
                load[load.index(min(load))] += 20.0 if t < na else 16.6
            worst = max(worst, max(load))
        return worst
    wide, a, b = _native.i8_tile_plan(313, 60)
    assert wide and (a, b) == (12, 5) and span(a, b, 60) == 3 * 20.0 + 16.6
    for ct, ntx in ((313, 60), (157, 60), (625, 60), (219, 60), (1250, 60), (313, 231), (40, 4), (1000, 7)):
        wide, a, b = _native.i8_tile_plan(ct, ntx)
        if wide:
            assert 20 * a + 16 * b >= ct and 20 * a + 16 * b - ct < 20 + 16 * (b > 0), (ct, ntx, a, b)
            assert span(a, b, ntx) <= span((ct + 19) // 20, 0, ntx) + 1e-9, (ct, ntx, a, b)
            assert abs(span(a, b, ntx) - min(span(max(0, (ct - 16 * k + 19) // 20), k, ntx) for k in range(0, min((ct + 15) // 16, 48) + 1)
                                             if max(0, (ct - 16 * k + 19) // 20) or 16 * k >= ct)) < 1e-9
        _, a0, b0 = _native.i8_tile_plan(ct, ntx, mix=False)
        assert b0 == 0 and 20 * a0 >= ct
    assert _native.i8_tile_plan(256, 60)[0] is False        # 4,096 replicates: 960 tiles of 256 = 3.75 rounds x 16.35 beat every cut
    assert _native.i8_tile_plan(4, 60)[0] is False          # one tile row either way: the lower tile wins
    assert _native.i8_tile_plan(313, 60) == _native.i8_tile_plan(313, 60)


@pytest.mark.skipif(_native.device_count() > 0, reason="needs a box WITHOUT a GPU")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    assert _native.device_count() == 0
    with pytest.raises(_native.NativeBackendError):
        _native.NativeModel(np.array([0, 2, 4]), np.array([[0, 0], [1, 0]]), np.array([0, 0]), 0, True, 100, 1e-6)
    from plspm.plspm import Plspm
    sat = satisfaction_frame()
    s = c.Structure(); s.add_path(["IMAG"], ["EXPE"])
    cfg = c.Config(s.path())
    cfg.add_lv_with_columns_named("IMAG", Mode.A, sat, "imag"); cfg.add_lv_with_columns_named("EXPE", Mode.A, sat, "expe")
    with pytest.raises(_native.NativeBackendError):
        Plspm(sat, cfg)


def test_unsupported_paths_raise_not_implemented():
    from plspm.plspm import Plspm
    sat = satisfaction_frame()
    s = c.Structure(); s.add_path(["IMAG"], ["EXPE"])
    cfg = c.Config(s.path(), default_scale=Scale.ORD)                 # optimal scaling runs on the device: no CPU fallback
    cfg.add_lv_with_columns_named("IMAG", Mode.A, sat, "imag"); cfg.add_lv_with_columns_named("EXPE", Mode.A, sat, "expe")
    from plspm._native import NativeBackendError
    with pytest.raises(NativeBackendError):
        Plspm(sat, cfg)
    from plspm.util import MissingDataError
    miss = sat.copy(); miss.iloc[0, 0] = np.nan                       # a NaN in an ORD column: the reference's own failure, by name and message
    with pytest.raises(MissingDataError, match="exog contains inf or nans"):
        Plspm(miss, cfg)
    mixed = c.Config(s.path(), default_scale=Scale.NUM)               # NaN in a NUM column beside ORD columns: the reference estimates, this backend does not yet
    mixed.add_lv("IMAG", Mode.A, *[c.MV(n, Scale.ORD) for n in sat.columns if n.startswith("imag")])
    mixed.add_lv_with_columns_named("EXPE", Mode.A, sat, "expe")
    miss2 = sat.copy(); miss2.loc[miss2.index[3], [n for n in sat.columns if n.startswith("expe")][0]] = np.nan
    with pytest.raises(NotImplementedError):
        Plspm(miss2, mixed)
    hoc = c.Config(s.path())                                          # metric HOC: the reference cannot run it either
    hoc.add_higher_order("EXPE", Mode.A, ["A", "B"])
    hoc.add_lv_with_columns_named("IMAG", Mode.A, sat, "imag"); hoc.add_lv_with_columns_named("A", Mode.A, sat, "expe")
    hoc.add_lv_with_columns_named("B", Mode.A, sat, "qual")
    with pytest.raises(NotImplementedError):
        Plspm(sat, hoc)
    with pytest.raises(AssertionError):
        Plspm(sat, cfg, tolerance=0)
    with pytest.raises(AssertionError):
        Plspm(sat, cfg, bootstrap_iterations=100, processes=3)


def test_hoc_first_stage_path_matches_reference_construction():
    """reference tests/test_estimator.py:22-36."""
    from plspm.estimator import Estimator
    structure = c.Structure()
    structure.add_path(["MANDRILL", "BONOBO"], ["APE"])
    structure.add_path(["APE"], ["GOAT"])
    initial = structure.path()
    config = c.Config(initial)
    config.add_higher_order("APE", Mode.A, ["CHEDDAR", "GOUDA"])
    estimator = Estimator(config)
    st = c.Structure(initial)
    st.add_path(["MANDRILL", "BONOBO"], ["CHEDDAR"])
    st.add_path(["MANDRILL", "BONOBO"], ["GOUDA"])
    st.add_path(["GOUDA", "CHEDDAR"], ["GOAT"])
    expected = st.path().drop("APE").drop("APE", axis=1)
    pdt.assert_frame_equal(expected, estimator.hoc_path_first_stage(config))


def test_workload_generator_matches_the_oracle_copy():
    """bench.py / tools use tools/synthetic.py (so that they never import the test-only oracle for their workload); the oracle's own
    generator, which made the goldens, must stay bit-identical."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synthetic
    import plspm_oracle as orc
    assert np.array_equal(synthetic.satisfaction_C(), orc.satisfaction_C()) and np.array_equal(synthetic.chain_C(20), orc.chain_C(20))
    for n, C, k, seed in ((500, orc.satisfaction_C(), 10, 0), (300, orc.chain_C(7), 3, 5)):
        Xa, ba = synthetic.synth(n, C, k, seed=seed)
        Xb, bb = orc.synth(n, C, k, seed=seed)
        assert np.array_equal(Xa, Xb) and all(np.array_equal(u, v) for u, v in zip(ba, bb))


def test_bench_guard_fixture_is_current():
    """tests/golden/bench_guard.npz (bench.py's correctness guard) == the oracle on replicate 0 of the seeded stream."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synthetic
    import plspm_oracle as orc
    g = np.load(os.path.join(GOLDEN, "bench_guard.npz"))
    idx0 = _native.bootstrap_indices(1, 0, 10000)
    assert int(idx0.astype(np.int64).sum()) == int(g["idx_sum"])
    X, blocks = synthetic.synth(10000, synthetic.satisfaction_C(), 10, seed=0)
    row, its = orc.bootstrap_replicate(X, orc.Model(blocks, synthetic.satisfaction_C(), "AAAAAA", "path", True), idx0, orc.correction(10000))
    assert its == int(g["iterations"])
    np.testing.assert_allclose(row, g["row"], rtol=1e-12)


def _newest_profile(suffix):
    """profiles/rNN<suffix> with the largest round number NN (the latest committed measurement of that kind)."""
    import glob
    import re
    hits = [(int(re.match(r"r(\d+)", os.path.basename(p)).group(1)), p) for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*" + suffix))
            if re.fullmatch(r"r\d+" + re.escape(suffix), os.path.basename(p))]
    assert hits, suffix
    return max(hits)


def test_committed_bench_line_follows_the_driver_contract():
    """The newest profiles/rNN_bench_n1.json is the line `python bench.py` printed on the GPU box: keys, types and the tier's conventions (dtype =
    arithmetic type, vs_baseline null without a published number, config names the workload, roofline + cpu_baseline + single_fit) -- and its
    `roofline` object describes the launch that was TIMED: the kernel name occurs in the same round's rocprofv3 summary with an average duration
    within 5 % of avg_launch_ms (timed-region average), the executed ops are 1.0-1.1 x the algorithmic ones, the traffic is that kernel's."""
    import json
    rnd, path = _newest_profile("_bench_n1.json")
    assert rnd >= 6, "the committed line predates round 6's bench.py"
    line = json.load(open(path))
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                      ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict),
                      ("single_fit", dict)):
        assert isinstance(line[key], kind), key
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert "workload" in line["config"] and "model" not in line["config"]
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"].split(" at ")[0] in baseline["metric"]
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s", "TOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["traffic"] > 0

    def fracs(obj, trail=""):
        for k, v in obj.items():
            if isinstance(v, dict):
                yield from fracs(v, trail + k + ".")
            elif k == "frac" or k.startswith("frac_"):
                yield trail + k, v
    for name, v in list(fracs(roof)) + list(fracs(line["single_fit"])):
        assert 0.0 < v <= 1.0, (name, v)                     # nothing called a fraction of a peak exceeds it
    reps = line["config"]["replicates_per_step"] // line["n_gpus"]
    ratio = roof["executed_ops"] / (roof["algorithmic_ops_per_replicate"] * reps)
    assert 1.0 <= ratio <= 1.1 and abs(ratio - roof["executed_over_algorithmic"]) < 1e-3, ratio
    assert roof["tile_rows"]["replicate_slots"] >= reps
    summary = json.load(open(os.path.join(ROOT, "profiles", "r%02d_rocprof_summary.json" % rnd)))
    mine = [k for k in summary["kernels"] if k["kernel"] == roof["kernel"]]
    assert len(mine) == 1, (roof["kernel"], [k["kernel"] for k in summary["kernels"] if "gram" in k["kernel"]])
    assert summary["kernels"][0]["kernel"] == roof["kernel"]                         # ... and it is that run's dominant kernel
    timed = [p for p in summary["dominant_by_phase"] if p["phase"].startswith("TIMED")][0]
    assert abs(timed["avg_us"] * 1e-3 - roof["avg_launch_ms"]) <= 0.05 * roof["avg_launch_ms"], (timed, roof["avg_launch_ms"])
    ctr = summary["counters"][roof["kernel"]]
    assert abs(ctr["hbm_bytes_per_dispatch"] - roof["traffic"]) <= 0.05 * roof["traffic"], (ctr, roof["traffic"])
    traffic = json.load(open(os.path.join(ROOT, "profiles", "r%02d_gram_i8_traffic.json" % rnd)))
    assert traffic["kernel"] == roof["kernel"]
    assert line["cold"]["value"] > 0 and line["cold"]["ms_per_step"] >= 0.9 * line["ms_per_step"]
    cfg = line["config"]
    assert cfg["transport"] in ("none", "rccl") and cfg["replicate_ranges"][0][0] == 0 and cfg["replicate_ranges"][-1][1] == cfg["replicates_per_step"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and isinstance(cpu["sample"], str)
    assert abs(line["value"] - line["config"]["replicates_per_step"] * line["n_gpus"] / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    # SURVEY 8(f) rank 1 beside the headline (round 6: one solver launch + verification): the Scale.NUM counterpart of the workload, measured by the same run
    num = line["next_rows"]["nonmetric_num_bootstrap"]
    assert num["one_launch_solver"] == 1 and num["all_ok"] and num["replicates_per_s"] >= 6.0e6 and num["kernel_ms_per_step"]["solver"] <= 0.12, num
    assert num["replicate_iterations"][0] >= 2 and num["fit_iterations"] >= 2
    # ... and the categorical (Scale.ORD) one (round 6: one launch + verification, six columns per lane)
    cat = line["next_rows"]["categorical_bootstrap"]
    assert cat["replicates_per_step_1000"]["replicates_per_s"] >= 4.0e5 and cat["replicates_per_step_5000"]["replicates_per_s"] >= 5.0e5, cat
    assert cat["replicates_per_step_1000"]["all_ok"] and cat["replicates_per_step_5000"]["all_ok"]
    # ... and both stages of a higher order construct per replicate on the reference's mobi data (round 6: ten-point items at two waves per SIMD)
    hoc = line["next_rows"]["hoc_two_stage_bootstrap"]
    assert hoc["ORD"]["replicates_per_s"] >= 1.0e5 and hoc["ORD"]["replicates_per_s_at_40000_per_call"] >= 1.5e5 and hoc["NUM"]["replicates_per_s"] >= 2.0e6, hoc
    assert hoc["ORD"]["ok_replicates"] >= 4900 and hoc["NUM"]["ok_replicates"] == 5000
    # the two single-fit configurations of BASELINE.json: iteration counts, HIP-event kernel times, A_fit / F_fit rooflines
    for key, (n, p, l) in (("configs[1]", (10000, 60, 6)), ("configs[4]", (1000000, 200, 20))):
        fit = line["single_fit"][key]
        assert (fit["N"], fit["P"], fit["L"]) == (n, p, l) and fit["status"] == 0 and fit["iterations"] >= 2
        r = fit["roofline"]
        assert r["A_fit_bytes"] == 16.0 * n * p + 8.0 * n * l and r["F_fit_flops"] == float(n) * p * (p + 1) + 2.0 * n * p * l
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert abs(sum(fit["kernel_ms"].values()) - fit["device_ms_total"]) < 1e-3 and fit["device_ms_total"] >= r["bound_ms"]


def test_ordnom_missing_probe_pins_the_reference_behaviour_this_backend_mirrors():
    """tests/golden/g16_ordnom_missing_probe.json (tests/golden/probe_ordnom_missing.py driving the real reference in the build container): with NaN cells in
    a Scale.ORD / NOM column the reference never returns a usable estimate -- it raises (statsmodels MissingDataError "exog contains inf or nans" in the
    majority of the cases: every Mode-B case among them), returns all-zero weights, or returns weights that move by more than 1e-2 when the rows of the data
    set are permuted (complete data: 1e-15).  plspm.util.MissingDataError carries that majority failure's name and message."""
    import json
    from plspm.util import MissingDataError
    probe = json.load(open(os.path.join(GOLDEN, "g16_ordnom_missing_probe.json")))
    assert probe["control_complete_data_max_abs_weight_change_under_row_permutation"] < 1e-12
    recs = probe["records"]
    assert len(recs) == 30
    raised = [r for r in recs if r["outcome"] == "raises"]
    named = [r for r in raised if r["exception"].endswith("MissingDataError")]
    assert len(named) > len(raised) / 2 and all(r["message"] == "exog contains inf or nans" for r in named)
    assert all(r["outcome"] == "raises" for r in recs if r["modes"] == "BBB")
    for r in recs:
        if r["outcome"] == "estimates":                       # never a usable estimate: degenerate, or not a function of the data SET
            zero = max(abs(w) for w in r["weights"]) == 0.0
            moved = r.get("max_abs_weight_change_under_row_permutation", 0.0) > 1e-2 or r["rows_permuted"]["outcome"] == "raises"
            assert zero or moved, r["case"]
    assert issubclass(MissingDataError, Exception) and MissingDataError.__name__ == named[0]["exception"].rsplit(".", 1)[1]


def test_gather_mode_env_is_validated(monkeypatch):
    """PLSPM_GATHER picks the exchange of a multi-GPU bootstrap (plspm/parallel.py): all (ONE ncclAllGather per sub-batch) or root (the gather to rank 0 the reference's parent
    process performs, bootstrap.py:96-111); anything else is a ValueError naming the choices -- no silent default."""
    from plspm import parallel
    monkeypatch.delenv("PLSPM_GATHER", raising=False)
    assert parallel.gather_to_root() is False
    monkeypatch.setenv("PLSPM_GATHER", "root")
    assert parallel.gather_to_root() is True
    monkeypatch.setenv("PLSPM_GATHER", "ALL")
    assert parallel.gather_to_root() is False
    monkeypatch.setenv("PLSPM_GATHER", "ring")
    with pytest.raises(ValueError, match="'all' or 'root'"):
        parallel.gather_to_root()
