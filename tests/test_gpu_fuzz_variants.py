"""GPU differential test of the non-plain variants THROUGH THE HOST API (exercises plspm/_compile.py augment /
with_missing_indicators, plspm/weights.py _incomplete_rows and the matching device paths): seeded random models with
(0) ORD / NOM / NUM mixes on Likert-quantised data, (1) metric data with NaNs, (2) Scale.NUM data with NaNs -- single fit plus the
first bootstrap replicate of the seeded stream, against the oracle."""
import numpy as np
import pandas as pd
import pytest

import plspm_oracle as orc
from helpers import assert_close
from test_gpu_parity import _ragged, _random_dag

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-9


def make_case(seed):
    rng = np.random.default_rng(7000 + seed)
    variant = seed % 3
    L = int(rng.integers(2, 7))
    C = _random_dag(L, rng, density=float(rng.uniform(0.4, 0.9)))
    sizes = [int(rng.integers(2, 7)) for _ in range(L)]
    n = int(rng.integers(120, 600))
    X, blocks = _ragged(n, C, sizes, seed=500 + seed)
    P = X.shape[1]
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    modes = "".join("AB"[int(rng.integers(0, 2))] for _ in range(L))
    scales = None
    if variant == 0:
        scales = [["ORD", "NOM", "NUM"][int(rng.integers(0, 3))] for _ in range(P)]
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        for p in range(P):
            if scales[p] != "NUM":
                X[:, p] = np.clip(np.round(3 + 1.1 * Z[:, p]), 1, 5)
    else:
        holes = int(rng.integers(3, 40))
        hole_lvs = set()
        for _ in range(holes):
            p = int(rng.integers(0, P))
            X[int(rng.integers(0, n)), p] = np.nan
            hole_lvs.add(next(l for l, b in enumerate(blocks) if p in b))
        if variant == 2:
            scales = ["NUM"] * P
            modes = "".join("A" if l in hole_lvs else modes[l] for l in range(L))       # Mode B needs a complete block (mode.py:55-56)
    model = orc.Model(blocks, C, modes, scheme, bool(rng.integers(0, 2)), tol=1e-7 if scales else 1e-6, scales=scales)
    return X, model, variant


def build_config(model, names):
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.scale import Scale
    lvs = ["L%d" % l for l in range(model.L)]
    path = pd.DataFrame(model.C, index=lvs, columns=lvs)
    kinds = {"NUM": Scale.NUM, "ORD": Scale.ORD, "NOM": Scale.NOM, "RAW": Scale.RAW}
    config = c.Config(path, scaled=model.scaled, default_scale=None if model.scales is None else Scale.NUM)
    for l, b in enumerate(model.blocks):
        config.add_lv(lvs[l], Mode.A if model.modes[l] == "A" else Mode.B,
                      *[c.MV(names[p], None if model.scales is None else kinds[model.scales[p]]) for p in b])
    return config, lvs


@pytest.mark.parametrize("seed", range(36))
def test_random_variant_through_the_api(seed):
    from plspm import _native
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    X, model, variant = make_case(seed)
    Xf = orc.filter_missing(X, model)
    tag = "seed %d variant %d L=%d P=%d n=%d %s %s" % (seed, variant, model.L, X.shape[1], Xf.shape[0], model.modes, model.scheme)
    try:
        r = orc.fit(Xf, model)
    except Exception:
        pytest.skip("the oracle (= the reference's arithmetic) fails on this random model")
    if not all(np.all(np.isfinite(r[k])) for k in ("weights", "loadings", "path_coef")):
        pytest.skip("non-finite reference result")
    names = ["v%d" % p for p in range(X.shape[1])]
    frame = pd.DataFrame(X, columns=names)
    config, lvs = build_config(model, names)
    scheme = {"centroid": Scheme.CENTROID, "factorial": Scheme.FACTORIAL, "path": Scheme.PATH}[model.scheme]
    calc = Plspm(frame, config, scheme, model.max_iter, model.tol, bootstrap=True, bootstrap_iterations=10, seed=seed)
    assert calc.iterations() == r["iterations"], tag
    om = calc.outer_model()
    assert_close(om.loc[names, "weight"].values, r["weights"], RTOL, ATOL, what=tag + " weights")
    assert_close(om.loc[names, "loading"].values, r["loadings"], RTOL, ATOL, what=tag + " loadings")
    assert_close(calc.path_coefficients().loc[lvs, lvs].values, r["path_coef"], RTOL, ATOL, what=tag + " path")
    assert_close(calc.scores().loc[:, lvs].values, r["scores"], 1e-6, 1e-8, what=tag + " scores")
    assert_close(calc.crossloadings().loc[names, lvs].values, r["crossloadings"], RTOL, 1e-8, what=tag + " crossloadings")
    # first replicate of the seeded stream
    boot = calc.bootstrap()
    n = Xf.shape[0]
    if boot.status()[0] != 0:
        return
    try:
        mine, its = orc.bootstrap_replicate(Xf, model, _native.bootstrap_indices(seed, 0, n), orc.correction(n))
    except Exception:
        pytest.fail(tag + ": the device accepted a replicate the reference's arithmetic rejects")
    if not np.all(np.isfinite(mine)):
        return
    assert its == boot.replicate_iterations()[0], tag
    dev = calc._result.compiled                              # rows are in device column order
    inv = dev.inv_index[dev.inv_index >= 0]
    row = boot.replicates()[0]
    P, L = X.shape[1], model.L
    ne = (len(row) - 2 * P - L) // 2
    got = np.concatenate((row[:P][inv], row[P:P + L + 2 * ne], row[P + L + 2 * ne:][inv]))
    assert_close(got, mine, RTOL, 1e-8, what=tag + " replicate 0")
