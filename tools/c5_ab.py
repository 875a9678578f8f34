#!/usr/bin/env python3
"""A/B of launch-geometry options on BASELINE.json configs[4] (1M x 200 x 20, Mode B, FACTORIAL): per-kernel HIP-event times of plspm_fit."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import synthetic as orc  # noqa: E402
from plspm import _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
C = orc.chain_C(20)
X, blocks = orc.synth(n, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, C.astype(np.uint8), np.ones(20, dtype=np.int32), 1, True, 100, 1e-6, 0)
m.upload(X)
ref = m.fit(want_scores=True)
for opts in ({}, {"scores_tile": 16}, {"fit_chunks": 512}):
    for k, v in opts.items():
        m.set_option(k, v)
    out = m.fit(want_scores=True)
    assert np.array_equal(out["scores"], ref["scores"]) or np.allclose(out["scores"], ref["scores"], rtol=1e-12, atol=1e-14)
    m.profile(True); m.profile_reset()
    for _ in range(5):
        m.fit(want_scores=True)
    ms = {name: round(m.profile_read(name)[0] / max(m.profile_read(name)[1], 1), 4) for name in ("gram", "reduce", "solver", "scores")}
    m.profile(False)
    print(json.dumps({"opts": opts, "ms": ms, "gram_TF_contract": round(float(n) * 200 * 201 / ms["gram"] / 1e9, 1),
                      "scores_TBps_moved": round(8.0 * n * (208 + 20) / ms["scores"] / 1e9, 2)}), flush=True)
    for k in opts:
        m.set_option(k, {"scores_tile": 0, "wide_nw": 4, "fit_chunks": 0}[k])
