#!/usr/bin/env python3
"""Fixture for bench.py's correctness guard: replicate 0 of the seeded resampling stream (seed 1) on the headline workload,
computed with the oracle (pinned on the reference).  Needs libplspm_hip.so for the host mirror of the Philox stream
(plspm_bootstrap_indices runs on the CPU).  Run: python tests/golden/make_bench_guard.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "plspm-python_amd")]
import plspm_oracle as orc  # noqa: E402
import synthetic  # noqa: E402
from plspm import _native  # noqa: E402

N, K, L = 10000, 10, 6
X, blocks = synthetic.synth(N, synthetic.satisfaction_C(), K, seed=0)
model = orc.Model(blocks, synthetic.satisfaction_C(), "A" * L, "path", True)
idx0 = _native.bootstrap_indices(1, 0, N)
row, its = orc.bootstrap_replicate(X, model, idx0, orc.correction(N))
np.savez_compressed(os.path.join(HERE, "bench_guard.npz"), row=row, iterations=its, idx_sum=int(idx0.astype(np.int64).sum()), idx_head=idx0[:16])
print("wrote bench_guard.npz: iterations", its, "row", row.shape)
