"""GPU: the drop-in API under torch.distributed (RCCL) gives the same bootstrap summaries as the single-process path
(one rank on the one GPU of the test box: exercises rendezvous, the device-resident all_gather and the device summaries)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(cmd):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert lines, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[-1][7:])


def test_api_bootstrap_under_torchrun_equals_single_process():
    script = os.path.join(HERE, "dist_api_script.py")
    plain = _run([sys.executable, script])
    env_cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29533", script]
    dist = _run(env_cmd)
    for key in plain:
        np.testing.assert_allclose(np.array(dist[key], dtype=float), np.array(plain[key], dtype=float), rtol=1e-12, atol=1e-14, err_msg=key)
