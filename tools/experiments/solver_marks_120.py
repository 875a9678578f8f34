"""Phase clocks of one quad-solver problem (solver_quad.h; with nm.set_option("solver_quad", 0): of the split rows solver) (marks build) on the 10k x 120 x 12 model of tools/size_bench.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import chain_C, synth
C = chain_C(12)
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(12, dtype=np.int32), 2, True, 100, 1e-6, 0)
nm.upload(X)
for B in (256, 512, 5000):
    sys.stderr.write("--- B = %d\n" % B); sys.stderr.flush()
    nm.bootstrap_device(B, seed=1, rep_offset=0)
    nm.sync()
print("solver", nm.get_option("last_solver"))
