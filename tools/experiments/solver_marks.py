"""Phase clocks of one solver problem (library built with EXTRA=-DPLSPM_DEBUG_MARKS) at different batch sizes: 256 replicates = one
problem per CU (no co-resident waves), 2048 = eight per CU, 5000 = the headline batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
nm.upload(X)
for opt in sys.argv[1:]:
    k, v = opt.split("=")
    nm.set_option(k, int(v))
for B in (256, 256, 2048, 5000, 5000):
    sys.stderr.write("--- B = %d\n" % B); sys.stderr.flush()
    nm.bootstrap_device(B, seed=1, rep_offset=0)
    nm.sync()
sys.stderr.write("--- summary of the 5,000 records\n"); sys.stderr.flush()
for k in range(3):
    nm.summary(5000, np.ones(nm.row_width))
