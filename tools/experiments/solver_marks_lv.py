"""Phase clocks of one solver problem (marks build) on a chain model of L LVs x `per` MVs (tools/size_bench_models.py): usage solver_marks_lv.py L per"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import synth
from size_bench_models import chain_C
L, per = int(sys.argv[1]), int(sys.argv[2])
C = chain_C(L)
X, blocks = synth(10000, C, per, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(L, dtype=np.int32), 2, True, 100, 1e-6, 0)
nm.upload(X)
for B in (256, 5000, 5000):
    sys.stderr.write("--- B = %d\n" % B); sys.stderr.flush()
    nm.bootstrap_device(B, seed=1, rep_offset=0)
    nm.sync()
print("solver", nm.get_option("last_solver"))
