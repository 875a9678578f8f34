"""Phase clocks of one wave-solver problem under load (marks build: make -C plspm-python_amd/csrc marks; PLSPM_HIP_LIB=.../build/marks/libplspm_hip_marks.so),
headline shape, all-Mode-A against all-Mode-B blocks; the library prints the clocks of problem 0 of every batch to stderr.
usage: wave_marks.py [B [L]]      (L: chain model of L LVs x 10 MVs instead -- 12: the split rows solver's 120 x 12 case)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
C = satisfaction_C()
if len(sys.argv) > 2:
    Lc = int(sys.argv[2])
    C = np.zeros((Lc, Lc), dtype=np.int64)
    for j in range(Lc):
        if j - 1 >= 0: C[j, j - 1] = 1
        if j - 3 >= 0: C[j, j - 3] = 1
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
Lm = C.shape[0]
for modes in ("A" * Lm, "B" * Lm, "AB" * (Lm // 2) + "A" * (Lm % 2)):
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.array([1 if c == "B" else 0 for c in modes], dtype=np.int32), 2, True, 100, 1e-6, 0)
    nm.upload(X)
    sys.stderr.write("== modes %s\n" % modes); sys.stderr.flush()
    for w in range(3):
        rows, st, it = nm.bootstrap(B, seed=1, rep_offset=w * B)
    print(json.dumps({"modes": modes, "B": B, "iterations_histogram": np.bincount(it).tolist(), "last_solver": nm.get_option("last_solver"), "status_ok": bool(np.all(st == 0))}), flush=True)
    nm.close()
