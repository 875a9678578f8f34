import sys
for p in ("plspm-python_amd","oracle","tests"): sys.path.insert(0,p)
import numpy as np, plspm_oracle as orc
from test_gpu_nmx import gpu_model
np.set_printoptions(linewidth=220, precision=5)
C = orc.chain_C(3)
X, blocks = orc.synth(120, C, 3, seed=44)
X[:, 0] = 0.0; X[:4, 0] = 1.0
model = orc.Model(blocks, C, "AAA", "path", True, tol=1e-6, scales=["NUM"] * X.shape[1])
rs = np.random.RandomState(3)
idx = np.vstack([np.arange(120), 4 + rs.randint(116, size=120), rs.randint(120, size=120), 4 + rs.randint(116, size=120)]).astype(np.int32)
for gp in (0, 1, 2):
    nm, inv = gpu_model(X, model)
    nm.set_option("gram_path", gp)
    rows, status, iters = nm.bootstrap(4, idx=idx)
    print("gram_path", gp, "last", nm.get_option("last_gram_path"), "wave16", nm.get_option("last_nm_wave16"), status, iters)
    print(rows[1][:12]); print(rows[0][:12])
