"""Schedule variants of gram_i8_kernel<6, 4, V, 16, 20> (six planes, 320-replicate tile, eight waves; library built with
`make experiments`): Gram time per variant, alternating rounds.  usage: PLSPM_HIP_LIB=.../libplspm_hip_exp.so i8_variants20.py [variants...]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
B = 5000
variants = [int(a) for a in sys.argv[1:]] or [3, 0, 6, 1, 4, 7, 5, 12, 13, 18, 21]
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
nm.upload(X)
nm.set_option("i8_rt", 20); nm.set_option("i8_short_rows", int(os.environ.get("I8_SHORT", "-1"))); nm.set_option("i8_waves", 8)
ref = nm.bootstrap(400, seed=1)[0]
res, ok = {v: [] for v in variants}, {}
for w in range(30): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
nm.sync()
for rnd in range(4):
    for var in (variants if rnd % 2 == 0 else variants[::-1]):
        nm.set_option("i8_variant", var)
        try:
            if var not in ok: ok[var] = bool(np.array_equal(nm.bootstrap(400, seed=1)[0], ref)) if var < 100 else None
            for w in range(2): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
            nm.sync(); nm.profile(True); nm.profile_reset()
            for k in range(10): nm.bootstrap_device(B, seed=1, rep_offset=(2 + k) * B)
            nm.sync(); nm.profile(False)
            ms, n = nm.profile_read("gram")
            res[var].append(ms / n)
        except Exception as e:
            ok[var] = "error: %s" % e
for var in variants:
    V = var % 100
    print(json.dumps({"S": 6, "rt": 20, "waves": 8, "variant": var, "NS": 5 if V >= 18 else 3 + V % 3, "rstep": 1 + (V // 3) % 3, "dma_head": (V % 18) // 9, "pair_barrier": V >= 18,
                      "ablate": var // 100, "gram_ms": [round(x, 4) for x in res[var]], "identical_rows": ok.get(var)}), flush=True)
