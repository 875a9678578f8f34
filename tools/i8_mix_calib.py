"""Cost model of the tile-row cut (plspm_gram_i8.hip i8_mix_plan) for gram_i8p_kernel: Gram kernel time of 960 tiles = 3.75 rounds of each height --
six planes: tall 320 replicates (B = 5,120) / short 256 (B = 4,096, "i8_short_rows" 16); seven planes: tall 256 (B = 4,096) / short 192 (B = 3,072) --
then the automatic cut against all-tall rows for a few batch sizes.  usage: i8_mix_calib.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
def model(S):
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    nm.upload(X); nm.set_option("i8_slices", S)
    for w in range(40): nm.bootstrap_device(5000, seed=1, rep_offset=w * 5000)
    nm.sync()
    return nm
def gram_ms(nm, B, rt, short):
    nm.set_option("i8_rt", rt); nm.set_option("i8_short_rows", short)
    for w in range(3): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
    nm.sync(); nm.profile(True); nm.profile_reset()
    for k in range(12): nm.bootstrap_device(B, seed=1, rep_offset=(3 + k) * B)
    nm.sync(); nm.profile(False)
    ms, n = nm.profile_read("gram")
    return round(ms / n, 4), nm.get_option("last_i8_rt"), nm.get_option("last_i8_short"), nm.get_option("last_i8_priv")
for S, tall, short_b in ((6, 20, 4096), (7, 16, 3072)):
    nm = model(S)
    cases = [("tall x 960", 16 * 16 * tall, tall, -1), ("short x 960", short_b, tall, 16)]
    for rnd in range(4):
        for name, B, rt, sh in (cases if rnd % 2 == 0 else cases[::-1]):
            t, lrt, lsh, pv = gram_ms(nm, B, rt, sh)
            print(json.dumps({"S": S, "case": name, "B": B, "gram_ms": t, "last_i8_rt": lrt, "last_i8_short": lsh, "priv": pv}), flush=True)
    for B in (2000, 2500, 3000, 3500, 4096, 5000, 6000, 7500, 10000, 20000):
        out = {"S": S, "B": B}
        for rnd in range(2):
            for name, rt in ((("auto", 0), ("tall", tall)) if rnd == 0 else (("tall", tall), ("auto", 0))):
                t, lrt, lsh, pv = gram_ms(nm, B, rt, -1)
                out.setdefault(name, []).append(t)
                if name == "auto": out["auto_cut_short_rows"] = lsh
        print(json.dumps(out), flush=True)
