"""The int8 Gram with private count fragments (kernels_gram_i8p.h, set_option("i8_priv", 1)) against the round-3 kernel: Gram time from the
library's HIP events over alternating rounds, step time of un-profiled steps, bit-identity of the records.
usage: i8p_bench.py [B ...]            (PLSPM_HIP_LIB=.../libplspm_hip_exp.so with I8P_ABLATE=1 / I8P_VARIANTS=1: ablation probes / schedule variants as well)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
Bs = [int(a) for a in sys.argv[1:]] or [5000, 5120]
slices = int(os.environ.get("I8P_SLICES", "0"))
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
nm.upload(X)
if slices: nm.set_option("i8_slices", slices)
for w in range(60): nm.bootstrap_device(5000, seed=1, rep_offset=w * 5000)
nm.sync()
configs = [("r3", {"i8_priv": 0}), ("priv", {"i8_priv": 1}), ("priv_alltall", {"i8_priv": 1, "i8_rt": 20})]
if os.environ.get("I8P_ABLATE") or os.environ.get("I8P_VARIANTS"):
    # experiments build: "i8_variant" is the kernel template's VAR -- bits 0-3 ablations (1 no LDS-DMA, 2 no barrier, 4 no fragment reads, 8 no count
    # loads), 16 digit blocks through staging registers, 32 / 64 the filler schedules (64 = the release kernel), 128 one barrier per two k-steps
    if os.environ.get("I8P_VARIANTS"):
        for v in (0, 32, 64, 16, 128, 192):
            configs.append(("var%d_alltall" % v, {"i8_priv": 1, "i8_variant": v, "i8_rt": 20}))
    if os.environ.get("I8P_ABLATE"):
        for v, what in ((1, "no_dma"), (2, "no_barrier"), (4, "no_reads"), (8, "no_count_loads"), (5, "no_dma_no_reads"), (13, "no_dma_reads_loads"), (15, "mfma_only")):
            configs.append(("priv_" + what, {"i8_priv": 1, "i8_variant": 64 + v, "i8_rt": 20}))
def apply(opts):
    nm.set_option("i8_priv", 1); nm.set_option("i8_rt", 0)
    if nm.get_option("build_experiments"): nm.set_option("i8_variant", -1)
    for k, v in opts.items(): nm.set_option(k, v)
for B in Bs:
    apply({})
    ref = nm.bootstrap(min(B, 700), seed=1)[0]
    res = {n: {"gram_ms": [], "step_ms": []} for n, _ in configs}
    same = {}
    for rnd in range(4):
        for name, opts in (configs if rnd % 2 == 0 else configs[::-1]):
            apply(opts)
            if name not in same:
                same[name] = bool(np.array_equal(nm.bootstrap(min(B, 700), seed=1)[0], ref)) if "i8_variant" not in opts else None
            for w in range(3): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
            nm.sync()
            t = time.perf_counter()
            for k in range(20): nm.bootstrap_device(B, seed=1, rep_offset=(3 + k) * B)
            nm.sync()
            res[name]["step_ms"].append((time.perf_counter() - t) / 20 * 1e3)
            nm.profile(True); nm.profile_reset()
            for k in range(10): nm.bootstrap_device(B, seed=1, rep_offset=(30 + k) * B)
            nm.sync(); nm.profile(False)
            ms, n = nm.profile_read("gram")
            res[name]["gram_ms"].append(ms / n)
            res[name]["rt"] = nm.get_option("last_i8_rt"); res[name]["short_rows"] = nm.get_option("last_i8_short"); res[name]["mt"] = nm.get_option("last_i8_mt")
            res[name]["priv"] = nm.get_option("last_i8_priv"); res[name]["S"] = nm.get_option("last_i8_slices")
    for name, _ in configs:
        r = res[name]
        print(json.dumps({"B": B, "config": name, "S": r["S"], "priv": r["priv"], "rt": r["rt"], "short_rows": r["short_rows"], "count_tiles": r["mt"],
                          "gram_ms": [round(x, 4) for x in r["gram_ms"]], "gram_ms_min": round(min(r["gram_ms"]), 4), "step_ms": [round(x, 4) for x in r["step_ms"]],
                          "step_ms_min": round(min(r["step_ms"]), 4), "identical_rows": same.get(name)}), flush=True)
