"""Bootstrap throughput of a higher order construct model on Scale.ORD data (the reference's mobi data, 250 x 24, HOC Satisfaction =
{Image, Value}): both stages of every replicate batched on the device (round 2: one host-orchestrated two-stage
estimate per replicate, ~70 replicates/s).  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
import numpy as np
import pandas as pd
import plspm.config as c
import plspm.weights as w
from plspm.estimator import Estimator
from plspm.mode import Mode
from plspm.plspm import Plspm
from plspm.scale import Scale
from plspm.scheme import Scheme
mobi = pd.read_csv(os.path.join(ROOT, "tests", "golden", "ref_data", "mobi.csv"), index_col=0).astype(float)
out = {}
for scale_name, scale in (("ORD", Scale.ORD), ("NUM", Scale.NUM)):
    structure = c.Structure()
    structure.add_path(["Expectation", "Quality"], ["Satisfaction"])
    structure.add_path(["Satisfaction"], ["Complaints", "Loyalty"])
    config = c.Config(structure.path(), default_scale=scale)
    config.add_higher_order("Satisfaction", Mode.A, ["Image", "Value"])
    for lv, prefix in (("Expectation", "CUEX"), ("Quality", "PERQ"), ("Loyalty", "CUSL"), ("Image", "IMAG"), ("Complaints", "CUSCO"), ("Value", "PERV")):
        config.add_lv_with_columns_named(lv, Mode.A, mobi, prefix)
    observations = config.filter(mobi)
    calculator = w.WeightsCalculatorFactory(config, 100, 1e-7, np.sqrt(250 / 249), Scheme.PATH, 0)
    pair = Estimator(config).two_stage_bootstrap_handles(calculator, observations)
    for kv in filter(None, os.environ.get("HOC_BENCH_OPTS2", "").split(",")):      # set_option key=value on the SECOND stage's handle (A/B runs), e.g. HOC_BENCH_OPTS2=nm_threads=256
        pair.native._second.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    B = 5000
    for k in range(2): pair.native.bootstrap_device(B, seed=1, rep_offset=k * B)
    pair.native.sync()
    t0 = time.perf_counter()
    for k in range(3): pair.native.bootstrap_device(B, seed=1, rep_offset=(2 + k) * B)
    pair.native.sync()
    dt = (time.perf_counter() - t0) / 3
    rows, status, iters = pair.native.fetch(0, B)
    t1 = time.perf_counter()
    m = Plspm(mobi, config, Scheme.PATH, 100, 1e-7, bootstrap=True, bootstrap_iterations=B, seed=5)
    api = time.perf_counter() - t1
    out[scale_name] = {"replicates_per_s": round(B / dt, 1), "ms_per_%d" % B: round(dt * 1e3, 2), "ok_replicates": int((status == 0).sum()),
                       "stage2_iterations": [int(iters.min()), int(iters.max())], "api_Plspm_bootstrap_wall_ms": round(api * 1e3, 1), "api_used": int(m.bootstrap().used())}
    # the throughput regime: the late trips of a batch run at the latency of ONE problem's step whatever the batch holds, so a larger batch costs little more
    # (BASELINE configs[3] asks for 40,000 replicates)
    B2 = 40000
    pair.native.bootstrap_device(B2, seed=1, rep_offset=10 * B)
    pair.native.sync()
    t0 = time.perf_counter()
    pair.native.bootstrap_device(B2, seed=1, rep_offset=10 * B + B2)
    pair.native.sync()
    dt2 = time.perf_counter() - t0
    out[scale_name].update({"replicates_per_s_at_%d_per_call" % B2: round(B2 / dt2, 1), "ms_per_%d" % B2: round(dt2 * 1e3, 2)})
print(json.dumps({"workload": "mobi 250 x 24, 6 first-stage LVs, HOC Satisfaction = {Image, Value}, Mode A, PATH, 5000 replicates per call", **out}))
