"""Helper for tests/test_gpu_dist.py: run under torch.distributed.run; exercises the RCCL branch of plspm.bootstrap.Bootstrap
(device-resident gather + device summaries) and prints the summary frames as JSON on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
import pandas as pd  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import plspm.config as c  # noqa: E402
from plspm.mode import Mode  # noqa: E402
from plspm.plspm import Plspm  # noqa: E402
from plspm.scheme import Scheme  # noqa: E402

use_dist = "RANK" in os.environ
if use_dist:
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
sat = pd.read_csv(os.path.join(ROOT, "tests", "golden", "ref_data", "satisfaction.csv"), index_col=0)
s = c.Structure()
s.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); s.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
s.add_path(["QUAL"], ["VAL", "SAT"]); s.add_path(["VAL"], ["SAT"]); s.add_path(["SAT"], ["LOY"])
cfg = c.Config(s.path(), scaled=False)
for lv in ["IMAG", "EXPE", "QUAL", "VAL", "SAT", "LOY"]:
    cfg.add_lv_with_columns_named(lv, Mode.A, sat, lv.lower())
m = Plspm(sat, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=300, processes=1, seed=11,
          device_id=int(os.environ.get("LOCAL_RANK", "0")))
b = m.bootstrap()
if int(os.environ.get("RANK", "0")) == 0:
    print("RESULT " + json.dumps({"weights": b.weights().values.tolist(), "paths": b.paths().values.tolist(),
                                  "r2": b.r_squared().values.tolist(), "loading": b.loading().values.tolist(),
                                  "total": b.total_effects().values.tolist()}))
if use_dist:
    dist.barrier()
    dist.destroy_process_group()
