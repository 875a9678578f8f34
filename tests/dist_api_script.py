"""Helper for tests/test_gpu_dist.py: run plain or under a process launcher (python -m torch.distributed.run is used as a mere
spawner; nothing here imports torch).  With RANK in the environment it joins the one-process-per-GPU job through
plspm.parallel.init_process_group() -- file rendezvous of the ncclUniqueId + ncclCommInitRank inside libplspm_hip.so -- so that
Plspm(bootstrap=True) takes the RCCL route (shard -> ONE all-gather -> device summaries); prints the summary frames on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
import pandas as pd  # noqa: E402

import plspm.config as c  # noqa: E402
from plspm import parallel  # noqa: E402
from plspm.mode import Mode  # noqa: E402
from plspm.plspm import Plspm  # noqa: E402
from plspm.scheme import Scheme  # noqa: E402

ctx = parallel.init_process_group() if "RANK" in os.environ else None
sat = pd.read_csv(os.path.join(ROOT, "tests", "golden", "ref_data", "satisfaction.csv"), index_col=0)
s = c.Structure()
s.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); s.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
s.add_path(["QUAL"], ["VAL", "SAT"]); s.add_path(["VAL"], ["SAT"]); s.add_path(["SAT"], ["LOY"])
cfg = c.Config(s.path(), scaled=False)
for lv in ["IMAG", "EXPE", "QUAL", "VAL", "SAT", "LOY"]:
    cfg.add_lv_with_columns_named(lv, Mode.A, sat, lv.lower())
m = Plspm(sat, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=300, processes=1, seed=11,
          device_id=ctx.local_rank if ctx else 0)
b = m.bootstrap()
if ctx is not None:
    assert b._via_group and b.ranks() == ctx.world and ctx.comm.uses_rccl, "the RCCL route was not taken"
    # a SECOND live Plspm(bootstrap=True) of the same job while the first object is still referenced (the job's one
    # communicator used to stay bound to the first object's group) -- same seed, same records
    m2 = Plspm(sat, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=300, processes=1, seed=11, device_id=ctx.local_rank)
    assert m2.bootstrap()._via_group
    if parallel.gather_to_root() and ctx.rank != 0:
        # PLSPM_GATHER=root: the records live on rank 0 only -- the summaries are everywhere (one broadcast of the table)
        assert (m2.bootstrap().weights().values == b.weights().values).all()
        try:
            b.replicates()
            raise SystemExit("a rank without records answered replicates()")
        except RuntimeError:
            pass
    else:
        assert (m2.bootstrap().replicates() == b.replicates()).all()
if (ctx.rank if ctx else 0) == 0:
    print("RESULT " + json.dumps({"weights": b.weights().values.tolist(), "paths": b.paths().values.tolist(),
                                  "r2": b.r_squared().values.tolist(), "loading": b.loading().values.tolist(),
                                  "total": b.total_effects().values.tolist(), "status_sum": int(b.status().sum()),
                                  "rows_sum": float(b.replicates().sum())}))
if ctx is not None:
    parallel.destroy_process_group()
    # the lazy accessors outlive the process group: the records were adopted by the fit's handle (no use of a released group)
    if not (parallel.gather_to_root() and ctx.rank != 0):
        assert int(b.status().sum()) == 0 and b.replicate_iterations().shape == (300,)
