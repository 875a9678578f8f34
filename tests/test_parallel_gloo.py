"""N > 1 path on CPU: two real processes, gloo backend, world_size 2 (rendezvous on 127.0.0.1).
Checks the replicate sharding + the single all_gather of plspm.parallel against the single-process result,
including ragged shards and more ranks than replicates.  A deterministic stand-in plays the GPU shard runner
(no compute kernel is needed to test the exchange)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
from plspm import parallel  # noqa: E402

WIDTH = 11


def fake_shard(count, first):
    """Row r depends only on its global replicate id (like the Philox-keyed device stream)."""
    ids = np.arange(first, first + count)
    rows = np.sin(ids[:, None] * 0.37 + np.arange(WIDTH)[None, :]) * (1 + ids[:, None])
    status = (ids % 5 == 3).astype(np.int32)
    iters = (3 + ids % 4).astype(np.int32)
    return rows, status, iters


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, totals, out_dir):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        for total in totals:
            rows, status, iters = parallel.sharded_bootstrap(fake_shard, total, WIDTH)
            np.savez(os.path.join(out_dir, "r%d_t%d.npz" % (rank, total)), rows=rows, status=status, iters=iters)
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for total in (0, 1, 2, 7, 100, 5000, 40000):
        for world in (1, 2, 3, 4, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_single_process_passthrough():
    rows, status, iters = parallel.sharded_bootstrap(fake_shard, 9, WIDTH)
    ref = fake_shard(9, 0)
    assert np.array_equal(rows, ref[0]) and np.array_equal(status, ref[1]) and np.array_equal(iters, ref[2])


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_gather_matches_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    totals = (8, 7, 1, 30)          # even, ragged, fewer replicates than ranks, larger
    port = _free_port()
    mp.spawn(_worker, args=(world, port, totals, str(tmp_path)), nprocs=world, join=True)
    for total in totals:
        ref = fake_shard(total, 0)
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), "r%d_t%d.npz" % (rank, total)))
            assert np.array_equal(got["rows"], ref[0]), (total, rank)        # bit-exact, replicate-id order, on every rank
            assert np.array_equal(got["status"], ref[1]) and np.array_equal(got["iters"], ref[2])
