"""N > 1 path on CPU: real processes, gloo backend, world_size 2 and 3 (rendezvous on 127.0.0.1).

* replicate sharding + the single all_gather of plspm.parallel.sharded_bootstrap against the single-process result, including
  ragged shards and more ranks than replicates (a deterministic stand-in plays the shard runner);
* the same with the REAL solver source as shard runner: every rank draws its replicates' indices from the library's Philox
  stream (plspm_bootstrap_indices -- host code of libplspm_hip.so) and solves them with the CPU emulation build of
  csrc/solver_core.h; the merged rows must equal the single-process rows bit for bit and the oracle's to 1e-9;
* the file rendezvous that hands rank 0's ncclUniqueId to the other ranks of a one-process-per-GPU job."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from plspm import parallel  # noqa: E402

WIDTH = 11


def fake_shard(count, first):
    """Row r depends only on its global replicate id (like the Philox-keyed device stream)."""
    ids = np.arange(first, first + count)
    rows = np.sin(ids[:, None] * 0.37 + np.arange(WIDTH)[None, :]) * (1 + ids[:, None])
    status = (ids % 5 == 3).astype(np.int32)
    iters = (3 + ids % 4).astype(np.int32)
    return rows, status, iters


def _worker(rank, world, port, totals, out_dir):
    from helpers_dist import GlooComm
    comm = GlooComm(rank, world, port)
    try:
        for total in totals:
            rec = parallel.sharded_bootstrap(fake_shard, total, WIDTH, comm)
            rows, status, iters = parallel.split_records(rec, WIDTH)
            np.savez(os.path.join(out_dir, "r%d_t%d.npz" % (rank, total)), rows=rows, status=status, iters=iters)
    finally:
        comm.close()


def test_shard_range_partitions_exactly():
    for total in (0, 1, 2, 7, 100, 5000, 40000):
        for world in (1, 2, 3, 4, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_single_process_passthrough():
    rows, status, iters = parallel.split_records(parallel.sharded_bootstrap(fake_shard, 9, WIDTH), WIDTH)
    ref = fake_shard(9, 0)
    assert np.array_equal(rows, ref[0]) and np.array_equal(status, ref[1]) and np.array_equal(iters, ref[2])


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_gather_matches_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    from helpers_dist import free_port
    totals = (8, 7, 1, 30)          # even, ragged, fewer replicates than ranks, larger
    mp.spawn(_worker, args=(world, free_port(), totals, str(tmp_path)), nprocs=world, join=True)
    for total in totals:
        ref = fake_shard(total, 0)
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), "r%d_t%d.npz" % (rank, total)))
            assert np.array_equal(got["rows"], ref[0]), (total, rank)        # bit-exact, replicate-id order, on every rank
            assert np.array_equal(got["status"], ref[1]) and np.array_equal(got["iters"], ref[2])


# ---------------------------------------------------------------------------------------------- real solver source as the shard runner
N_EMU, SEED_EMU, TOTAL_EMU = 400, 21, 7


def _emu_inputs():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import plspm_oracle as orc
    C = orc.satisfaction_C()
    X, blocks = orc.synth(N_EMU, C, 3, seed=9)
    return orc, X, orc.Model(blocks, C, "ABABAB", "path", True)


def _emu_shard(count, first):
    """Replicates [first, first + count) of the library's Philox stream, solved by the CPU build of csrc/solver_core.h."""
    import ctypes
    import subprocess
    from plspm import _native
    from test_solver_hostemu import EMU, run_emu
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    lib = ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))
    _, X, model = _emu_inputs()
    shift = X[:, model.mv_order].mean(axis=0)
    rows, status, iters = [], [], []
    for r in range(first, first + count):
        idx = _native.bootstrap_indices(SEED_EMU, r, N_EMU)
        out = run_emu(lib, X, model, counts=np.bincount(idx, minlength=N_EMU), shift=shift, nthreads=2)
        rows.append(out["row"][:-2]); status.append(out["status"]); iters.append(out["iterations"])
    return np.array(rows), np.array(status, dtype=np.int32), np.array(iters, dtype=np.int32)


def _emu_worker(rank, world, port, out_dir):
    from helpers_dist import GlooComm
    comm = GlooComm(rank, world, port)
    try:
        width = _emu_shard(1, 0)[0].shape[1]
        rec = parallel.sharded_bootstrap(_emu_shard, TOTAL_EMU, width, comm)
        np.save(os.path.join(out_dir, "emu_r%d.npy" % rank), rec)
    finally:
        comm.close()


def test_gloo_world2_real_solver_shards_equal_single_process_and_oracle(tmp_path):
    import torch.multiprocessing as mp
    from helpers_dist import free_port
    mp.spawn(_emu_worker, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    single = parallel.join_records(*_emu_shard(TOTAL_EMU, 0))
    for rank in range(2):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "emu_r%d.npy" % rank)), single), rank
    # and the stream itself is the reference arithmetic: replicate 5 against the data-level oracle
    from plspm import _native
    orc, X, model = _emu_inputs()
    mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(SEED_EMU, 5, N_EMU), orc.correction(N_EMU))
    P, L = X.shape[1], model.L
    ne = (single.shape[1] - 2 - 2 * P - L) // 2
    inv = np.empty(P, dtype=np.int64); inv[model.mv_order] = np.arange(P)
    row = single[5]
    got = np.concatenate((row[:P][inv], row[P:P + L + 2 * ne], row[P + L + 2 * ne:2 * P + L + 2 * ne][inv]))
    assert its == int(row[-1]) and int(row[-2]) == 0
    np.testing.assert_allclose(got, mine, rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------------------------------------- ncclUniqueId rendezvous
def _rdzv_worker(rank, world, directory, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29999"
    got = []
    for k in range(2):                                   # two exchanges in a row: the sequence number keeps them apart
        got.append(parallel.exchange_unique_id(rank, world, directory, timeout=60.0, make_id=lambda k=k: bytes([k + 1]) * 128))
    with open(os.path.join(out_dir, "uid_r%d" % rank), "wb") as fh:
        fh.write(b"".join(got))


def test_unique_id_rendezvous_file_world3(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_rdzv_worker, args=(3, str(tmp_path), str(tmp_path)), nprocs=3, join=True)
    for rank in range(3):
        assert open(os.path.join(str(tmp_path), "uid_r%d" % rank), "rb").read() == b"\x01" * 128 + b"\x02" * 128


def test_rendezvous_directory_is_private_and_id_file_exclusive(tmp_path, monkeypatch):
    """the ncclUniqueId is a bearer token of the job's RCCL bootstrap.  Default directory: per user, mode 0700, refused when
    somebody else could write into it; the id file is created exclusively (a planted / stale file is an error, not an id)."""
    import stat
    import tempfile
    from plspm import _native
    monkeypatch.delenv("PLSPM_RDZV_DIR", raising=False)
    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    d = parallel._rendezvous_dir(None)
    assert d.startswith(str(tmp_path)) and stat.S_IMODE(os.stat(d).st_mode) == 0o700
    os.chmod(d, 0o777)
    with pytest.raises(_native.NativeBackendError, match="not private"):
        parallel._rendezvous_dir(None)
    os.chmod(d, 0o700)
    # a file already sitting at rank 0's path: refused (O_EXCL semantics), and removed names do not linger
    seq = parallel._rendezvous_seq
    planted = parallel._rendezvous_path(None)
    parallel._rendezvous_seq = seq
    with open(planted, "wb") as fh:
        fh.write(b"\x07" * 128)
    with pytest.raises(_native.NativeBackendError, match="already exists"):
        parallel.exchange_unique_id(0, 2, None, timeout=1.0, make_id=lambda: b"\x01" * 128)
    assert open(planted, "rb").read() == b"\x07" * 128 and not [n for n in os.listdir(d) if ".tmp" in n]
    os.remove(planted)
    # a reader does not pick up the left-over of an earlier job with the same launcher tag -- an id file carries its publication
    # time, and one published long before this rank arrived is ignored (rank 0 of the live job refuses the name and reports it)
    import struct
    import time
    seq = parallel._rendezvous_seq
    stale = parallel._rendezvous_path(None)
    parallel._rendezvous_seq = seq
    with open(stale, "wb") as fh:
        fh.write(b"\x09" * 128 + struct.pack("<d", time.time() - 3600.0))
    with pytest.raises(_native.NativeBackendError, match="did not publish"):
        parallel.exchange_unique_id(1, 2, None, timeout=0.3)
    os.remove(stale)
    seq = parallel._rendezvous_seq
    fresh = parallel._rendezvous_path(None)
    parallel._rendezvous_seq = seq
    with open(fresh, "wb") as fh:
        fh.write(b"\x0a" * 128 + struct.pack("<d", time.time()))
    assert parallel.exchange_unique_id(1, 2, None, timeout=5.0) == b"\x0a" * 128
    # a caller-supplied directory that another user owns and that is not sticky is refused
    foreign = tmp_path / "foreign"
    foreign.mkdir()
    real_stat = os.stat
    class FakeStat:
        def __init__(self, st): self.st_uid, self.st_mode = st.st_uid + 1, st.st_mode & ~0o1000
    monkeypatch.setattr(os, "stat", lambda p, *a, **k: FakeStat(real_stat(p)) if str(p) == str(foreign) else real_stat(p, *a, **k))
    with pytest.raises(_native.NativeBackendError, match="not sticky"):
        parallel._rendezvous_dir(str(foreign))


def test_multi_node_world_is_refused_with_a_clear_message(monkeypatch):
    from plspm import _native
    monkeypatch.setattr(parallel, "_context", None)
    monkeypatch.setattr(_native, "device_count", lambda: 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    with pytest.raises(_native.NativeBackendError, match="single-node only"):
        parallel.init_process_group(rank=3, world_size=16, local_rank=3)


def test_multi_gpu_sharding_is_opt_in(monkeypatch):
    """the reference's default processes=2 must not reach for device_id + 1.  GPUs are named by the caller (devices=) or by
    PLSPM_DEVICES; processes and MIN_REPLICATES_PER_GPU cap how many are taken; the handle's own device always comes first."""
    from plspm import _native
    monkeypatch.setattr(_native, "device_count", lambda: 8)
    monkeypatch.delenv("PLSPM_DEVICES", raising=False)
    assert parallel.devices_for(2, 40000, 3) == [3]
    assert parallel.devices_for(8, 40000, 0, devices=[4, 5, 0, 6]) == [0, 4, 5, 6]
    assert parallel.devices_for(2, 40000, 5, devices=[4, 5, 6]) == [5, 4]
    assert parallel.devices_for(8, 2500, 0, devices=range(8)) == [0, 1]            # 1,000 replicates per GPU at least
    monkeypatch.setenv("PLSPM_DEVICES", "0,1,2,3,4,5,6,7")
    assert parallel.devices_for(8, 40000, 0) == list(range(8))
    assert parallel.devices_for(2, 40000, 6) == [6, 0]
    with pytest.raises(ValueError, match="not among"):
        parallel.devices_for(4, 40000, 0, devices=[0, 9])
