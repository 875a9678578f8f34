"""GPU: the multi-GPU routes of the bootstrap, all through the C-ABI of libplspm_hip.so (plspm_comm_* / plspm_group_*).

The test box has ONE MI355X, so more than one rank means several handles on device 0: RCCL refuses duplicate devices and the
group then moves the records with device-to-device copies through exactly the same buffers, events and shard arithmetic; the
RCCL transport itself (dlopen, ncclCommInit*, the all-gather ordered behind the handle's stream) runs with one rank.
What must hold everywhere: the merged records are bit-identical to the single-handle stream for every number of ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import plspm_oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _run(cmd, env=None):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert lines, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[-1][7:])


def _model(n=2000, mvs=10, seed=3, device=0, nonmetric=False, modes=None):
    from plspm import _native
    C = orc.satisfaction_C()
    X, blocks = orc.synth(n, C, mvs, seed=seed)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.array(modes or [0] * 6, dtype=np.int32), 2, True, 100, 1e-6, device, nonmetric=nonmetric)
    nm.upload(X)
    return nm


def test_api_bootstrap_one_process_per_gpu_rccl_equals_single_process():
    """Plspm(bootstrap=True) in a launcher-spawned rank (RCCL, one rank) == the plain single-process result."""
    script = os.path.join(HERE, "dist_api_script.py")
    plain = _run([sys.executable, script])
    env_cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29533", script]
    dist = _run(env_cmd)
    for key in plain:
        np.testing.assert_allclose(np.array(dist[key], dtype=float), np.array(plain[key], dtype=float), rtol=1e-12, atol=1e-14, err_msg=key)
    # round 6: the gather to rank 0 (PLSPM_GATHER=root; with one rank: ncclSend / ncclRecv to itself) gives the same frames
    env_cmd[env_cmd.index("29533")] = "29535"
    dist = _run(env_cmd, env=dict(os.environ, PLSPM_GATHER="root"))
    for key in plain:
        np.testing.assert_allclose(np.array(dist[key], dtype=float), np.array(plain[key], dtype=float), rtol=1e-12, atol=1e-14, err_msg=key)


@pytest.mark.parametrize("nranks,B", [(1, 64), (2, 64), (2, 65), (3, 100), (4, 3)])
def test_group_records_bit_identical_to_single_handle(nranks, B):
    """G handles (one device) = G ranks of the group: shards + one gather give the single-handle stream, bit for bit, for even,
    ragged and more-ranks-than-replicates splits; the device summary of the gathered records equals the single-handle summary."""
    from plspm import _native
    models = [_model() for _ in range(nranks)]
    ref_rows, ref_status, ref_iters = models[0].bootstrap(B, seed=5, rep_offset=7)
    original = np.linspace(-1.0, 1.0, models[0].row_width)
    ref_table, ref_used = models[0].summary(B, original)
    comm = _native.NativeComm([0] * nranks)
    assert comm.uses_rccl == (nranks == 1)                   # one rank: the real RCCL transport; several on one GPU: device copies
    group = _native.NativeGroup(comm, models)
    for _ in range(3):                                       # both buffer slots, and a slot re-used
        group.bootstrap(B, seed=5, rep_offset=7)
    rows, status, iters = group.rows()
    assert np.array_equal(rows, ref_rows) and np.array_equal(status, ref_status) and np.array_equal(iters, ref_iters)
    table, used = group.summary(original)
    assert used == ref_used == B
    assert np.array_equal(table, ref_table)
    ptr, n_rec, stride = group.records(0)
    assert n_rec == nranks * ((B + nranks - 1) // nranks) and stride == models[0].row_stride and ptr
    starts = [group.shard(B, r) for r in range(nranks)]
    assert starts[0][0] == 0 and sum(c for _, c in starts) == B
    group.barrier()
    assert group.max(1.25) == 1.25
    group.close(); comm.close()


@pytest.mark.parametrize("nranks,B,chunks", [(1, 64, 1), (2, 65, 1), (3, 5003, 3), (4, 9000, 4), (4, 3, 1)])
def test_group_gather_to_root_is_bit_identical_and_only_root_holds_records(nranks, B, chunks):
    """Group option "gather_root" (round 6: only the rank whose handle summarises receives the shards -- the reference's own merge, bootstrap.py:96-111; 1 / nranks of
    the all-gather's bytes): rows, status, iteration counts, summary and the adopted handle are bit for bit those of the all-gather and of the single-handle
    stream, for every sub-batch cut; the other local handles hold no records and say so.  One rank = the real RCCL transport (ncclSend / ncclRecv to itself in
    one group call); several ranks on this one GPU = the same-device copy launch with one destination."""
    from plspm import _native
    models = [_model(1500, 5, seed=9) for _ in range(nranks)]
    ref_rows, ref_status, ref_iters = models[0].bootstrap(B, seed=5, rep_offset=11)
    original = np.linspace(-1.0, 1.0, models[0].row_width)
    ref_table, ref_used = models[0].summary(B, original)
    comm = _native.NativeComm([0] * nranks)
    group = _native.NativeGroup(comm, models)
    assert group.first_rank == 0
    group.set_option("chunks", chunks); group.set_option("chunk_align", 64)
    group.set_option("gather_root", 1)
    for _ in range(3):                                       # both buffer slots, and a slot re-used
        group.bootstrap(B, seed=5, rep_offset=11)
    rows, status, iters = group.rows()
    assert np.array_equal(rows, ref_rows) and np.array_equal(status, ref_status) and np.array_equal(iters, ref_iters)
    table, used = group.summary(original)
    assert used == ref_used and np.array_equal(table, ref_table, equal_nan=True)
    assert group.records(0)[0]
    for local in range(1, nranks):
        with pytest.raises(_native.NativeBackendError, match="gathered to rank 0 only"):
            group.records(local)
    group.set_option("gather_root", 0)                       # ... and back: the all-gather fills every handle's buffer again
    group.bootstrap(B, seed=5, rep_offset=11)
    assert all(group.records(local)[0] for local in range(nranks))
    rows2, status2, iters2 = group.rows()
    assert np.array_equal(rows2, ref_rows) and np.array_equal(iters2, ref_iters)
    group.set_option("gather_root", 1)
    group.bootstrap(B, seed=5, rep_offset=11)
    group.adopt()
    got = models[0].fetch(0, B)
    assert np.array_equal(got[0], ref_rows) and np.array_equal(got[1], ref_status)
    group.close(); comm.close()


def test_plspm_api_gather_to_root_env(monkeypatch):
    """PLSPM_GATHER=root through the public API (one process, two handles on this GPU): the frames of the default all-gather."""
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    import pandas as pd
    X, blocks = orc.synth(1200, orc.satisfaction_C(), 4, seed=3)
    cols = ["%s%d" % (lv.lower(), k) for lv in orc.SAT_LVS for k in range(4)]
    frame = pd.DataFrame(X, columns=cols)
    structure = c.Structure()
    for frm, to in orc.SAT_EDGES:
        structure.add_path([frm], [to])

    def run():
        cfg = c.Config(structure.path(), scaled=True)
        for lv in orc.SAT_LVS:
            cfg.add_lv_with_columns_named(lv, Mode.A, frame, lv.lower())
        m = Plspm(frame, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=1200, processes=2, seed=6)
        return m.bootstrap().weights(), m.bootstrap().paths(), m.bootstrap().ranks(), m.bootstrap().replicates()
    from plspm import parallel
    monkeypatch.setattr(parallel, "devices_for", lambda processes, replicates, first_device=0, devices=None: [0] * min(int(processes), 2))
    base = run()
    monkeypatch.setenv("PLSPM_GATHER", "root")
    root = run()
    assert base[2] == root[2] == 2
    pd.testing.assert_frame_equal(base[0], root[0], check_exact=True)
    pd.testing.assert_frame_equal(base[1], root[1], check_exact=True)
    assert np.array_equal(base[3], root[3])
    monkeypatch.setenv("PLSPM_GATHER", "sideways")
    with pytest.raises(ValueError, match="PLSPM_GATHER"):
        run()


@pytest.mark.parametrize("nranks,B,chunks", [(2, 3000, 2), (2, 3001, 3), (3, 5003, 3), (1, 2600, 2), (2, 4500, 0), (4, 9000, 4)])
def test_group_call_as_sub_batches_is_bit_identical(nranks, B, chunks):
    """ONE plspm_group_bootstrap call as sub-batches (round 5: the gather of sub-batch k beside the kernels of sub-batch k + 1): consecutive ranges
    of the replicate ids, each sharded over the ranks like a call of its own.  Rows, status, iteration counts and the device summary are bit
    for bit those of the single-handle stream for even, ragged and automatic cuts; the plan covers the call; the gathered buffer holds one
    block of nranks x ceil(count / nranks) records per sub-batch."""
    from plspm import _native
    models = [_model(1500, 5, seed=9) for _ in range(nranks)]
    ref_rows, ref_status, ref_iters = models[0].bootstrap(B, seed=5, rep_offset=11)
    original = np.linspace(-1.0, 1.0, models[0].row_width)
    ref_table, ref_used = models[0].summary(B, original)
    comm = _native.NativeComm([0] * nranks)
    group = _native.NativeGroup(comm, models)
    group.set_option("chunks", chunks)
    units = models[0].get_option("boot_round_units")        # replicates of one round of the device for this model (a small model: thousands)
    assert units >= 64 and len(group.plan(B)) <= max(1, -(-B // (units * nranks)))      # never more sub-batches than (started) rounds per rank
    group.set_option("chunk_align", 64)                      # ... so the cuts of this test are asked for in count tiles
    plan = group.plan(B)
    assert plan[0][0] == 0 and sum(c for _, c in plan) == B and all(plan[k][0] + plan[k][1] == plan[k + 1][0] for k in range(len(plan) - 1))
    if chunks:
        assert 2 <= len(plan) <= chunks, plan               # (never more sub-batches than whole rounds of the device per rank)
    elif nranks > 1:
        assert len(plan) == 1, plan                         # ranks that share a device exchange with one copy launch: nothing worth hiding
    assert all(c % (64 * nranks) == 0 for _, c in plan[:-1]), plan
    for _ in range(3):                                       # both buffer slots, and a slot re-used
        group.bootstrap(B, seed=5, rep_offset=11)
    rows, status, iters = group.rows()
    assert np.array_equal(rows, ref_rows) and np.array_equal(status, ref_status) and np.array_equal(iters, ref_iters)
    table, used = group.summary(original)
    assert used == ref_used == B and np.array_equal(table, ref_table, equal_nan=True)
    _, n_rec, _ = group.records(0)
    assert n_rec == nranks * sum((c + nranks - 1) // nranks for _, c in plan)
    group.adopt()                                            # replicate-id order in the handle's own buffer
    got = models[0].fetch(0, B)
    assert np.array_equal(got[0], ref_rows) and np.array_equal(got[2], ref_iters)
    # a one-sub-batch call on the same group afterwards (the buffers are re-used with another layout)
    group.set_option("chunks", 1)
    group.bootstrap(B, seed=5, rep_offset=11)
    assert np.array_equal(group.rows()[0], ref_rows) and len(group.plan(B)) == 1
    with pytest.raises(_native.NativeBackendError):
        group.set_option("chunks", 9)
    group.close(); comm.close()


@pytest.mark.parametrize("max_channels", [1, 4])
def test_rccl_communicator_with_a_channel_cap_and_its_split(max_channels):
    """plspm_comm_create_ex(max_channels) / plspm_comm_split: RCCL communicators whose collectives run at most that many workgroups
    (ncclCommInitRankConfig / ncclCommSplit with ncclConfig_t.maxCTAs) -- one rank here (the box has one GPU): the all-gather, the barrier and
    the max go through RCCL and the records are the single-handle stream."""
    from plspm import _native
    nm = _model(1200, 5, seed=6)
    ref = nm.bootstrap(700, seed=3)
    capped = _native.NativeComm([0], transport="rccl", max_channels=max_channels)
    assert capped.uses_rccl and capped.transport == "rccl" and capped.max_channels == max_channels and capped.create_s > 0
    group = _native.NativeGroup(capped, [nm])
    for _ in range(3):
        group.bootstrap(700, seed=3)
    assert np.array_equal(group.rows()[0], ref[0])
    group.barrier()
    assert group.max(2.5) == 2.5
    group.close()
    twin = capped.split(0)                                   # RCCL's default channel count, split off the capped one
    assert twin.uses_rccl and twin.max_channels == 0 and twin.nranks == 1
    g2 = _native.NativeGroup(twin, [nm])
    g2.bootstrap(700, seed=3)
    assert np.array_equal(g2.rows()[0], ref[0])
    g2.close(); twin.close(); capped.close()
    with pytest.raises(_native.NativeBackendError):          # ranks sharing a device cannot be an RCCL communicator
        _native.NativeComm([0, 0], transport="rccl")
    shared = _native.NativeComm([0, 0], transport="copy")    # ... and the copy transport on one device is the one-launch exchange
    assert not shared.uses_rccl and shared.transport == "device-copies"
    shared.close()


def test_group_exchange_of_odd_sized_record_blocks():
    """Handles on one device exchange their records in ONE launch (16-byte pieces when a rank's block is a whole number of them, 8-byte
    ones otherwise): a model whose record stride is odd, with an odd number of replicates per rank."""
    from plspm import _native
    C = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0]], dtype=float)      # record stride = 2 P + L + 2 effects + 2: odd with an odd number of LVs
    X, blocks = orc.synth(500, C, 4, seed=11)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    models = []
    for _ in range(3):
        nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(3, dtype=np.int32), 2, True, 100, 1e-6, 0)
        nm.upload(X); models.append(nm)
    B = 21                                                   # 7 replicates per rank
    assert models[0].row_stride % 2 == 1 and ((B + 2) // 3) % 2 == 1, models[0].row_stride
    ref_rows, ref_status, ref_iters = models[0].bootstrap(B, seed=2, rep_offset=1)
    comm = _native.NativeComm([0] * 3)
    group = _native.NativeGroup(comm, models)
    for _ in range(2): group.bootstrap(B, seed=2, rep_offset=1)
    rows, status, iters = group.rows()
    assert np.array_equal(rows, ref_rows) and np.array_equal(status, ref_status) and np.array_equal(iters, ref_iters)
    group.close(); comm.close()


def test_group_growing_and_shrinking_batches_and_a_second_group_on_the_same_comm():
    from plspm import _native
    models = [_model(800, 4, seed=8) for _ in range(2)]
    comm = _native.NativeComm([0, 0])
    group = _native.NativeGroup(comm, models)
    for B in (10, 300, 31, 300):
        group.bootstrap(B, seed=2)
        rows, status, _ = group.rows()
        ref = models[1].bootstrap(B, seed=2)
        assert np.array_equal(rows, ref[0]) and np.array_equal(status, ref[1])
    with pytest.raises(_native.NativeBackendError):          # a communicator serves one group at a time
        _native.NativeGroup(comm, models)
    group.close()
    again = _native.NativeGroup(comm, models[::-1])
    again.bootstrap(12, seed=2)
    assert np.array_equal(again.rows()[0], models[0].bootstrap(12, seed=2)[0])
    again.close(); comm.close()


def test_group_nonmetric_two_ranks_threads():
    """Non-metric models iterate with host read-backs: the group drives each handle from its own host thread."""
    from plspm import _native
    models = [_model(1500, 5, seed=4, nonmetric=True, modes=[0, 1, 0, 1, 0, 1]) for _ in range(2)]
    ref = models[0].bootstrap(90, seed=13)
    comm = _native.NativeComm([0, 0])
    group = _native.NativeGroup(comm, models)
    group.bootstrap(90, seed=13)
    rows, status, iters = group.rows()
    assert np.all(status == 0) and np.array_equal(iters, ref[2])
    assert np.array_equal(rows, ref[0])
    group.close(); comm.close()


def _gloo_worker(rank, world, port, total, out_dir):
    """Both ranks on GPU 0; the shard runner is the product's NativeModel.bootstrap, the transport a host-side gloo gather."""
    from helpers_dist import GlooComm
    from plspm import parallel
    comm = GlooComm(rank, world, port)
    try:
        nm = _model(1200, 5, seed=6)
        rec = parallel.sharded_bootstrap(lambda count, first: nm.bootstrap(count, 17, first), total, nm.row_width, comm)
        nm.store(rec)
        table, used = nm.summary(total, np.ones(nm.row_width))
        np.savez(os.path.join(out_dir, "g%d.npz" % rank), rec=rec, table=table, used=used)
    finally:
        comm.close()


def test_gloo_world2_real_cabi_shards_equal_single_process(tmp_path):
    import torch.multiprocessing as mp
    from helpers_dist import free_port
    total = 77
    mp.spawn(_gloo_worker, args=(2, free_port(), total, str(tmp_path)), nprocs=2, join=True)
    nm = _model(1200, 5, seed=6)
    rows, status, iters = nm.bootstrap(total, 17, 0)
    table, used = nm.summary(total, np.ones(nm.row_width))
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "g%d.npz" % rank))
        assert np.array_equal(got["rec"][:, :-2], rows) and np.array_equal(got["rec"][:, -2], status) and np.array_equal(got["rec"][:, -1], iters)
        assert int(got["used"]) == used and np.array_equal(got["table"], table)


def test_config4_40000_replicates_properties_and_sharding_invariance():
    """BASELINE.json configs[3] on the one GPU of the box: 40,000 replicates of the 10k x 60 x 6 PATH model, as 8 shards of
    5,000 (the 8-GPU split, here 2 handles x 4 sequential offsets and a 2-rank group) against ONE 40,000-replicate call.
    Size-independent properties: every replicate converges in the same 3 iterations as the oracle's, shards are bit-identical to
    the monolithic stream, summaries of the gathered records equal the single-call summaries, and spot rows match the oracle."""
    from plspm import _native
    a, b = _model(10000, 10, seed=0), _model(10000, 10, seed=0)
    B, seed = 40000, 1
    a.bootstrap_device(B, seed=seed)
    original = np.ones(a.row_width)
    table, used = a.summary(B, original)
    assert used == B
    mono = a.fetch(0, B)
    assert np.all(mono[1] == 0) and mono[2].min() == 3 and mono[2].max() <= 4
    comm = _native.NativeComm([0, 0])
    group = _native.NativeGroup(comm, [a, b])
    group.bootstrap(B, seed=seed)                           # 2 ranks x 20,000
    g_table, g_used = group.summary(original)
    assert g_used == B and np.array_equal(g_table, table, equal_nan=True)       # (t stat. of a zero-variance column, R2 of an exogenous LV, is inf/nan)
    g_rows = group.rows()
    assert np.array_equal(g_rows[0], mono[0]) and np.array_equal(g_rows[2], mono[2])
    for shard in (0, 3, 7):                                 # the 8-GPU shards of 5,000 as separate calls
        part = b.bootstrap(5000, seed=seed, rep_offset=5000 * shard)
        assert np.array_equal(part[0], mono[0][5000 * shard:5000 * (shard + 1)])
    group.close(); comm.close()
    X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "path", True)
    for r in (0, 19999, 20000, 39999):
        mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(seed, r, 10000), orc.correction(10000))
        assert its == mono[2][r]
        np.testing.assert_allclose(mono[0][r], mine, rtol=1e-8, atol=1e-11)
    # the summary statistic itself against NumPy on the fetched rows
    np.testing.assert_allclose(table[:, 1], mono[0].mean(axis=0), rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(table[:, 2], mono[0].std(axis=0, ddof=1), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(table[:, 3], np.quantile(mono[0], 0.025, axis=0), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(table[:, 4], np.quantile(mono[0], 0.975, axis=0), rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("kind", ["metric", "nonmetric", "ordinal"])
def test_plspm_processes_shards_over_handles_with_identical_results(kind, monkeypatch):
    """Plspm(..., processes=k) = k GPUs of this process (reference: k forked workers, plspm.py:35-37, bootstrap.py:89-94).  The box has
    one GPU, so the device list is forced to [0, 0]: the fit's handle builder replicates model + data on the 'second GPU', the two
    handles form a group, and every summary frame equals the single-handle result (same seed)."""
    import pandas as pd
    import plspm.config as c
    from plspm import parallel
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    from helpers import satisfaction_frame
    sat = satisfaction_frame()

    def run(processes):
        s = c.Structure()
        s.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); s.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
        s.add_path(["QUAL"], ["VAL", "SAT"]); s.add_path(["VAL"], ["SAT"]); s.add_path(["SAT"], ["LOY"])
        cfg = c.Config(s.path(), scaled=True, default_scale={"nonmetric": Scale.NUM, "ordinal": Scale.ORD}.get(kind))      # (ordinal: the categorical solver, host read-backs inside a call)
        for lv in ["IMAG", "EXPE", "QUAL", "VAL", "SAT", "LOY"]:
            cfg.add_lv_with_columns_named(lv, Mode.A, sat, lv.lower())
        return Plspm(sat, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=2400, processes=processes, seed=21).bootstrap()
    single = run(1)
    assert single.ranks() == 1
    assert run(2).ranks() == 1              # processes alone never reaches for a second GPU: sharding is opt-in (devices= / PLSPM_DEVICES)
    monkeypatch.setattr(parallel, "devices_for", lambda processes, replicates, first_device=0, devices=None: [0] * min(int(processes), 2))
    double = run(2)
    # the gathered records were adopted by the fit's handle and the group is gone: the communicator is free for the next bootstrap
    assert double.ranks() == 2 and double._group is None and double._helpers is None
    comm = parallel.local_comm([0, 0])
    assert not comm.busy()
    for name in ("weights", "r_squared", "total_effects", "paths", "loading"):
        a, b = getattr(single, name)(), getattr(double, name)()
        assert list(a.index) == list(b.index)
        np.testing.assert_array_equal(a.values, b.values, err_msg=name)
    assert np.array_equal(single.replicates(), double.replicates(), equal_nan=True) and single.used() == double.used() and (single.used() == 2400 or kind == "ordinal")
    # a second multi-GPU bootstrap while the first object is alive re-uses the cached communicator (it used to be refused
    # or to need a communicator of its own), and both objects still answer their lazy accessors
    again = run(2)
    assert again.ranks() == 2 and parallel.local_comm([0, 0]) is comm
    assert np.array_equal(again.replicates(), double.replicates(), equal_nan=True)
    assert np.array_equal(double.status(), single.status()) and np.array_equal(again.replicate_iterations(), single.replicate_iterations())


def test_real_multi_gpu_rccl_all_gather_when_the_box_has_two_gpus():
    """Skipped on the one-GPU test box.  With >= 2 MI355X: ncclCommInitAll over DISTINCT devices, one handle per GPU, ONE ncclAllGather
    over xGMI -- the records are bit-identical to the single-handle stream (even, ragged and multi-call splits), the transport is RCCL
    on every rank, and the one-process-per-GPU launcher route gives the same frames (reference fan-out: bootstrap.py:89-111)."""
    from plspm import _native
    G = min(_native.device_count(), 8)
    if G < 2:
        pytest.skip("needs at least two GPUs (the driver's multi-GPU box)")
    models = [_model(device=d) for d in range(G)]
    ref = models[0].bootstrap(1003, seed=5, rep_offset=7)
    comm = _native.NativeComm(list(range(G)))
    assert comm.uses_rccl and comm.nranks == G
    group = _native.NativeGroup(comm, models)
    for call in range(3):                                       # the double-buffered slots: three calls in a row, then the last one's records
        group.bootstrap(1003, seed=5, rep_offset=7)
    rows, status, iters = group.rows()
    assert np.array_equal(rows, ref[0]) and np.array_equal(status, ref[1]) and np.array_equal(iters, ref[2])
    covered = [group.shard(1003, r) for r in range(G)]
    assert covered[0][0] == 0 and sum(c for _, c in covered) == 1003
    original = np.linspace(-1.0, 1.0, models[0].row_width)
    table, used = group.summary(original)
    ref_table, ref_used = models[0].summary(1003, original)
    assert used == ref_used == 1003 and np.array_equal(table, ref_table)
    group.barrier()
    assert group.max(3.5) == 3.5
    group.close(); comm.close()
    # every transport (round 5): RCCL with a channel cap, a capped communicator split off the default one, the copy-engine exchange on
    # peer-mapped buffers -- each with one sub-batch and with three, against the same single-handle stream
    for make in (lambda: _native.NativeComm(list(range(G)), transport="rccl", max_channels=4),
                 lambda: _native.NativeComm(list(range(G)), transport="copy")):
        comm = make()
        assert comm.nranks == G and comm.transport == ("rccl" if comm.uses_rccl else "copy-engines")
        comms = [comm] + ([comm.split(8)] if comm.uses_rccl else [])
        for cc in comms:
            group = _native.NativeGroup(cc, models)
            for chunks, root in ((1, 0), (3, 0), (1, 1), (3, 1)):      # (round 6: ... and the gather to rank 0 -- ncclSend / ncclRecv, or rank 0's copy engines alone)
                group.set_option("chunks", chunks)
                group.set_option("gather_root", root)
                for call in range(3):
                    group.bootstrap(8 * 1003, seed=5, rep_offset=7)
                rows, status, iters = group.rows()
                big = models[0].bootstrap(8 * 1003, seed=5, rep_offset=7)
                assert np.array_equal(rows, big[0]) and np.array_equal(status, big[1]) and np.array_equal(iters, big[2]), (cc.transport, cc.max_channels, chunks, root)
                if root:
                    with pytest.raises(_native.NativeBackendError, match="gathered to rank 0 only"):
                        group.records(1)
            group.close()
        for cc in comms[::-1]:
            cc.close()
    script = os.path.join(HERE, "dist_api_script.py")
    plain = _run([sys.executable, script])
    dist = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(min(G, 2)), "--master-addr", "127.0.0.1",
                 "--master-port", "29537", script])
    for key in plain:
        np.testing.assert_allclose(np.array(dist[key], dtype=float), np.array(plain[key], dtype=float), rtol=1e-12, atol=1e-14, err_msg=key)

    # ... and the launcher route with PLSPM_GATHER=root: every rank gets the summary table through the broadcast, rank 0 prints the frames
    dist_root = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(min(G, 2)), "--master-addr", "127.0.0.1",
                      "--master-port", "29539", script], env=dict(os.environ, PLSPM_GATHER="root"))
    for key in plain:
        np.testing.assert_allclose(np.array(dist_root[key], dtype=float), np.array(plain[key], dtype=float), rtol=1e-12, atol=1e-14, err_msg=key)


def test_bench_multi_rank_path_on_one_device_calibrates_the_tile_plan(tmp_path):
    """bench.py --gpus 2 with both ranks on device 0 (test seam PLSPM_BENCH_SHARED_DEVICE): the multi-rank branch of the bench -- group steps,
    max-over-ranks timing, the tile-plan calibration ("gram_tile_plan_cus") -- runs and prints its one JSON line."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, PLSPM_BENCH_SHARED_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--reps-per-gpu", "1000"], capture_output=True, text=True,
                         timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["config"]["replicates_per_step"] == 2000 and d["config"]["transport"] == "device-copies"
    plan = d["config"]["gram_tile_plan_cus"]
    assert set(plan["tried_ms_per_step"]) == {"0", "248", "240", "232", "224", "208"} and plan["chosen"] in (0, 248, 240, 232, 224, 208)


@pytest.mark.parametrize("scale_name", ["NUM", "ORD"])
def test_plspm_processes_shards_a_hoc_model_with_identical_results(scale_name, monkeypatch):
    """Plspm(..., processes=2) on a higher order construct (both stages of every replicate on the device; on ordinal items the categorical solver with its host read-backs
    inside a call): two handle PAIRS on one device form the group, and the replicates are the bits of the single-pair run."""
    import pandas as pd
    import plspm.config as c
    from helpers import GOLDEN
    from plspm import parallel
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    mobi = pd.read_csv(os.path.join(GOLDEN, "ref_data", "mobi.csv"), index_col=0).astype(float)

    def run(processes):
        structure = c.Structure()
        structure.add_path(["Expectation", "Quality"], ["Satisfaction"])
        structure.add_path(["Satisfaction"], ["Complaints", "Loyalty"])
        config = c.Config(structure.path(), default_scale=getattr(Scale, scale_name))
        config.add_higher_order("Satisfaction", Mode.A, ["Image", "Value"])
        for lv, prefix in (("Expectation", "CUEX"), ("Quality", "PERQ"), ("Loyalty", "CUSL"), ("Image", "IMAG"), ("Complaints", "CUSCO"), ("Value", "PERV")):
            config.add_lv_with_columns_named(lv, Mode.A, mobi, prefix)
        return Plspm(mobi, config, Scheme.PATH, 100, 1e-7, bootstrap=True, bootstrap_iterations=600, processes=processes, seed=5).bootstrap()
    single = run(1)
    monkeypatch.setattr(parallel, "devices_for", lambda processes, replicates, first_device=0, devices=None: [0] * min(int(processes), 2))
    double = run(2)
    assert single.ranks() == 1 and double.ranks() == 2
    assert single.used() == double.used() >= 590
    assert np.array_equal(single.replicates(), double.replicates(), equal_nan=True)
    assert np.array_equal(single.status(), double.status()) and np.array_equal(single.replicate_iterations(), double.replicate_iterations())
