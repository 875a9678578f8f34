"""Replicate sharding across GPUs: no data-path collective, ONE gather at the end.  No torch anywhere.

Replicates are independent (reference bootstrap.py:54-66 has no cross-replicate state; the reference shards them over
``processes`` forked workers, bootstrap.py:89-94, and merges through a Queue, bootstrap.py:96-111).  Here rank g runs the
contiguous replicate-id range ``shard_range(B, g, G)`` of one logical Philox stream keyed by (seed, replicate id), so the merged
result is identical for every G, and the Queue becomes a single RCCL all-gather over xGMI inside libplspm_hip.so
(``plspm_group_*``, include/plspm_hip.h).  Three ways to run more than one shard:

* **one process, several GPUs** -- what ``Plspm(..., processes=k)`` does: ``local_comm(devices)`` (ncclCommInitAll, cached per
  process) + one handle per device bound into a ``NativeGroup``;
* **one process per GPU** (``python -m torch.distributed.run`` / mpirun; the launcher is only a process spawner, torch is not
  imported): ``init_process_group()`` reads RANK / WORLD_SIZE / LOCAL_RANK, rank 0 publishes the ncclUniqueId through a
  rendezvous file, every rank joins with ncclCommInitRank; ``Plspm`` then shards its bootstrap over the job automatically;
* **a host-side transport of the caller's** (MPI, gloo, ...): ``sharded_bootstrap(run_shard, total, width, comm)`` with any
  object offering ``rank``, ``world`` and ``all_gather(ndarray) -> ndarray``; the merged records go back to the device for the
  summaries (``NativeModel.store``).  The CPU test-suite drives this with a gloo communicator.
"""
import os
import tempfile
import time

import numpy as np

MIN_REPLICATES_PER_GPU = 1000      # below this a second GPU costs more (upload + launch) than it saves; results do not depend on G


def shard_range(total, rank, world):
    """Contiguous, balanced [start, stop) of replicate ids for ``rank`` (same split as plspm_group_shard)."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def split_records(records, width):
    """[row | status | iterations] records -> (rows, status int32, iters int32)."""
    records = np.asarray(records)
    return np.ascontiguousarray(records[:, :width]), records[:, width].astype(np.int32), records[:, width + 1].astype(np.int32)


def join_records(rows, status, iters):
    return np.concatenate((rows, status[:, None].astype(np.float64), iters[:, None].astype(np.float64)), axis=1)


# ------------------------------------------------------------------------------------------------ one process, several GPUs
_local_comms = {}


def gather_to_root() -> bool:
    """``PLSPM_GATHER=root``: the shards travel to rank 0 only (group option "gather_root": 1 / nranks of the all-gather's bytes, the summary table
    broadcast back); default ``all``: every rank receives every shard (ONE ncclAllGather per sub-batch).  Every rank of a job must see the same value."""
    mode = os.environ.get("PLSPM_GATHER", "all").lower()
    if mode not in ("all", "root"):
        raise ValueError("PLSPM_GATHER must be 'all' or 'root' (got %r)" % mode)
    return mode == "root"


def local_comm(devices, transport=None, max_channels=None):
    """The process-wide communicator over ``devices`` (created on first use: librccl load + ncclCommInitAll take of the order
    of a second, every later bootstrap re-uses it).  A communicator serves one group at a time: while the bootstrap of an earlier
    ``Plspm`` object is still alive on the cached one, the caller gets a communicator of its own (which becomes the cached one).
    ``transport`` ("auto" / "rccl" / "copy": ``PLSPM_TRANSPORT``) and ``max_channels`` (``PLSPM_RCCL_MAX_CHANNELS``; 0 = RCCL's
    default) pick how the records travel -- see ``_native.NativeComm``; the rows do not depend on either."""
    from plspm import _native
    transport = transport or os.environ.get("PLSPM_TRANSPORT", "auto")
    max_channels = int(os.environ.get("PLSPM_RCCL_MAX_CHANNELS", "0")) if max_channels is None else int(max_channels)
    key = (tuple(int(d) for d in devices), transport, max_channels)
    comm = _local_comms.get(key)
    if comm is None or not comm._h or comm.busy():
        comm = _native.NativeComm(key[0], transport=transport, max_channels=max_channels)
        _local_comms[key] = comm
    return comm


def allowed_devices():
    """The GPUs this process may shard a bootstrap over, from ``PLSPM_DEVICES`` (comma-separated HIP device ids), or None when the
    variable is unset -- sharding over several GPUs of one process is OPT-IN: a caller that pinned ``device_id`` on a shared node
    must not find its replicates (and 570 MB of librccl) on a neighbouring GPU that belongs to another job."""
    spec = os.environ.get("PLSPM_DEVICES", "").strip()
    if not spec:
        return None
    return [int(tok) for tok in spec.replace(";", ",").split(",") if tok.strip() != ""]


def devices_for(processes, replicates, first_device=0, devices=None):
    """GPUs a single-process bootstrap uses.  ``devices`` (the ``Plspm(devices=[...])`` argument) or, without it, the
    ``PLSPM_DEVICES`` allow-list names the candidates; ``processes`` (the reference's worker count, plspm.py:35-37) and
    MIN_REPLICATES_PER_GPU cap how many of them are taken, in the order given, ``first_device`` always first.  With neither the
    answer is ``[first_device]``: the reference's default ``processes=2`` alone never reaches for a second GPU.  Purely a
    performance decision -- the rows are the same for every answer."""
    from plspm import _native
    first = int(first_device)
    candidates = list(devices) if devices is not None else allowed_devices()
    if not candidates:
        return [first]
    available = _native.device_count()
    ordered = [first] + [int(d) for d in candidates if int(d) != first]
    for d in ordered:
        if not (0 <= d < available):
            raise ValueError("device %d is not among the %d visible HIP devices" % (d, available))
    n = max(1, min(int(processes), len(ordered), int(replicates) // MIN_REPLICATES_PER_GPU))
    return ordered[:n]


# ------------------------------------------------------------------------------------------------ one process per GPU
class ProcessContext:
    def __init__(self, rank, world, local_rank, comm):
        self.rank, self.world, self.local_rank, self.comm = rank, world, local_rank, comm


_context = None
_rendezvous_seq = 0


def _rendezvous_dir(directory):
    """Directory of the id exchange: the caller's, ``PLSPM_RDZV_DIR``, or a per-user one under the temp dir created with mode 0700
    (the id is a bearer token of the job's RCCL bootstrap: other local users must neither read it nor plant one)."""
    directory = directory or os.environ.get("PLSPM_RDZV_DIR")
    if directory:
        # a caller-supplied directory: the user's own, or a sticky one (a shared /tmp-like place where nobody can replace another user's files);
        # the files themselves are created exclusively with mode 0600 and readers accept their own user's files only
        st = os.stat(directory)
        if st.st_uid != os.geteuid() and not (st.st_mode & 0o1000):
            from plspm import _native
            raise _native.NativeBackendError("rendezvous directory %s belongs to user %d and is not sticky: another user could replace the id file" % (directory, st.st_uid))
        return directory
    directory = os.path.join(tempfile.gettempdir(), "plspm-rdzv-%d" % os.geteuid())
    os.makedirs(directory, mode=0o700, exist_ok=True)
    st = os.stat(directory)
    if st.st_uid != os.geteuid() or (st.st_mode & 0o077):
        from plspm import _native
        raise _native.NativeBackendError("rendezvous directory %s is not private to this user (owner %d, mode %o): set PLSPM_RDZV_DIR"
                                         % (directory, st.st_uid, st.st_mode & 0o777))
    return directory


def _rendezvous_path(directory):
    global _rendezvous_seq
    _rendezvous_seq += 1
    # per-job nonce: launcher address / port, the launcher's pid, its run id and restart count (an elastic launcher that restarts its workers
    # keeps pid and port: the ranks of the new incarnation must not pick up the dead one's id), this process's call count
    tag = "%s-%s-%d-%s-%s-%d" % (os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0"), os.getppid(),
                                 os.environ.get("TORCHELASTIC_RUN_ID", "norun"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), _rendezvous_seq)
    tag = "".join(ch if (ch.isalnum() or ch in "-._") else "_" for ch in tag)
    return os.path.join(_rendezvous_dir(directory), "plspm-rdzv-" + tag)


def exchange_unique_id(rank, world, directory=None, timeout=300.0, make_id=None):
    """Rank 0 draws the ncclUniqueId and publishes it through a file that the other ranks of the same launcher (same parent
    process, same MASTER_ADDR/PORT) poll for: single-node rendezvous without a store service.  The file lives in a directory private
    to the user, is created exclusively with mode 0600, and a reader only accepts a file owned by its own user.  Every rank must call
    this the same number of times (the file name carries a per-process sequence number).  ``make_id`` replaces ncclGetUniqueId (tests)."""
    import struct
    from plspm import _native
    path = _rendezvous_path(directory)
    entered = time.time()
    deadline = entered + timeout
    if rank == 0:
        uid = (make_id or _native.rccl_unique_id)()
        tmp = path + ".tmp%d" % os.getpid()
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)       # never through a name somebody else prepared
        with os.fdopen(fd, "wb") as fh:
            fh.write(uid + struct.pack("<d", time.time()))                   # id + when it was published (readers reject a left-over of an earlier job)
        try:
            os.link(tmp, path)                     # atomic AND exclusive: fails when the name exists (no check-then-act window), a reader never sees a partial id
        except FileExistsError:
            raise _native.NativeBackendError("rendezvous: %s already exists (a stale or foreign file): remove it or set PLSPM_RDZV_DIR" % path) from None
        finally:
            os.remove(tmp)
        acks = [path + ".ack%d" % r for r in range(1, world)]
        try:
            while not all(os.path.exists(a) for a in acks):          # every rank has the id: nothing of this exchange stays behind
                if time.time() > deadline:
                    raise _native.NativeBackendError("rendezvous: not every rank picked up %s within %.0f s (ranks of one job must share "
                                                     "the launcher's parent process and MASTER_ADDR / MASTER_PORT; single node only)" % (path, timeout))
                time.sleep(0.002)
        finally:
            for name in [path] + acks:
                try:
                    os.remove(name)
                except FileNotFoundError:
                    pass
        return uid
    while True:
        try:
            with open(path, "rb") as fh:
                if os.fstat(fh.fileno()).st_uid != os.geteuid():
                    raise _native.NativeBackendError("rendezvous: %s belongs to another user" % path)
                blob = fh.read()
            if len(blob) == _native.UNIQUE_ID_BYTES + 8:
                published = struct.unpack("<d", blob[_native.UNIQUE_ID_BYTES:])[0]
                # a file published long before this rank came here is the left-over of a job that died (same launcher tag): its id leads
                # into a dead bootstrap -- ignore it, rank 0 of THIS job refuses the name and says so
                if published >= entered - timeout:
                    open(path + ".ack%d" % rank, "wb").close()
                    return blob[:_native.UNIQUE_ID_BYTES]
        except FileNotFoundError:
            pass
        if time.time() > deadline:
            raise _native.NativeBackendError("rendezvous: rank 0 did not publish %s within %.0f s (ranks of one job must share the "
                                             "launcher's parent process and MASTER_ADDR / MASTER_PORT; single node only)" % (path, timeout))
        time.sleep(0.002)


def init_process_group(rank=None, world_size=None, local_rank=None, rendezvous_dir=None, timeout=300.0):
    """Join the one-process-per-GPU job described by RANK / WORLD_SIZE / LOCAL_RANK (or the arguments): collective, every rank
    calls it once.  Afterwards ``Plspm(bootstrap=True)`` shards its replicates over the job.  Returns the context."""
    global _context
    from plspm import _native
    if _context is not None:
        return _context
    rank = int(os.environ.get("RANK", "0") if rank is None else rank)
    world = int(os.environ.get("WORLD_SIZE", "1") if world_size is None else world_size)
    local = int(os.environ.get("LOCAL_RANK", str(rank)) if local_rank is None else local_rank)
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > local_world:
        raise _native.NativeBackendError("WORLD_SIZE %d > LOCAL_WORLD_SIZE %d: the file rendezvous of plspm.parallel is single-node only "
                                         "(north_star: the 8 GPUs of ONE node); hand every rank the ncclUniqueId yourself and build "
                                         "NativeComm([local_rank], nranks, rank, unique_id) for a multi-node job" % (world, local_world))
    if local >= _native.device_count():
        raise _native.NativeBackendError("LOCAL_RANK %d but only %d HIP devices are visible" % (local, _native.device_count()))
    uid = exchange_unique_id(rank, world, rendezvous_dir, timeout)
    comm = _native.NativeComm([local], nranks=world, first_rank=rank, unique_id=uid)
    _context = ProcessContext(rank, world, local, comm)
    return _context


def context():
    """The process-group context, or None when this process runs on its own."""
    return _context


def destroy_process_group():
    global _context
    if _context is not None:
        _context.comm.close()
        _context = None


# ------------------------------------------------------------------------------------------------ host-side transport
def sharded_bootstrap(run_shard, total, width, comm=None):
    """Run ``total`` replicates split over a host-side communicator; every rank gets all of them back in replicate-id order.

    run_shard(count, first_id) -> (rows [count, width], status, iters) on the host (e.g. ``NativeModel.bootstrap``);
    comm: None (this process alone) or an object with ``rank``, ``world`` and ``all_gather(block) -> stacked blocks`` where every
    rank contributes a float64 array of identical shape.  Returns the merged [total, width + 2] records."""
    rank, world = (0, 1) if comm is None else (int(comm.rank), int(comm.world))
    start, stop = shard_range(total, rank, world)
    mine = stop - start
    cap = shard_range(total, 0, world)[1]              # rank 0 always holds the largest shard
    block = np.full((cap, width + 2), np.nan)          # NaN status = padding of a ragged split (the device summary skips it too)
    if mine > 0:
        block[:mine] = join_records(*run_shard(mine, start))
    if comm is None:
        return block[:mine]
    merged = np.asarray(comm.all_gather(block)).reshape(world, cap, width + 2)
    parts = []
    for r in range(world):
        a, b = shard_range(total, r, world)
        parts.append(merged[r, :b - a])
    return np.concatenate(parts, axis=0)
