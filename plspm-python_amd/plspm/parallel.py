"""Replicate sharding across GPUs: one process per GPU, no data-path collective, ONE gather at the end.

Replicates are independent (reference bootstrap.py:54-66 has no cross-replicate state; the reference shards
them over ``multiprocessing`` workers, bootstrap.py:91-94, and merges through a Queue, bootstrap.py:96-111).
Here rank g runs the contiguous replicate-id range ``shard_range(B, g, G)`` of one logical Philox stream keyed
by (seed, replicate id), so the merged result is identical for every G, and the Queue becomes a single
``all_gather`` (RCCL over xGMI when the process group's backend is "nccl", gloo in the CPU tests).
torch.distributed is used for rendezvous + the collective only; the device buffer that libplspm_hip filled is
handed to RCCL in place (CUDA-array-interface view, no staging copy when the shards are equal).
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous, balanced [start, stop) of replicate ids for ``rank``."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class _DeviceView:
    """Zero-copy view of a device buffer owned by libplspm_hip for torch (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


_view_cache = {}
_recv_cache = {}


def device_rows(ptr, count, stride):
    """torch view [count, stride] float64 of the result rows libplspm_hip left in HBM (cached per buffer)."""
    import torch
    key = (int(ptr), int(count), int(stride), torch.cuda.current_device())
    t = _view_cache.get(key)
    if t is None:
        if len(_view_cache) > 64:
            _view_cache.clear()
        t = torch.as_tensor(_DeviceView(ptr, (count, stride), "<f8"), device=torch.device("cuda", torch.cuda.current_device()))
        _view_cache[key] = t
    return t


def _world(group=None):
    try:
        import torch.distributed as dist
    except ImportError:
        return None, 0, 1
    if not (dist.is_available() and dist.is_initialized()):
        return None, 0, 1
    return dist, dist.get_rank(group), dist.get_world_size(group)


def split_records(records, width):
    """[row | status | iterations] records -> (rows, status int32, iters int32)."""
    records = np.asarray(records)
    return np.ascontiguousarray(records[:, :width]), records[:, width].astype(np.int32), records[:, width + 1].astype(np.int32)


def gather_records(send, total, group=None, async_op=False, slot=0):
    """all_gather of per-rank record blocks (a torch tensor [mine, width+2] on the group's device) into one
    [total, width+2] tensor in replicate-id order, on every rank.  This is the single collective of a bootstrap.
    async_op=True (equal shards only) returns (recv, work): the collective runs on RCCL's own stream and the caller
    overlaps it with other work, calling work.wait() before touching ``recv`` or re-using ``send``; ``slot`` selects
    one of several receive buffers so that consecutive jobs do not share one."""
    import torch
    dist, rank, world = _world(group)
    if dist is None:
        return send
    if dist.get_backend(group) == "nccl" and not send.is_cuda:
        send = send.cuda()                               # RCCL moves device buffers only
    stride = send.shape[1]
    cap = shard_range(total, 0, world)[1]                 # rank 0 always holds the largest shard
    equal = (total % world == 0)
    if not equal:
        padded = torch.zeros((cap, stride), dtype=send.dtype, device=send.device)
        padded[:send.shape[0]] = send
        send = padded
    key = (world, cap, stride, str(send.device), slot)
    recv = _recv_cache.get(key)
    if recv is None:
        recv = torch.empty((world * cap, stride), dtype=send.dtype, device=send.device)
        _recv_cache[key] = recv
    if async_op:
        if not equal:
            raise ValueError("async gather needs equal shards")
        work = dist.all_gather_into_tensor(recv, send.contiguous(), group=group, async_op=True)
        return recv, work
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if equal:
        return recv
    parts = []
    for r in range(world):
        a, b = shard_range(total, r, world)
        parts.append(recv[r * cap:r * cap + (b - a)])
    return torch.cat(parts, dim=0)


def sharded_bootstrap(run_shard, total, width, group=None, on_device=False, to_host=True):
    """Run ``total`` replicates split over the process group; every rank gets all of them back, in replicate-id order.

    run_shard(count, first_id) executes one shard and returns
      * host arrays (rows [count, width], status, iters)                        when on_device is False, or
      * (device pointer of [count, width+2] records, sync callable)             when on_device is True
        (the buffer stays in HBM and goes straight into RCCL).
    Returns (rows, status, iters) as NumPy arrays, or the merged record tensor when to_host is False.
    """
    import torch
    dist, rank, world = _world(group)
    start, stop = shard_range(total, rank, world)
    mine = stop - start
    if on_device:
        if mine > 0:
            ptr, sync = run_shard(mine, start)
            sync()
            send = device_rows(ptr, mine, width + 2)
        else:
            send = torch.zeros((0, width + 2), dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
    else:
        if mine > 0:
            rows, status, iters = run_shard(mine, start)
            rec = np.concatenate((rows, status[:, None].astype(np.float64), iters[:, None].astype(np.float64)), axis=1)
        else:
            rec = np.zeros((0, width + 2))
        if dist is None:
            return split_records(rec, width) if to_host else rec
        send = torch.from_numpy(np.ascontiguousarray(rec))
    merged = gather_records(send, total, group)
    if not to_host:
        return merged
    return split_records(merged.cpu().numpy(), width)
