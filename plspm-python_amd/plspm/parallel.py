"""Replicate sharding across GPUs: one process per GPU, no data-path collective, ONE gather at the end.

Replicates are independent (reference bootstrap.py:54-66 has no cross-replicate state; the reference shards
them over ``multiprocessing`` workers, bootstrap.py:91-94, and merges through a Queue, bootstrap.py:96-111).
Here rank g runs the contiguous replicate-id range ``shard_range(B, g, G)`` of one logical Philox stream keyed
by (seed, replicate id), so the merged result is identical for every G, and the Queue becomes a single
``all_gather`` (RCCL over xGMI when the process group's backend is "nccl", gloo in the CPU tests).
torch.distributed is used for rendezvous + the collective only.
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


def _world(group=None):
    try:
        import torch.distributed as dist
    except ImportError:
        return None, 0, 1
    if not (dist.is_available() and dist.is_initialized()):
        return None, 0, 1
    return dist, dist.get_rank(group), dist.get_world_size(group)


def sharded_bootstrap(run_shard, total, width, group=None, on_device=False):
    """Run ``total`` replicates split over the process group and return the merged
    (rows [total, width] float64, status [total] int32, iters [total] int32) on every rank, in replicate-id order.

    run_shard(count, first_id) executes one shard and returns
      * host arrays (rows, status, iters)                       when on_device is False, or
      * device pointers (rows_ptr, status_ptr, iters_ptr) + a ``sync`` callable  when on_device is True
        (buffers stay on the GPU and go straight into RCCL).
    """
    dist, rank, world = _world(group)
    start, stop = shard_range(total, rank, world)
    mine = stop - start
    res = run_shard(mine, start) if mine > 0 else None
    if dist is None:
        if on_device:
            import torch
            rows_ptr, st_ptr, it_ptr, sync = res
            sync()
            rows = torch.as_tensor(_DeviceView(rows_ptr, (mine, width), "<f8"), device="cuda").cpu().numpy()
            status = torch.as_tensor(_DeviceView(st_ptr, (mine,), "<i4"), device="cuda").cpu().numpy()
            iters = torch.as_tensor(_DeviceView(it_ptr, (mine,), "<i4"), device="cuda").cpu().numpy()
            return rows, status, iters
        return res
    import torch
    cap = shard_range(total, 0, world)[1]            # rank 0 always holds the largest shard
    dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
    # one packed record per replicate: [row | status | iters] as float64 so a single collective moves everything
    send = torch.zeros((cap, width + 2), dtype=torch.float64, device=dev)
    if mine > 0:
        if on_device:
            rows_ptr, st_ptr, it_ptr, sync = res
            sync()
            send[:mine, :width] = torch.as_tensor(_DeviceView(rows_ptr, (mine, width), "<f8"), device=dev)
            send[:mine, width] = torch.as_tensor(_DeviceView(st_ptr, (mine,), "<i4"), device=dev).to(torch.float64)
            send[:mine, width + 1] = torch.as_tensor(_DeviceView(it_ptr, (mine,), "<i4"), device=dev).to(torch.float64)
        else:
            rows, status, iters = res
            send[:mine, :width] = torch.from_numpy(np.ascontiguousarray(rows))
            send[:mine, width] = torch.from_numpy(status.astype(np.float64))
            send[:mine, width + 1] = torch.from_numpy(iters.astype(np.float64))
    recv = torch.empty((world * cap, width + 2), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)             # the single collective of the job
    merged = recv.cpu().numpy().reshape(world, cap, width + 2)
    parts = [merged[r, :shard_range(total, r, world)[1] - shard_range(total, r, world)[0]] for r in range(world)]
    flat = np.concatenate(parts, axis=0)
    return np.ascontiguousarray(flat[:, :width]), flat[:, width].astype(np.int32), flat[:, width + 1].astype(np.int32)
