"""Test infrastructure: a gloo-backed host communicator with the interface plspm.parallel.sharded_bootstrap expects
(rank / world / all_gather).  torch is used by the TESTS only; the product never imports it."""
import numpy as np


class GlooComm:
    def __init__(self, rank, world, port):
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        self._dist = dist
        self.rank, self.world = rank, world

    def all_gather(self, block):
        import torch
        send = torch.from_numpy(np.ascontiguousarray(block, dtype=np.float64))
        recv = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=torch.float64)
        self._dist.all_gather_into_tensor(recv, send)
        return recv.numpy().reshape((self.world,) + tuple(send.shape))

    def close(self):
        self._dist.destroy_process_group()


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p
