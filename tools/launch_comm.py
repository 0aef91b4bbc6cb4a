"""Launcher-side communicator on torch.distributed (gloo, host buffers) for malstroem_amd.distributed.BandPipeline.

The product package is torch-free: its data path is RCCL inside libmalstroem_hip.so and its control path needs an object
with four methods (exchange_rows / allreduce_max / allgather / clone).  A launcher that is started by
``python -m torch.distributed.run`` (bench.py, the gloo process tests) already has a rendezvous, so it wraps it here
instead of opening the package's own SocketComm.  Nothing in here touches a GPU.
"""
import atexit
import datetime
import os
import time

import numpy as np

# development aid: MALSTROEM_COMM_STATS=1 prints, per process at exit, calls and wall time of the three collectives (the time includes
# the wait for the slowest rank)
_STATS = {} if os.environ.get("MALSTROEM_COMM_STATS") else None      # method -> [calls, seconds] of this process (printed at exit)


def _timed(fn):
    if _STATS is None:
        return fn

    def wrapper(self, *a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(self, *a, **kw)
        finally:
            st = _STATS.setdefault(fn.__name__, [0, 0.0])
            st[0] += 1
            st[1] += time.perf_counter() - t0
    return wrapper


if _STATS is not None:
    @atexit.register
    def _report():
        import sys
        print("[TorchComm pid %d] %s" % (os.getpid(), ", ".join("%s: %d calls %.1f ms" % (k, v[0], v[1] * 1e3) for k, v in sorted(_STATS.items()))),
              file=sys.stderr, flush=True)


class TorchComm(object):
    """torch.distributed process group (gloo) as a BandPipeline control-plane ``Comm``."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    @_timed
    def exchange_rows(self, to_up, to_down):
        """host-staged neighbour exchange (bands without an RCCL communicator)"""
        torch, dist = self._torch, self._dist
        ops, recv_up, recv_down, keep = [], None, None, []
        if self.rank > 0:
            t = torch.from_numpy(np.ascontiguousarray(to_up).view(np.uint8).reshape(-1).copy())
            recv_up = torch.empty_like(t)
            keep.append(t)
            ops += [dist.P2POp(dist.isend, t, self.rank - 1, self._group), dist.P2POp(dist.irecv, recv_up, self.rank - 1, self._group)]
        if self.rank < self.size - 1:
            t = torch.from_numpy(np.ascontiguousarray(to_down).view(np.uint8).reshape(-1).copy())
            recv_down = torch.empty_like(t)
            keep.append(t)
            ops += [dist.P2POp(dist.isend, t, self.rank + 1, self._group), dist.P2POp(dist.irecv, recv_down, self.rank + 1, self._group)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        like_up = to_up if to_up is not None else to_down
        like_down = to_down if to_down is not None else to_up
        from_up = None if recv_up is None else recv_up.numpy().view(like_up.dtype).reshape(like_up.shape)
        from_down = None if recv_down is None else recv_down.numpy().view(like_down.dtype).reshape(like_down.shape)
        return from_up, from_down

    @_timed
    def allreduce_max(self, value):
        t = self._torch.tensor([float(value)], dtype=self._torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self._group)
        return float(t[0])

    @_timed
    def allgather(self, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj, group=self._group)
        return out

    def clone(self):
        """a second host group for the labelling branch's thread (collective call; short timeout: a rank that died must not
        leave the others waiting for gloo's default half hour)"""
        return TorchComm(self._dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=600)))
