"""Multi-GPU sharding of reference views.

Reference views are independent units (reference: app/smvsrecon.cc:658-733),
so the path shards with NO data-path collective: one process per GPU, each
rank optimises its own views.  Collectives appear only
  * in the measurement (barrier, max-over-ranks time, summed work units), and
  * in the opt-in "shared lighting" mode, where the 16x16 + 16 lighting normal
    equations (light_optimizer.cc:32-49) of the views processed in lock step
    are summed over ranks before the host pseudo inverse (2,176 bytes: pure
    latency, ring-vs-tree and xGMI link bandwidth are irrelevant).
torch.distributed is plumbing here: "nccl" is RCCL on ROCm, "gloo" on CPU.
"""
import contextlib
import os
import sys

import numpy as np


def flush_c_stdio():
    """fflush(NULL): what C libraries in this process (RCCL prints a version
    banner on stdout when a communicator is created) still hold in their stdio
    buffers."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


@contextlib.contextmanager
def quiet_stdout():
    """File descriptor 1 points to /dev/null inside the block (C stdio is
    flushed on both sides): RCCL's banner does not end up beside the ONE JSON
    line bench.py owes its caller."""
    sys.stdout.flush()
    flush_c_stdio()
    saved = os.dup(1)
    null = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(null, 1)
        yield
    finally:
        sys.stdout.flush()
        flush_c_stdio()
        os.dup2(saved, 1)
        os.close(saved)
        os.close(null)


def assign_views(num_views, world_size, rank):
    """Round-robin view -> rank map: rank r gets views r, r + W, r + 2W, ...
    Neighbouring view ids usually share neighbour images, round-robin keeps
    the per-rank work balanced when views differ in cost."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return list(range(rank, num_views, world_size))


def lockstep_batches(num_views, world_size):
    """Batches of view ids processed in lock step (one per rank) so that
    collective calls of the shared-lighting mode match across ranks; the last
    batch may leave ranks idle (None)."""
    batches = []
    for start in range(0, num_views, world_size):
        batches.append([start + r if start + r < num_views else None
                        for r in range(world_size)])
    return batches


def allreduce_lighting(A, b, dist=None, device=None):
    """Sum the lighting normal equations over ranks.  A (16,16), b (16) numpy;
    returns the summed (A, b).  With dist None (single process) it is the
    identity.  Ranks without a view in the batch pass zeros."""
    buf = np.concatenate([np.asarray(A, dtype=np.float64).reshape(256),
                          np.asarray(b, dtype=np.float64).reshape(16)])
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        t = torch.from_numpy(buf.copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        buf = t.cpu().numpy()
    return buf[:256].reshape(16, 16), buf[256:]


class _DevicePointer:
    """Zero-copy view of `count` doubles at a raw device address for
    torch.as_tensor (the __cuda_array_interface__ protocol)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, count, device):
    """torch float64 tensor aliasing libsmvs_hip's device buffer (no copy)."""
    import torch
    return torch.as_tensor(_DevicePointer(ptr, count), device=device)


def allreduce_lighting_device(ptrs, dist=None, device=None):
    """Shared-lighting collective directly on the buffers
    smvs_light_accumulate_dev left on the device (272 doubles each: A then b).
    `ptrs`: the device addresses of the views this rank processes in the
    current lock-step round.  They are summed on the device, all-reduced over
    the ranks in place with RCCL (no host hop) and the result is written back
    to every buffer, so that each view's smvs_light_download returns the sum
    over all views of the round."""
    import torch
    ts = [device_tensor(p, 272, device) for p in ptrs]
    if not ts:
        return
    total = ts[0]
    for t in ts[1:]:
        total += t
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    for t in ts[1:]:
        t.copy_(total)
    torch.cuda.synchronize(device)


class NativeComm:
    """RCCL communicator of this rank behind include/smvs_rccl.h
    (csrc/libsmvs_rccl.so): the lighting all-reduce without PyTorch in the data
    path.  `dist` (torch.distributed, any backend) is used ONCE, to hand rank
    0's 128-byte unique id to the other ranks."""

    def __init__(self, device_index, dist=None):
        import ctypes as C
        import os
        from . import _capi
        _capi.load()
        path = os.path.join(_capi.HERE, "csrc", "libsmvs_rccl.so")
        if not os.path.exists(path):
            raise _capi.SmvsError(-2, "RCCL library %s is missing" % path)
        self.lib = C.CDLL(path)
        rank, world = 0, 1
        if dist is not None and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
        ident = (C.c_char * 128)()
        if rank == 0:
            _capi.check(self.lib.smvs_comm_unique_id(ident))
        if world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_char * 128).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        with quiet_stdout():
            _capi.check(self.lib.smvs_comm_create(device_index, rank, world, ident,
                                                  C.byref(self.handle)))
        self.rank, self.world = rank, world

    def allreduce_lighting(self, ctxs):
        """Sum the device-resident normal equations (smvs_light_accumulate_dev)
        of the ViewContexts `ctxs` and over the ranks, in place."""
        import ctypes as C
        from . import _capi
        arr = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
        _capi.check(self.lib.smvs_light_allreduce(self.handle, arr, len(ctxs)))

    def ranks(self):
        """(ranks in the communicator, this rank) as RCCL reports them
        (ncclCommCount, ncclCommUserRank)."""
        import ctypes as C
        from . import _capi
        n, r = C.c_int(0), C.c_int(-1)
        _capi.check(self.lib.smvs_comm_ranks(self.handle, C.byref(n), C.byref(r)))
        return n.value, r.value

    def close(self):
        if self.handle:
            self.lib.smvs_comm_destroy(self.handle)
            import ctypes as C
            self.handle = C.c_void_p()


def solve_lighting(A, b):
    """params = pinv(A) b (light_optimizer.cc:50-52), singular values below
    1e-12 of the largest dropped."""
    return np.linalg.pinv(np.asarray(A, dtype=np.float64), rcond=1e-12,
                          hermitian=True) @ np.asarray(b, dtype=np.float64)


def aggregate_throughput(units, seconds, dist=None, device=None):
    """Whole-job numbers of one timed region: work units summed over ranks,
    time = max over ranks.  Returns (total_units, max_seconds)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(units), float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64)
    u = torch.tensor([float(units)], dtype=torch.float64)
    if device is not None:
        t = t.to(device); u = u.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())
