"""Frame sharding across the GPUs of one node and the gather of the fixed-size result
records: frames are independent through FrameScanResult, so rank g of G
takes the contiguous range [g*N/G, (g+1)*N/G) and the only exchange is the gather of the
fixed-size records of a batch on rank 0 (point-to-point sends over xGMI with RCCL on GPUs; gloo
in the CPU tests).  `RootGatherer` issues the gather asynchronously so that it overlaps the next
batch's kernels; `gather_results` is the blocking all-gather for callers that want the records
on every rank."""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous frame range [lo, hi) of `rank`; ranges tile [0, total) in rank order."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    return (total * rank) // world, (total * (rank + 1)) // world


def gather_results(local, world=None, group=None, out=None):
    """All-gather per-rank result records (uint8 tensor [n_local, 1024], same n_local on
    every rank) into [world * n_local, 1024] ordered by rank, i.e. by frame index."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, local, group=group)
    else:  # gloo
        chunks = list(out.chunk(world, dim=0))
        dist.all_gather(chunks, local, group=group)
    return out


class RootGatherer:
    """Gathers every rank's record tensor of a batch on rank `dst`, asynchronously.

    xGMI is point-to-point, so a gather is G-1 independent sends into the root's links -- 1/G of
    the traffic of an all-gather, none of it on the critical path: `submit()` enqueues the
    exchange behind the kernels that produced `local` and returns; the caller must not overwrite
    `local` (or read the returned buffer on the root) before `wait()`.  With two alternating
    record buffers the exchange of batch k runs while batch k+1 is being scanned.
    """

    def __init__(self, world=None, dst=0, group=None):
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = dist.get_rank(group) if (dist.is_initialized() and self.world > 1) else 0
        self.dst = dst
        self._pending = {}  # slot -> outstanding work handles
        self._bufs = {}
        self._all_gather = False

    def _root_buffer(self, local, slot):
        key = (slot, tuple(local.shape), local.dtype, str(local.device))
        buf = self._bufs.get(key)
        if buf is None:
            buf = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
            self._bufs[key] = buf
        return buf

    def submit(self, local, slot=0):
        """Start gathering `local` ([n_local, ...], same shape on every rank); returns the root's
        [world * n_local, ...] buffer (ordered by rank, i.e. by frame index) or None elsewhere."""
        if self.world == 1:
            return local
        if not self._all_gather:
            out = None
            gather_list = None
            if self.rank == self.dst:
                out = self._root_buffer(local, slot)
                gather_list = list(out.chunk(self.world, dim=0))
            try:
                work = dist.gather(local, gather_list, dst=self.dst, group=self.group, async_op=True)
                self._pending.setdefault(slot, []).append(work)
                return out
            except NotImplementedError:
                # a backend without gather: every rank takes the same exit on its first call, so the
                # ranks stay in step; from here on use the blocking all-gather
                self._all_gather = True
        out = gather_results(local, self.world, self.group, out=self._root_buffer(local, slot))
        return out if self.rank == self.dst else None

    def wait(self, slots=None):
        """Block the current stream (GPU) / the caller (CPU) until the gathers submitted on `slots` (default:
        all of them) are done -- e.g. the slots whose `local` buffers are about to be overwritten."""
        for slot in (list(self._pending) if slots is None else slots):
            for w in self._pending.pop(slot, []):
                w.wait()
