"""Frame sharding across the GPUs of one node and the gather of the fixed-size result
records: frames are independent through FrameScanResult, so rank g of G
takes the contiguous range [g*N/G, (g+1)*N/G) and the only exchange is one all-gather of
1 KiB records per batch (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
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
