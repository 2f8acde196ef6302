"""Frames are independent: the multi-GPU path is a contiguous shard per rank and ONE collective at the
end, an all-gather of the ``[N/G, d_out]`` output shards (RCCL over xGMI when the backend is "nccl";
"gloo" on CPU tensors for tests).  One process per GPU (`torch.distributed`)."""

import torch
import torch.distributed as dist


def shard_bounds(n_frames, rank, world_size):
    """Contiguous block of frames owned by ``rank``: sizes differ by at most one frame."""
    base, extra = divmod(int(n_frames), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_frames, world_size):
    return [shard_bounds(n_frames, r, world_size)[1] - shard_bounds(n_frames, r, world_size)[0]
            for r in range(world_size)]


def all_gather_outputs(y_local, n_frames_total=None, group=None):
    """Every rank receives the outputs of all ranks, in frame order: ``[N, d_out]``.

    Equal shards use one ``all_gather_into_tensor`` (a single direct collective: each rank's shard goes
    over its own xGMI link to every peer); ragged shards are padded to the largest shard first.
    """
    world = dist.get_world_size(group)
    if world == 1:
        return y_local
    d_out = y_local.shape[1]
    if n_frames_total is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=y_local.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([y_local.shape[0]], dtype=torch.int64, device=y_local.device), group=group)
        sizes = [int(s.item()) for s in sizes]
    else:
        sizes = shard_sizes(n_frames_total, world)
    mx = max(sizes)
    if all(s == mx for s in sizes):
        out = torch.empty((world * mx, d_out), dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx, d_out), dtype=y_local.dtype, device=y_local.device)
    pad[:y_local.shape[0]] = y_local
    buf = torch.empty((world * mx, d_out), dtype=y_local.dtype, device=y_local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx:r * mx + sizes[r]] for r in range(world)], dim=0)


def forward_sharded(model, x_local, n_frames_total=None, group=None, gather=True):
    """Run ``model`` on this rank's frames and (optionally) all-gather the outputs."""
    with torch.no_grad():
        y = model(x_local)
    if gather and dist.is_initialized():
        return all_gather_outputs(y, n_frames_total, group)
    return y
