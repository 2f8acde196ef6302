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


def chunk_bounds(n, c, chunks):
    """Piece ``c`` of ``chunks`` near-equal contiguous pieces of ``n`` rows (same arithmetic as `shard_bounds`)."""
    return shard_bounds(n, c, chunks)


class OutputGather(object):
    """The path's one collective, for equal shards of ``frames`` rows per rank, in two forms with the same result
    (``[world * frames, d_out]`` in frame order on every rank):

    ``collective(y)``             one ``all_gather_into_tensor`` behind the step that produced ``y``;
    ``forward_overlapped(m, x)``  the step itself cut into ``chunks`` pieces: piece i is computed, copied into this
                                  rank's rows of the result and posted to every peer (one send + one receive per peer,
                                  batched, each received piece landing in its final rows - over xGMI that is one
                                  direct transfer per link, no ring, no staging copy) while piece i+1 computes.
                                  torch.distributed runs the transfers on the communicator's own stream behind an event
                                  on the launch stream, which is what lets them run beside the next piece's kernel.

    The result buffer is allocated once and reused by every call (launches do not allocate).
    """

    def __init__(self, frames, d_out, device, world=None, rank=None, chunks=4, dtype=torch.float32, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        self.frames, self.d_out = int(frames), int(d_out)
        self.chunks = max(1, min(int(chunks), max(1, self.frames)))
        self.out = torch.empty((self.world, self.frames, self.d_out), dtype=dtype, device=device)

    def result(self):
        return self.out.view(self.world * self.frames, self.d_out)

    def collective(self, y_local):
        assert tuple(y_local.shape) == (self.frames, self.d_out)
        dist.all_gather_into_tensor(self.result(), y_local.contiguous(), group=self.group)
        return self.result()

    def forward_overlapped(self, model, x_local):
        assert x_local.shape[0] == self.frames
        pending = []
        for c in range(self.chunks):
            a, b = chunk_bounds(self.frames, c, self.chunks)
            if b == a:
                continue
            y = model(x_local[a:b])
            mine = self.out[self.rank, a:b]
            mine.copy_(y)
            ops = []
            for step in range(1, self.world):
                to, frm = (self.rank + step) % self.world, (self.rank - step) % self.world
                ops.append(dist.P2POp(dist.isend, mine, to, group=self.group))
                ops.append(dist.P2POp(dist.irecv, self.out[frm, a:b], frm, group=self.group))
            if ops:
                pending.extend(dist.batch_isend_irecv(ops))
        for work in pending:
            work.wait()
        return self.result()


def forward_sharded(model, x_local, n_frames_total=None, group=None, gather=True):
    """Run ``model`` on this rank's frames and (optionally) all-gather the outputs."""
    with torch.no_grad():
        y = model(x_local)
    if gather and dist.is_initialized():
        return all_gather_outputs(y, n_frames_total, group)
    return y
