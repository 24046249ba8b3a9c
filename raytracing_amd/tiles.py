"""Image-row partitioner and frame assembly for one-process-per-GPU rendering.

The reference splits the image into ``threads`` contiguous row stripes, one Python thread each, and
stacks them afterwards (T3DR.py:1111-1130, 1142-1165).  Here a "stripe" belongs to one rank of a
``torch.distributed`` job (one process per GPU, RCCL over xGMI when the backend is ``nccl``).
Pixels are independent and the random numbers are keyed by absolute pixel coordinates, so the
picture does not depend on the partition.  The scene is replicated on every GPU; the only data-path
exchange is one gather of uint8 BGR tiles to rank 0 plus a tiny sum of counters.

Two partitions:
* ``"bands"``       rank r renders rows [r*H/G, (r+1)*H/G)   -- the reference's stripes
* ``"interleaved"`` rank r renders rows r, r+G, r+2G, ...    -- default: the Cornell frame is cheap
  at the top and bottom (rays that miss the box), interleaving evens out the load
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class RowTile:
    row0: int     # first image row
    nrows: int    # rows rendered by this rank
    stride: int   # distance between consecutive rendered rows

    def rows(self):
        return range(self.row0, self.row0 + self.nrows * self.stride, self.stride)


def partition_rows(height: int, world: int, rank: int, mode: str = "interleaved") -> RowTile:
    """Rows of ``rank``.  ``height`` must be a multiple of ``world`` (the reference's rule for its
    thread count, T3DR.py:1111-1112), so every rank's tile has the same shape."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} for world size {world}")
    if height % world != 0:
        raise ValueError("Number of GPUs must be a factor of image height")
    n = height // world
    if mode == "bands":
        return RowTile(rank * n, n, 1)
    if mode == "interleaved":
        return RowTile(rank, n, world)
    raise ValueError(f"unknown partition mode {mode!r}")


def assemble(tiles, height: int, mode: str = "interleaved"):
    """Stack per-rank tiles ``[H/G, W, C]`` (a list ordered by rank, tensors or arrays) into the frame."""
    world = len(tiles)
    first = tiles[0]
    is_torch = hasattr(first, "new_empty")
    if is_torch:
        frame = first.new_empty((height,) + tuple(first.shape[1:]))
    else:
        import numpy as np
        frame = np.empty((height,) + tuple(first.shape[1:]), dtype=first.dtype)
    for rank, tile in enumerate(tiles):
        t = partition_rows(height, world, rank, mode)
        frame[t.row0:t.row0 + t.nrows * t.stride:t.stride] = tile
    return frame


_workspace = {}   # receive buffer and frame per gather shape and stream (rank dst only)


def _gather_workspace(tile, world: int, height: int, mode: str):
    """Receive buffer [G, (B,) H/G, W, C] with its per-rank views, the frame(s) [(B,) H, W, C], and the two sides of
    the one strided copy that puts the received rows in place: made once per shape, used every frame."""
    stacked = tile.new_empty((world,) + tuple(tile.shape))
    lead, rest = tuple(tile.shape[:-3]), tuple(tile.shape[-2:])     # (B,) or (), and (W, C)
    frame = tile.new_empty(lead + (height,) + rest)
    n = height // world
    b = len(lead)                                                    # 0 or 1: where the row axes start in `frame`
    ranks_last = list(range(1, 1 + b)) + [0] + list(range(1 + b, stacked.ndim))   # [G, (B,) n, ...] -> [(B,) G, n, ...]
    if mode == "bands":
        dst_view, src_view = frame.view(lead + (world, n) + rest), stacked.permute(ranks_last)
    elif mode == "interleaved":
        dst_view, src_view = frame.view(lead + (n, world) + rest), stacked.permute(ranks_last).transpose(b, b + 1)
    else:
        raise ValueError(f"unknown partition mode {mode!r}")
    return [stacked[r] for r in range(world)], frame, dst_view, src_view


def gather_frame(tile, height: int, mode: str = "interleaved", group=None, dst: int = 0, slot: int = 0, collective_always: bool = False):
    """Collect every rank's tile on rank ``dst`` and return the assembled frame there (``None`` on
    the other ranks).  ``tile`` is a torch tensor [H/G, W, C] on the device the backend moves
    (GPU for nccl/RCCL, CPU for gloo) -- or [B, H/G, W, C], the tiles of B frames rendered by one call: they
    travel in one collective and come back as frames [B, H, W, C].  Single-process use returns the tile's own
    frame.  The returned frame is a buffer that the next call with the same shapes, stream and ``slot``
    overwrites: a caller that keeps several frames in flight gives each chain a slot (and, on a GPU, a stream)
    of its own.  ``collective_always``: a group of one rank goes through the collective and the strided copy as well
    (the whole several-rank code path, exercised on a one-GPU box) instead of being answered from the tile."""
    import torch.distributed as dist

    batched = tile.ndim == 4
    grouped = dist.is_available() and dist.is_initialized()
    if not grouped or (dist.get_world_size(group) == 1 and not collective_always):
        if tile.shape[-3] == height:
            return tile                                                      # one rank: the tile is the frame
        if batched:
            import torch
            return torch.stack([assemble([t], height, mode) for t in tile])
        return assemble([tile], height, mode)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    tile = tile.contiguous()
    if rank == dst:
        # one receive buffer [G, (B,) H/G, W, C] and one frame, kept between calls; the tiles land in the buffer
        # and a single strided copy puts the rows in place
        # per stream: two frames in flight on two streams must not share a receive buffer
        stream_id = 0
        if tile.is_cuda:
            import torch
            stream_id = torch.cuda.current_stream(tile.device).cuda_stream
        key = (tuple(tile.shape), tile.dtype, tile.device, world, height, stream_id, slot, mode)
        ws = _workspace.get(key)
        if ws is None:
            if len(_workspace) > 16:
                _workspace.clear()
            ws = _gather_workspace(tile, world, height, mode)
            _workspace[key] = ws
        parts, frame, dst_view, src_view = ws
        dist.gather(tile, parts, dst=dst, group=group)
        dst_view.copy_(src_view)
        return frame
    dist.gather(tile, None, dst=dst, group=group)
    return None


def sum_counters(values, device=None, group=None):
    """All-reduce (sum) a short list of integer counters (rays, queries, tests) across ranks."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(v) for v in t.tolist()]
