"""Multi-GPU sharding of the hot path: image tiles across ranks, one final framebuffer exchange.

Replaces the reference's image-space work splitting (spiral 32x32 blocks handed to workers, src/librender/imageproc.cpp:28-79)
and its merge step Film::put(block) under a mutex (src/librender/renderproc.cpp:142-149) by: static tiles per rank (contiguous row bands, or rows interleaved over the ranks for load balance)
(every rank holds a full scene replica), sampler state depending only on GLOBAL (px, py, sampleIndex), and ONE sum-reduce of
the raw film (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" in the CPU tests).  A sum is used rather than a
gather because reconstruction filters wider than a pixel splat across tile borders.
"""
import numpy as np


def shard_rows(height, rank, world):
    """Contiguous row band [y0, y1) of rank `rank`; bands differ by at most one row."""
    base, rem = divmod(height, world)
    y0 = rank * base + min(rank, rem)
    return y0, y0 + base + (1 if rank < rem else 0)


def tile_of(width, height, rank, world):
    y0, y1 = shard_rows(height, rank, world)
    return (0, y0, width, y1)


def interleaved_rows(width, height, rank, world):
    """Row-interleaved ownership: rank k renders film rows k, k + world, ... -> (tile, row_stride) for mi_render_run_rows.  Every rank gets a
    statistically identical slice of the image (walls, floor, light, blocks), which a contiguous band does not."""
    return (0, rank, width, height), world


def reduce_film(film, dist=None, dst=0):
    """Sum the per-rank raw films (un-normalised ImageBlock sums) onto rank `dst`. `film` is a torch tensor (device or CPU)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def render_sharded(render_tile, width, height, rank, world, s0, s1):
    """Render this rank's tile; `render_tile(tile, s0, s1)` is the renderer (the HIP path in production)."""
    tile = tile_of(width, height, rank, world)
    if tile[3] > tile[1]:
        render_tile(tile, s0, s1)
    return tile
