"""Multi-GPU work split and read-out for the progressive renderer.

Every (pixel, sample) is independent given the immutable scene (src/pathtracer.cpp:70-108),
so the path shards with no data-path collective: the scene is replicated and rank g renders
either sample indices g, g+G, g+2G, ... of every pixel (sample split) or every sample of the
image tiles k with k % G == g (tile split); a pixel a rank does not own has count 0 there.  The reference averages tone-mapped
samples (pathtracer.cpp:86-90), so the per-pixel value is linear in the samples and the
read-out is ONE exchange: sum_g n_g * mean_g and sum_g n_g, reduced to rank 0.  On GPUs that
reduce runs INSIDE the library, on the accumulators in HBM (mi355pt_comm_* + mi355pt_render_reduce,
or mi355pt_group_* for one process with several devices: RCCL over xGMI); `combine_running_means`
below is the same arithmetic on host tensors, for the gloo tests and as bench.py's fallback.
"""
import numpy as np


def rank_sample_split(rank, world):
    """(sample_offset, sample_stride) of a rank: samples rank, rank+world, ..."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return rank, world


def rank_tile_split(rank, world, tile_size=64):
    """(tile_size, tile_offset, tile_stride) of a rank for the image-tile partition: tile k of the row-major grid
    of tile_size x tile_size tiles goes to rank k % world -- interleaved, like the reference's strided workers
    (src/pathtracer.cpp:66-109), so every rank gets a sample of the whole image and the load balances."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return int(tile_size), rank, world


def tile_owner_mask(width, height, tile_size, tile_offset, tile_stride):
    """Boolean (height*width,) mask of the pixels a renderer with these tile parameters renders."""
    if tile_size <= 0 or tile_stride <= 1:
        return np.ones(width * height, dtype=bool)
    y, x = np.divmod(np.arange(width * height, dtype=np.int64), width)
    tx = (width + tile_size - 1) // tile_size
    return ((y // tile_size) * tx + x // tile_size) % tile_stride == tile_offset


def combine_running_means(mean, cnt, dist=None, dst=0):
    """mean: (P,3) float tensor, cnt: (P,) integer tensor of one rank.  Returns on `dst` the
    combined (mean, cnt) over all ranks, None elsewhere.  One reduce of 4 floats per pixel."""
    import torch
    n = cnt.to(torch.float32)
    packed = torch.cat([mean.to(torch.float32) * n[:, None], n[:, None]], dim=1).contiguous()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(packed, dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() != dst:
            return None
    total = packed[:, 3]
    out = packed[:, :3] / torch.clamp(total, min=1.0)[:, None]
    return out, total.to(torch.int64)


def resolve_rgb8(mean):
    """uint8(mean * 255) truncating, as the reference writes renderImage (pathtracer.cpp:101)."""
    m = np.asarray(mean, dtype=np.float32)
    return (m * np.float32(255.0)).astype(np.uint8)
