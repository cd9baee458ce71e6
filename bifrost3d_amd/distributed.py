"""Multi-GPU framebuffer tiling: one process per GPU, tiles dealt round-robin, one gather at the end.

Every pixel-sample is independent and the RNG is a pure function of (pixel, accumulation, bounce), so the
path shards with no data-path collective; the only exchange is the gather of the finished half4 tiles to
rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests). The reference is single GPU
(OptiXRenderer/Renderer.cpp:289-291), so there is no reference behaviour beyond "same image".
"""
from __future__ import annotations

import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def tile_grid(width: int, height: int):
    return (width + 7) // 8, (height + 7) // 8


def owned_tile_count(width: int, height: int, rank: int, world: int) -> int:
    tx, ty = tile_grid(width, height)
    total = tx * ty
    return (total + world - 1 - rank) // world


def padded_pixels_per_rank(width: int, height: int, world: int) -> int:
    """Compact buffer length every rank uses for the gather (rank 0 owns the most tiles)."""
    return owned_tile_count(width, height, 0, world) * 64


def compact_pixel_coords(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """(n, 2) int array of the (x, y) each compact slot of `rank` holds; (-1, -1) for slots outside the frame."""
    tx, ty = tile_grid(width, height)
    n_tiles = owned_tile_count(width, height, rank, world)
    k = np.arange(n_tiles * 64)
    tile = (k // 64) * world + rank
    lane = k % 64
    x = (tile % tx) * 8 + lane % 8
    y = (tile // tx) * 8 + lane // 8
    valid = (tile < tx * ty) & (x < width) & (y < height)
    out = np.stack([x, y], axis=1)
    out[~valid] = -1
    return out


def assemble_numpy(compact: np.ndarray, width: int, height: int, world: int) -> np.ndarray:
    """Reference assembly of gathered compact buffers [world, n, C] into [height, width, C] (tests only; the
    product path is the k_scatter_tiles kernel behind hipr_scatter_tiles)."""
    out = np.zeros((height, width, compact.shape[-1]), compact.dtype)
    for rank in range(world):
        coords = compact_pixel_coords(width, height, rank, world)
        valid = coords[:, 0] >= 0
        out[coords[valid, 1], coords[valid, 0]] = compact[rank, : len(coords)][valid]
    return out


def gather_to_root(local, world: int, rank: int, group=None):
    """Gathers equally sized per-rank tensors on rank 0; returns [world, ...] there, None elsewhere. `group`: another process group than the default one."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local.unsqueeze(0)
    if rank == 0:
        gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=list(gathered.unbind(0)), dst=0, group=group)      # the ranks' tiles land in place: no stacking copy afterwards
        return gathered
    dist.gather(local, gather_list=None, dst=0, group=group)
    return None
