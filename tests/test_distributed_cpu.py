"""world_size-2 and -4 gloo tests of the N > 1 path: tile partition -> per-rank compact buffers -> gather on rank 0 ->
assembly equals the single-process image. The per-rank pixels come from the CPU oracle (no GPU here); on the GPU
box the same gather feeds hipr_scatter_tiles (covered by tests/test_gpu_parity.py::test_scatter_tiles_roundtrip)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, width, height, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist
    from bifrost3d_amd import distributed
    from bifrost3d_amd.host import Scene
    from oracle_bindings import get_oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = Scene("cornell", diffuse_only=True)
    oracle = get_oracle(True)
    cam = scene.camera(width, height, max_bounce_count=2)
    full, _, _ = oracle.render(scene.desc, scene.state, cam, width, height, 1)   # deterministic: every rank gets the same image

    coords = distributed.compact_pixel_coords(width, height, rank, world)
    n = distributed.padded_pixels_per_rank(width, height, world)
    compact = np.zeros((n, 4), np.float64)
    valid = coords[:, 0] >= 0
    compact[: len(coords)][valid] = full[coords[valid, 1], coords[valid, 0]]
    gathered = distributed.gather_to_root(torch.from_numpy(compact), world, rank)
    if rank == 0:
        assembled = distributed.assemble_numpy(gathered.numpy(), width, height, world)
        np.save(out_path, np.stack([assembled, full]))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(40, 24), (37, 19)])
def test_two_rank_tile_gather_reassembles_the_frame(tmp_path, size):
    import torch.multiprocessing as mp
    out = tmp_path / "frame.npy"
    mp.spawn(_worker, args=(2, _free_port(), size[0], size[1], str(out)), nprocs=2, join=True)
    assembled, full = np.load(out)
    assert np.array_equal(assembled, full)
    assert full[..., :3].max() > 0


@pytest.mark.parametrize("size", [(37, 19), (72, 40)])
def test_four_rank_tile_gather_with_uneven_shares(tmp_path, size):
    """World size 4: 37 x 19 is 5 x 3 = 15 tiles -- an ODD count over four ranks, shares of 4 / 4 / 4 / 3 tiles with partial tiles on two
    edges, so the last rank's compact buffer is padded to rank 0's length and its padding must not land in the frame; 72 x 40 is 45 tiles (12 / 11 / 11 / 11)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT))
    from bifrost3d_amd import distributed
    tiles = [distributed.owned_tile_count(size[0], size[1], r, 4) for r in range(4)]
    assert sum(tiles) == ((size[0] + 7) // 8) * ((size[1] + 7) // 8) and tiles[0] > tiles[3]
    covered = np.zeros((size[1], size[0]), int)
    for r in range(4):
        coords = distributed.compact_pixel_coords(size[0], size[1], r, 4)
        valid = coords[:, 0] >= 0
        np.add.at(covered, (coords[valid, 1], coords[valid, 0]), 1)
    assert (covered == 1).all()      # every pixel owned by exactly one rank
    out = tmp_path / "frame.npy"
    mp.spawn(_worker, args=(4, _free_port(), size[0], size[1], str(out)), nprocs=4, join=True)
    assembled, full = np.load(out)
    assert np.array_equal(assembled, full)
    assert full[..., :3].max() > 0
