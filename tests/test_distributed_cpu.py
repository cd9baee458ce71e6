"""world_size-2 gloo test of the N > 1 path: tile partition -> per-rank compact buffers -> gather on rank 0 ->
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
