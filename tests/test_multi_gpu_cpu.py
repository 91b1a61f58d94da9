"""The N>1 path on CPU: world_size-2 (and 3) gloo processes.  Each rank owns the image bands tiles.band_layout deals
it, fills them with the oracle's rows (standing in for its GPU's render: the GPU-side band render is checked against
the full frame in test_gpu_parity.py), and one gather to rank 0 plus the row un-interleave must rebuild the exact frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, height, width, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from importlib import import_module
        tiles = import_module("tiny-raytracer_amd.tiles")
        scenes = import_module("tiny-raytracer_amd.scenes")
        from oracle import orc
        desc = scenes.cornell(width, height)
        world, cam = orc.world_from_description(desc)
        lay = tiles.band_layout(height, world_size, rank)
        # this rank's rows only: the oracle keys its RNG by the image pixel, like the kernels, so the rows a rank
        # renders do not depend on who renders the others
        local = np.zeros((lay["rows_local"], width, 3), np.float32)
        scratch = np.zeros((height, width, 3), np.float32)
        rows = lay["rows"]
        start = 0
        while start < len(rows):                       # contiguous runs of rows = bands
            end = start
            while end + 1 < len(rows) and rows[end + 1] == rows[end] + 1:
                end += 1
            band, _ = orc.render(world, cam, 3, 6, desc["background"], seed=4, row_begin=rows[start], row_end=rows[end] + 1)
            scratch[rows[start]:rows[end] + 1] = band[rows[start]:rows[end] + 1]
            start = end + 1
        local[:] = scratch[rows]
        full = tiles.gather_image(torch.from_numpy(local), height, width, world_size, rank)
        if rank == 0:
            np.save(out_path, full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,height", [(2, 70), (3, 50), (2, 16)])
def test_band_gather_rebuilds_the_frame(tmp_path, world_size, height):
    width = 24
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world_size, _free_port(), height, width, out), nprocs=world_size, join=True)
    got = np.load(out)
    sys.path.insert(0, ROOT)
    from importlib import import_module
    scenes = import_module("tiny-raytracer_amd.scenes")
    from oracle import orc
    desc = scenes.cornell(width, height)
    world, cam = orc.world_from_description(desc)
    want, _ = orc.render(world, cam, 3, 6, desc["background"], seed=4, nthreads=4)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_single_rank_gather_is_identity():
    sys.path.insert(0, ROOT)
    from importlib import import_module
    tiles = import_module("tiny-raytracer_amd.tiles")
    x = torch.arange(5 * 4 * 3, dtype=torch.float32).reshape(5, 4, 3)
    assert tiles.gather_image(x, 5, 4, 1, 0) is x
