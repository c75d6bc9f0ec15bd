"""Multi-GPU partition of the frame (actinon_amd/dist.py), exercised with world_size 2 over gloo on the CPU.

The renderer plugged in here is the CPU oracle (this is tests/): what is under test is the partition, the
zero-initialised per-pixel accumulators and the single sum all-reduce -- the same code path bench.py runs with RCCL.
The reduced frame must be bit-identical to a single-process render because the ranks' supports are disjoint."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import actinon_amd as A
from actinon_amd import dist as adist
import scenes_util as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, flat = S.build(name)
    w, h = int(flat.params.image_width), int(flat.params.image_height)
    oracle = Oracle()

    def render(pos):
        return oracle.render_positions(flat, pos, linear=True, threads=2)

    def all_reduce_sum(frame):
        t = torch.from_numpy(frame)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    frame = adist.render_frame_sharded(render, w, h, rank, world, all_reduce_sum, tile=64)
    if rank == 0:
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_process(tmp_path, oracle, world):
    name = "wine_glass_c2"
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), name, out), nprocs=world, join=True)
    frame = np.load(out)
    sc, flat = S.build(name)
    ref = oracle.render_positions(flat, S.positions(flat), linear=True)
    assert np.array_equal(frame, ref)


def test_partition_is_a_partition():
    n = 1920 * 1080
    for world in (1, 2, 4, 8):
        seen = np.zeros(n, dtype=np.int32)
        sizes = []
        for r in range(world):
            idx = adist.rank_pixels(n, r, world)
            seen[idx] += 1
            sizes.append(len(idx))
        assert (seen == 1).all()
        assert max(sizes) - min(sizes) <= adist.TILE
    # ragged: fewer tiles than ranks leaves some ranks empty, nothing is lost
    idx = [adist.rank_pixels(100, r, 8, tile=64) for r in range(8)]
    assert sum(len(i) for i in idx) == 100 and len(idx[5]) == 0
    pos = adist.pixel_positions(np.array([0, 1919, 1920]), 1920)
    assert pos.tolist() == [[0.5, 0.5], [1919.5, 0.5], [0.5, 1.5]]
