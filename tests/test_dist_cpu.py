"""Multi-GPU partition of a frame (include/actinon_hip.h: acn_shard_tile_*, ACN_SHARD_SAMPLES; actinon_amd/dist.py),
exercised with world_size 2 and 3 over gloo on the CPU.

The renderer plugged in here is the CPU oracle (this is tests/): what is under test is the GPU-free partition logic of
both splits -- the library's tile arithmetic with its all-gather, and the sample sub-ranges with their sum-reduce and
rank-0-only terms -- the same code paths bench.py runs over RCCL."""
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


def _worker(rank, world, port, name, split, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, flat = S.build(name)
    w, h = int(flat.params.image_width), int(flat.params.image_height)
    oracle = Oracle()

    if split == "tiles":
        def render(pos):
            return oracle.render_positions(flat, pos, linear=True, threads=2)

        def all_gather(part):
            out = torch.empty((world * part.shape[0], 3), dtype=torch.float64)     # rank-major concatenation
            dist.all_gather_into_tensor(out, torch.from_numpy(part))
            return out.numpy().reshape(world, part.shape[0], 3)

        frame = adist.render_frame_tiles(render, w, h, rank, world, all_gather)
    else:
        def render_shard(pos, r, n):
            return oracle.render_positions(flat, pos, linear=True, threads=2, shard=(r, n))

        def all_reduce_sum(frame):
            dist.all_reduce(torch.from_numpy(frame), op=dist.ReduceOp.SUM)

        frame = adist.render_frame_samples(render_shard, w, h, rank, world, all_reduce_sum)
    if rank == 0:
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_split_equals_single_process_bit_for_bit(tmp_path, oracle, world):
    name = "wine_glass_c2"
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), name, "tiles", out), nprocs=world, join=True)
    frame = np.load(out)
    sc, flat = S.build(name)
    ref = oracle.render_positions(flat, S.positions(flat), linear=True)
    assert np.array_equal(frame, ref)


@pytest.mark.parametrize("world,name", [(2, "wine_glass_c2"), (3, "diamond_c4")])
def test_sample_split_sums_to_single_process(tmp_path, oracle, world, name):
    """Every rank evaluates its share of the outermost sample loops (and rank 0 the terms under no loop); the sum over the
    ranks is the unsharded radiance up to reassociation of the floating-point sums."""
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), name, "samples", out), nprocs=world, join=True)
    frame = np.load(out)
    sc, flat = S.build(name)
    ref = oracle.render_positions(flat, S.positions(flat), linear=True)
    assert np.abs(frame - ref).max() <= 1e-12
    # the shares are real shares: no rank alone produces the frame
    alone = oracle.render_positions(flat, S.positions(flat), linear=True, shard=(1, world))
    assert np.abs(alone - ref).max() > 1e-3 and (alone <= ref + 1e-12).all()


def test_tile_partition_is_a_partition():
    n = 1920 * 1080
    for world in (1, 2, 4, 8):
        seen = np.zeros(n, dtype=np.int32)
        sizes = []
        for r in range(world):
            idx = adist.rank_pixels(n, r, world)
            seen[idx] += 1
            sizes.append(len(idx))
            assert len(idx) <= adist.padded(n, world)
        assert (seen == 1).all()
        assert max(sizes) - min(sizes) <= adist.TILE
    # ragged: fewer tiles than ranks leaves some ranks empty, nothing is lost
    idx = [adist.rank_pixels(100, r, 8) for r in range(8)]
    assert sum(len(i) for i in idx) == 100 and len(idx[5]) == 0 and adist.padded(100, 8) == adist.TILE
    # a short last tile belongs to one rank only
    idx = [adist.rank_pixels(256 * 5 + 7, r, 4) for r in range(4)]
    assert [len(i) for i in idx] == [512, 256 + 7, 256, 256]      # six tiles: 0 1 2 3 | 0 1(short)
    pos = adist.pixel_positions(np.array([0, 1919, 1920]), 1920)
    assert pos.tolist() == [[0.5, 0.5], [1919.5, 0.5], [0.5, 1.5]]
