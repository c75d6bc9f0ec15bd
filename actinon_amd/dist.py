"""Multi-GPU partition of the sample space (pixels x path_samples) for one node, one process per GPU.  The partition
logic itself lives behind the C ABI (include/actinon_hip.h: acn_shard_tile_*, ACN_SHARD_SAMPLES); this module is the
numpy face of it for tests and bench.py.

Two splits:
  tiles    (default)  pixels are independent in the reference (src/scene.c:976-1011 touches only lum_arr->data[index]),
           so the frame is cut into tiles of 256 consecutive pixels dealt round-robin to the ranks; every rank renders
           LINEAR radiance for its tiles into a compact part, the parts are all-gathered (RCCL over xGMI on GPUs, gloo on
           CPU) and interleaved back (acn_shard_unpack_dev).  No value is ever added to another: the frame is bit-identical
           to a single-GPU render, and a rank sends 1 / world of the frame.
  samples  every rank renders every pixel but only its share of the iterations of the outermost sample loops
           (src/scene.c:556,596); the ranks' partial radiance buffers are sum-reduced.  For few pixels with many samples
           (BASELINE.json configs[4]).
The non-linear steps -- gamma, clamp, 8-bit pack (src/vectors.h:372-384, src/scene.c:76-82) -- run after the exchange."""
import numpy as np

from ._lib import hip

TILE = 256


def rank_count(n, rank, world):
    return int(hip.acn_shard_tile_count(n, rank, world))


def padded(n, world):
    return int(hip.acn_shard_tile_padded(n, world))


def rank_pixels(n_pixels, rank, world):
    """Indices (ascending) of the pixels rank `rank` of `world` renders: acn_shard_tile_index for i < acn_shard_tile_count."""
    cnt = rank_count(n_pixels, rank, world)
    i = np.arange(cnt, dtype=np.int64)
    idx = ((i // TILE) * max(world, 1) + rank) * TILE + (i % TILE) if world > 1 else i
    if cnt:   # spot-check the closed form against the library's function
        for k in (0, cnt // 2, cnt - 1):
            assert idx[k] == hip.acn_shard_tile_index(n_pixels, rank, world, k)
    return idx


def pixel_positions(idx, width):
    pos = np.empty((idx.shape[0], 2), dtype=np.float64)
    pos[:, 0] = (idx % width) + 0.5
    pos[:, 1] = (idx // width) + 0.5
    return pos


def render_frame_tiles(render_fn, width, height, rank, world, all_gather):
    """Tile split.  render_fn(pos[n,2]) -> linear rgb[n,3]; all_gather(part[padded,3]) -> [world, padded, 3].
    Returns the complete linear frame [H*W,3] on every rank."""
    n = width * height
    idx = rank_pixels(n, rank, world)
    part = np.zeros((padded(n, world), 3), dtype=np.float64)
    if idx.shape[0]:
        part[:idx.shape[0]] = render_fn(pixel_positions(idx, width))
    gathered = all_gather(part) if world > 1 else part[None]
    frame = np.empty((n, 3), dtype=np.float64)
    for r in range(world):
        ri = rank_pixels(n, r, world)
        frame[ri] = gathered[r][:ri.shape[0]]
    return frame


def render_frame_samples(render_shard_fn, width, height, rank, world, all_reduce_sum):
    """Sample split.  render_shard_fn(pos, rank, world) -> this rank's linear partial radiance for ALL positions;
    all_reduce_sum(frame) sums in place over the ranks."""
    from .scene import main_pass_positions
    frame = np.ascontiguousarray(render_shard_fn(main_pass_positions(width, height), rank, world))
    if world > 1:
        all_reduce_sum(frame)
    return frame
