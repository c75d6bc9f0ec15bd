"""Multi-GPU partition of the sample space (pixels x path_samples) for one node, one process per GPU.

Pixels are independent in the reference (src/scene.c:976-1011 touches only lum_arr->data[index]), so the frame is
cut into tiles of TILE consecutive pixels dealt round-robin to the ranks (interleaving balances glass / floor / sky
regions).  Every rank renders LINEAR radiance for its tiles into a zero-initialised full-frame accumulator; the
frame is completed by ONE sum all-reduce of the per-pixel accumulators (RCCL over xGMI on GPUs, gloo on CPU) and the
non-linear steps -- gamma, clamp, 8-bit pack (src/vectors.h:372-384, src/scene.c:76-82) -- run after the reduce.
Because supports are disjoint the reduced frame is bit-identical to a single-GPU render."""
import numpy as np

TILE = 256


def rank_pixels(n_pixels, rank, world, tile=TILE):
    """Indices (ascending) of the pixels rank `rank` of `world` renders."""
    idx = np.arange(n_pixels, dtype=np.int64)
    return idx[(idx // tile) % world == rank]


def pixel_positions(idx, width):
    pos = np.empty((idx.shape[0], 2), dtype=np.float64)
    pos[:, 0] = (idx % width) + 0.5
    pos[:, 1] = (idx // width) + 0.5
    return pos


def render_frame_sharded(render_fn, width, height, rank, world, all_reduce_sum, xp=np, tile=TILE):
    """render_fn(pos[n,2]) -> linear rgb[n,3] (array type of `xp`); all_reduce_sum(frame) sums in place over ranks.
    Returns the complete linear frame [H*W,3] on every rank."""
    n = width * height
    idx = rank_pixels(n, rank, world, tile)
    frame = xp.zeros((n, 3), dtype=xp.float64)
    if idx.shape[0]:
        rgb = render_fn(pixel_positions(idx, width))
        frame[idx] = rgb
    if world > 1:
        all_reduce_sum(frame)
    return frame
