"""Frame split over devices: 8-row bands dealt round-robin to ranks, one all-gather to combine.

The render loop has no cross-pixel data flow, and the counter generator keys every sample by
(seed, pixel, sample), so a frame shards over devices without any exchange during rendering
(the reference's tile loop, src/main.c:804-833, shards over threads the same way).  Rank r
renders bands r, r + world, r + 2*world, ... into a packed buffer (RT_GPU_FLAG_COMPACT_BANDS);
an all-gather moves every rank's packed rows to every rank, and a view permutation puts the
bands back in image order.  Fine bands spread the expensive image regions over all ranks.
"""

BAND_ROWS = 8


def rows_of_rank(rank, world, height, band=BAND_ROWS):
    """Image rows owned by `rank`, in the order they sit in its packed buffer."""
    rows = []
    b = rank
    while b * band < height:
        rows.extend(range(b * band, min(height, (b + 1) * band)))
        b += world
    return rows


def packed_rows(world, height, band=BAND_ROWS):
    """Rows of every rank's packed buffer: the bands of the rank with the most bands (the same for all, so that
    the all-gather is regular; a rank with fewer bands, or a short last band, leaves padding rows).  The same
    number as rt_gpu_multi_rows_per_rank of the C entry."""
    bands = (height + band - 1) // band
    return ((bands + world - 1) // world) * band


_ROW_MAPS = {}


def assemble(gathered, world, height, width, band=BAND_ROWS):
    """gathered: tensor (world * packed_rows, width, 3), the ranks' packed buffers one after the other as
    all_gather_into_tensor leaves them -> (height, width, 3) in image order.  An even split is a view
    permutation; otherwise the rows are picked by index (image row y lives in band y // band of rank
    (y // band) % world, at packed row (y // band // world) * band + y % band)."""
    if height % (band * world) == 0:
        nb = height // (band * world)
        return gathered.view(world, nb, band, width, 3).permute(1, 0, 2, 3, 4).reshape(height, width, 3)
    import torch
    key = (world, height, band, str(gathered.device))
    if key not in _ROW_MAPS:
        per_rank = packed_rows(world, height, band)
        rows = [((y // band) % world) * per_rank + (y // band // world) * band + y % band for y in range(height)]
        _ROW_MAPS[key] = torch.tensor(rows, dtype=torch.long, device=gathered.device)
    return gathered.index_select(0, _ROW_MAPS[key])


# --------------------------------------------------------------------------- split by samples
#
# For small frames (few pixels per device) the other natural split is by samples: rank r renders
# samples [offset, offset + count) of EVERY pixel as un-normalised sums (RT_GPU_FLAG_ACCUMULATE
# into a zeroed buffer, sample_offset = offset), the partial sums are gathered and added in rank
# order (the same order on every rank, so all ranks hold identical bits), and the sum is scaled
# by the reference's final factor (src/main.c:788).  The per-pixel sum is then grouped by rank
# instead of running sample by sample, so the frame equals the single-device one up to float
# summation order, not bit for bit; the band split above is the default for that reason.

def sample_range(rank, world, spp):
    """(first sample, number of samples) of `rank`: an even split, the remainder to the low ranks."""
    base, extra = divmod(spp, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def pixel_norm(spp_total):
    """The reference's final factor 1/(0.5*0.5*2*pi*spp) as it rounds it: the float product 0.5f widened,
    the rest in double, the quotient narrowed to float (src/main.c:760,788)."""
    import math
    import numpy as np
    return np.float32(1.0 / (0.5 * math.pi * spp_total))


def combine_partial_sums(gathered, world, spp_total):
    """gathered: (world * H, W, 3) partial sums as all_gather_into_tensor leaves them -> (H, W, 3) frame."""
    parts = gathered.view(world, -1, gathered.shape[-2], 3)
    total = parts[0].clone()
    for r in range(1, world):
        total += parts[r]
    return total * float(pixel_norm(spp_total))
