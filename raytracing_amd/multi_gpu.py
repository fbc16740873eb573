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
    """Rows of every rank's packed buffer; requires an even split so the all-gather is regular."""
    if height % (band * world):
        raise ValueError(f"height {height} must be a multiple of band*world = {band * world}")
    return height // world


def assemble(gathered, world, height, width, band=BAND_ROWS):
    """gathered: tensor (world * height/world, width, 3), the ranks' packed buffers one after the
    other as all_gather_into_tensor leaves them -> (height, width, 3) in image order."""
    nb = height // (band * world)
    return gathered.view(world, nb, band, width, 3).permute(1, 0, 2, 3, 4).reshape(height, width, 3)
