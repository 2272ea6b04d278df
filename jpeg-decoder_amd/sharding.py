"""Work partitioning across GPUs: independent units, no exchange step.

  * images  : image i of a batch goes to rank i mod world (round-robin, equal work)      -> C5
  * stripes : a single large frame is cut into contiguous MCU-row stripes, rank g gets rows
              [g*nbmcuV/world, (g+1)*nbmcuV/world) of every component and writes the matching
              max_v*8-row band of the raster (replicate upsampling has no halo,
              src/decoder_utils.c:72-80)                                                 -> C2..C4
"""


def shard_images(nimages, rank, world):
    """Indices of the images rank <rank> decodes."""
    return list(range(rank, nimages, world))


def shard_mcu_rows(nbmcuV, rank, world):
    """[m0, m1): the MCU rows of one frame that rank <rank> decodes (balanced to within one row)."""
    return (rank * nbmcuV) // world, ((rank + 1) * nbmcuV) // world


def stripe_planes(desc, planes, m0, m1):
    """Sub-planes (views) holding MCU rows [m0, m1) of every component, and the stripe's descriptor."""
    sub = type(desc).from_buffer_copy(bytes(desc))
    sub.nbmcuV = m1 - m0
    out = []
    for c in range(desc.nbcomp):
        per_row = desc.comp[c].vs * desc.nbmcuH * desc.comp[c].hs
        out.append(planes[c][m0 * per_row:m1 * per_row])
    return sub, out
