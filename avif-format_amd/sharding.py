"""Row-tile sharding of one image across ranks: contiguous tiles, even boundaries so no 2x2 / 2x1
chroma block straddles a tile, no data dependence between tiles => no collective on the data path."""
from __future__ import annotations


def row_cut(height: int, world: int, k: int, even: bool = True) -> int:
    if k >= world:
        return height
    b = (height * k) // world
    if even:
        b -= b & 1
    return b


def row_tile(height: int, world: int, rank: int, even: bool = True):
    """(row0, nrows) of `rank`'s tile.  Tiles are disjoint, ordered, cover [0, height)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    r0 = row_cut(height, world, rank, even)
    r1 = row_cut(height, world, rank + 1, even)
    return r0, r1 - r0


def all_tiles(height: int, world: int, even: bool = True):
    return [row_tile(height, world, r, even) for r in range(world)]
