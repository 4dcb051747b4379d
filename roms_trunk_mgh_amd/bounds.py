"""Tile bounds: host-side mirror of the reference's get_tile / var_bounds /
get_bounds (ROMS/Utility/get_bounds.F:738, :1009, :2) and get_domain_edges
(:405-640), for the non-nested case.  Produces the ``roms_bounds_t`` block the
kernels take (every integer of ROMS/Include/set_bounds.h).
"""
from .abi import Bounds


def tile_bounds_2d(Imax, Jmax, ntileI, ntileJ, tile):
    """get_bounds.F:933-1007 (tile_bounds_2d)."""
    ChunkSizeI = (Imax + ntileI - 1) // ntileI
    ChunkSizeJ = (Jmax + ntileJ - 1) // ntileJ
    MarginI = (ntileI * ChunkSizeI - Imax) // 2
    MarginJ = (ntileJ * ChunkSizeJ - Jmax) // 2
    Jtile = tile // ntileI
    Itile = tile - Jtile * ntileI
    Istr = 1 + Itile * ChunkSizeI - MarginI
    Iend = Istr + ChunkSizeI - 1
    Istr = max(Istr, 1)
    Iend = min(Iend, Imax)
    Jstr = 1 + Jtile * ChunkSizeJ - MarginJ
    Jend = Jstr + ChunkSizeJ - 1
    Jstr = max(Jstr, 1)
    Jend = min(Jend, Jmax)
    return Itile, Jtile, Istr, Iend, Jstr, Jend


def _var_bounds_1d(my_str, my_end, low_edge, high_edge, periodic, Lm):
    """One direction of var_bounds (get_bounds.F:1348-1598 for I, :1600-1853
    for J).  Returns a dict keyed by the I-direction names."""
    d = {}
    if low_edge and not periodic:
        d["str"] = my_str
        d["strP"] = my_str
        d["strR"] = my_str - 1
        d["strT"] = d["strR"]
        d["strU"] = my_str + 1
        d["strB"] = d["strT"] + 1
        d["strM"] = d["strP"] + 1
        d["strm3"] = max(0, my_str - 3)
        d["strm2"] = max(0, my_str - 2)
        d["strUm2"] = max(1, d["strU"] - 2)
        d["strm1"] = max(1, my_str - 1)
        d["strUm1"] = max(2, d["strU"] - 1)
    else:
        d["str"] = d["strP"] = d["strR"] = d["strT"] = my_str
        d["strU"] = d["strB"] = d["strM"] = my_str
        d["strm3"] = my_str - 3
        d["strm2"] = my_str - 2
        d["strUm2"] = d["strU"] - 2
        d["strm1"] = my_str - 1
        d["strUm1"] = d["strU"] - 1
    if high_edge and not periodic:
        d["end"] = my_end
        d["endR"] = my_end + 1
        d["endP"] = d["endR"]
        d["endT"] = d["endR"]
        d["endB"] = d["endT"] - 1
        d["endp1"] = min(my_end + 1, Lm)
        d["endp2i"] = min(my_end + 2, Lm)
        d["endp2"] = min(my_end + 2, Lm + 1)
        d["endp3"] = min(my_end + 3, Lm + 1)
    else:
        d["end"] = d["endR"] = d["endP"] = d["endT"] = d["endB"] = my_end
        d["endp1"] = my_end + 1
        d["endp2i"] = my_end + 2
        d["endp2"] = my_end + 2
        d["endp3"] = my_end + 3
    return d


def make_bounds(Lm, Mm, N, NT, NAT, ntileI=1, ntileJ=1, tile=0,
                EWperiodic=True, NSperiodic=False, NghostPoints=2):
    """Build the roms_bounds_t of one tile (DISTRIBUTE semantics: each tile's
    arrays cover the tile plus NghostPoints, get_bounds.F:164-183)."""
    Itile, Jtile, Istr, Iend, Jstr, Jend = tile_bounds_2d(Lm, Mm, ntileI, ntileJ, tile)
    west = Itile == 0
    east = Itile == ntileI - 1
    south = Jtile == 0
    north = Jtile == ntileJ - 1
    di = _var_bounds_1d(Istr, Iend, west, east, EWperiodic, Lm)
    dj = _var_bounds_1d(Jstr, Jend, south, north, NSperiodic, Mm)

    # get_bounds.F:20-183, gtype=0 branch (full extents incl. ghost points).
    # Im, Jm carry the reference's even-size padding (mod_param.F initialize_param:
    # I_padd=(Lm+2)/2-(Lm+1)/2, Im=Lm+I_padd), so an even Lm allocates one spare column.
    Im = Lm + ((Lm + 2) // 2 - (Lm + 1) // 2)
    Jm = Mm + ((Mm + 2) // 2 - (Mm + 1) // 2)
    Imin = -NghostPoints if EWperiodic else 0
    Imax = Im + NghostPoints if EWperiodic else Im + 1
    Jmin = -NghostPoints if NSperiodic else 0
    Jmax = Jm + NghostPoints if NSperiodic else Jm + 1
    LBi = Imin if Itile == 0 else Istr - NghostPoints
    UBi = Imax if Itile == ntileI - 1 else Iend + NghostPoints
    LBj = Jmin if Jtile == 0 else Jstr - NghostPoints
    UBj = Jmax if Jtile == ntileJ - 1 else Jend + NghostPoints

    b = Bounds()
    b.Lm, b.Mm, b.N, b.NT, b.NAT = Lm, Mm, N, NT, NAT
    b.ntileI, b.ntileJ, b.tile, b.Itile, b.Jtile = ntileI, ntileJ, tile, Itile, Jtile
    b.NghostPoints = NghostPoints
    b.EWperiodic, b.NSperiodic = int(EWperiodic), int(NSperiodic)
    b.west_edge, b.east_edge = int(west), int(east)
    b.south_edge, b.north_edge = int(south), int(north)
    b.LBi, b.UBi, b.LBj, b.UBj = LBi, UBi, LBj, UBj
    for pre, d in (("I", di), ("J", dj)):
        for k, v in d.items():
            name = pre + k
            if k == "strU" and pre == "J":
                name = "JstrV"
            elif k == "strUm2" and pre == "J":
                name = "JstrVm2"
            elif k == "strUm1" and pre == "J":
                name = "JstrVm1"
            setattr(b, name, v)
    return b
