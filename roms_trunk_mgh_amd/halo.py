"""Host-side mirror of the halo-exchange protocol (ROMS/Utility/mp_exchange.F):
the neighbour table of `tile_neighbors` (:73-286) and the two-phase W/E then
S/N ghost update of `mp_exchange2d/3d/4d` (:290/1413/2753).

The product path does this on the device (csrc/halo.hip: pack kernel -> grouped
RCCL send/recv -> unpack kernel).  This numpy + torch.distributed(gloo)
restatement exists so that the N>1 protocol -- neighbour ranks, ghost widths
incl. the periodic Nghost+1 rule, phase order, corner propagation -- is covered
by world_size-2 CPU tests, and to cross-check the library's
`roms_hip_tile_neighbors`.
"""
import numpy as np


def tile_neighbors(rank, ntileI, ntileJ, Nghost, NghostPoints, EWperiodic, NSperiodic):
    """mp_exchange.F:73-286.  Tiles are ranked row-major: rank = Jtile*NtileI + Itile."""
    I, J = rank % ntileI, rank // ntileI

    def table(i, j):
        return j * ntileI + i if (0 <= i < ntileI and 0 <= j < ntileJ) else -1

    n = dict(GsendW=Nghost, GsendE=Nghost, GrecvW=Nghost, GrecvE=Nghost,
             GsendS=Nghost, GsendN=Nghost, GrecvS=Nghost, GrecvN=Nghost)
    n["Wtile"], n["Etile"] = table(I - 1, J), table(I + 1, J)
    if EWperiodic and ntileI > 1:
        if table(I - 1, J) < 0:
            n["Wtile"] = table(ntileI - 1, J)
            if NghostPoints != 3:
                n["GrecvW"] = Nghost + 1
        elif table(I + 1, J) < 0:
            n["Etile"] = table(0, J)
            if NghostPoints != 3:
                n["GsendE"] = Nghost + 1
    n["Stile"], n["Ntile"] = table(I, J - 1), table(I, J + 1)
    if NSperiodic and ntileJ > 1:
        if table(I, J - 1) < 0:
            n["Stile"] = table(I, ntileJ - 1)
            if NghostPoints != 3:
                n["GrecvS"] = Nghost + 1
        elif table(I, J + 1) < 0:
            n["Ntile"] = table(I, 0)
            if NghostPoints != 3:
                n["GsendN"] = Nghost + 1
    return n


def exchange(A, b, rank, sendrecv):
    """Two-phase ghost update of A (shape (ni, nj, nk), Fortran order view of
    the tile array incl. ghosts).  `sendrecv(peer_lo, send_lo, peer_hi, send_hi,
    shape_lo, shape_hi)` -> (recv_lo, recv_hi) performs the paired transfers of
    one phase (None for a missing neighbour)."""
    n = tile_neighbors(rank, b.ntileI, b.ntileJ, b.NghostPoints, b.NghostPoints,
                       bool(b.EWperiodic), bool(b.NSperiodic))
    # a periodic direction held by one tile row/column is a local copy (exchange_2d.F)
    if b.EWperiodic and b.ntileI == 1:
        Lm, o = b.Lm, -b.LBi
        for m in range(1, b.NghostPoints + 1):
            A[Lm + m + o, :, :] = A[m + o, :, :]
        for m in range(0, 3):
            A[-m + o, :, :] = A[Lm - m + o, :, :]
    for d, (lo, hi, gsl, gsh, grl, grh, s, e, off) in enumerate((
            (n["Wtile"], n["Etile"], n["GsendW"], n["GsendE"], n["GrecvW"], n["GrecvE"], b.Istr, b.Iend, -b.LBi),
            (n["Stile"], n["Ntile"], n["GsendS"], n["GsendN"], n["GrecvS"], n["GrecvN"], b.Jstr, b.Jend, -b.LBj))):
        if lo < 0 and hi < 0:
            continue

        def sl(a0, g):
            idx = slice(a0 + off, a0 + off + g)
            return (idx, slice(None), slice(None)) if d == 0 else (slice(None), idx, slice(None))

        send_lo = np.ascontiguousarray(A[sl(s, gsl)]) if lo >= 0 else None           # my first interior lines
        send_hi = np.ascontiguousarray(A[sl(e - gsh + 1, gsh)]) if hi >= 0 else None  # my last interior lines
        shp_lo = A[sl(s - grl, grl)].shape if lo >= 0 else None
        shp_hi = A[sl(e + 1, grh)].shape if hi >= 0 else None
        recv_lo, recv_hi = sendrecv(lo, send_lo, hi, send_hi, shp_lo, shp_hi)
        if lo >= 0:
            A[sl(s - grl, grl)] = recv_lo
        if hi >= 0:
            A[sl(e + 1, grh)] = recv_hi


def gloo_sendrecv(dist, torch):
    """sendrecv implementation over torch.distributed (gloo, CPU tensors)."""
    def fn(lo, send_lo, hi, send_hi, shp_lo, shp_hi):
        reqs, recv_lo, recv_hi = [], None, None
        # when lo == hi (two tiles in a periodic direction) my low-side send pairs
        # with the peer's high-side receive: order sends (lo, hi), receives (hi, lo)
        if lo >= 0:
            reqs.append(dist.isend(torch.from_numpy(send_lo), dst=lo, tag=1))
        if hi >= 0:
            reqs.append(dist.isend(torch.from_numpy(send_hi), dst=hi, tag=2))
        if hi >= 0:
            recv_hi = np.empty(shp_hi)
            th = torch.from_numpy(recv_hi)
            reqs.append(dist.irecv(th, src=hi, tag=1))
        if lo >= 0:
            recv_lo = np.empty(shp_lo)
            tl = torch.from_numpy(recv_lo)
            reqs.append(dist.irecv(tl, src=lo, tag=2))
        for r in reqs:
            r.wait()
        return recv_lo, recv_hi
    return fn
