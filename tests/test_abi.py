"""CPU: the C-ABI library loads without a GPU and exports every symbol
include/roms_hip.h declares; struct layouts match the ctypes mirror; the HIP
path fails loudly (no fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from roms_trunk_mgh_amd import abi, ana, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "roms_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(roms_hip_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(hip.DECLARED_SYMBOLS) <= set(names)


def test_struct_layouts_match():
    abi.check_abi(hip.load())
    import oracle
    abi.check_abi(oracle.lib())
    assert len(abi.FIELDS) == len(set(n for n, _, _ in abi.FIELDS))


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_no_cpu_fallback():
    st = ana.make_tile("SEAMOUNT")
    with pytest.raises(RuntimeError) as e:
        hip.RomsHip(st)
    assert "init" in str(e.value)


def test_entries_refuse_uninitialised_library():
    lib = hip.load()
    s = abi.StepIdx(iic=1, ntfirst=1, nstp=1, nnew=2, nrhs=1, kstp=1, krhs=1, knew=2, iif=1, predictor_2d_step=0)
    lib.roms_hip_finalize()
    rc = lib.roms_hip_step3d_t(ctypes.byref(s))
    assert rc != 0
    assert b"not initialised" in lib.roms_hip_last_error()


def test_tile_neighbors_matches_python_mirror():
    from roms_trunk_mgh_amd import halo
    for (nI, nJ) in [(1, 1), (2, 1), (4, 1), (4, 2), (2, 2), (1, 2), (3, 3)]:
        for ng in (2, 3):
            for ew in (True, False):
                for ns in (True, False):
                    for r in range(nI * nJ):
                        a = hip.tile_neighbors(r, nI, nJ, ng, ng, ew, ns)
                        bq = halo.tile_neighbors(r, nI, nJ, ng, ng, ew, ns)
                        assert a == bq, (nI, nJ, ng, ew, ns, r, a, bq)
    # the documented BENCHMARK3 4x2 case: rank 3 -> east neighbour 0
    n = hip.tile_neighbors(3, 4, 2, 2, 2, True, False)
    assert (n["Etile"], n["Wtile"], n["Ntile"], n["Stile"]) == (0, 2, 7, -1)
    assert n["GsendE"] == 3 and hip.tile_neighbors(0, 4, 2, 2, 2, True, False)["GrecvW"] == 3


@pytest.mark.parametrize("nI,nJ,ng", [(4, 2, 2), (4, 2, 3), (2, 2, 2), (3, 3, 2), (2, 1, 2), (1, 2, 2), (4, 1, 3), (3, 2, 2)])
def test_halo_plan_fills_every_ghost_point(nI, nJ, ng):
    """Play one halo update of a whole tiling on the host with the library's own message plan (one phase,
    up to eight messages per tile, corners from the diagonal tiles): k-th send of A to B pairs with the
    k-th receive of B from A (RCCL's rule, no tags), counts agree, and afterwards every ghost point of every
    tile holds the owner's value -- including across the periodic seam (Nghost+1 rule) and in the corners."""
    from roms_trunk_mgh_amd import bounds as B
    Lm, Mm = 48, 20
    G = np.arange((Lm + 1) * (Mm + 2), dtype=np.float64).reshape(Lm + 1, Mm + 2) + 0.5     # G[i, j], i = 1..Lm
    wrap = lambda i: (i - 1) % Lm + 1
    tiles = []
    for r in range(nI * nJ):
        b = B.make_bounds(Lm, Mm, 4, 2, 2, ntileI=nI, ntileJ=nJ, tile=r, NghostPoints=ng)
        a = np.full((b.UBi - b.LBi + 1, b.UBj - b.LBj + 1), np.nan)
        jlo = b.Jstr - 1 if b.south_edge else b.Jstr            # wall rows 0 and Mm+1 belong to the edge tiles
        jhi = b.Jend + 1 if b.north_edge else b.Jend
        for i in range(b.Istr, b.Iend + 1):
            a[i - b.LBi, jlo - b.LBj:jhi - b.LBj + 1] = G[i, jlo:jhi + 1]
        if nI == 1:                                             # one tile column: periodic copy is local
            for i in list(range(b.LBi, 1)) + list(range(Lm + 1, Lm + ng + 1)):
                a[i - b.LBi, :] = a[wrap(i) - b.LBi, :]
        tiles.append((b, a, *hip.halo_plan(b, r)))
    # post: queues per (src, dst) in list order
    queues = {}
    for r, (b, a, sends, recvs) in enumerate(tiles):
        for m in sends:
            blk = a[m["i0"] - b.LBi:m["i0"] - b.LBi + m["wi"], m["j0"] - b.LBj:m["j0"] - b.LBj + m["wj"]].copy()
            queues.setdefault((r, m["peer"]), []).append((m["tag"], blk))
    for r, (b, a, sends, recvs) in enumerate(tiles):
        assert len(sends) == len(recvs) <= 8
        for m in recvs:
            tag, blk = queues[(m["peer"], r)].pop(0)
            assert tag == m["tag"], (r, m, tag)
            assert blk.shape == (m["wi"], m["wj"]), (r, m, blk.shape)
            a[m["i0"] - b.LBi:m["i0"] - b.LBi + m["wi"], m["j0"] - b.LBj:m["j0"] - b.LBj + m["wj"]] = blk
    assert all(len(q) == 0 for q in queues.values())
    for r, (b, a, _, _) in enumerate(tiles):
        # periodic images the reference defines: i = -2..0 (three, whatever NghostPoints) and Lm+1..Lm+ng
        for i in range(max(b.LBi, -2), min(b.UBi, Lm + ng) + 1):
            for j in range(max(b.LBj, 0), min(b.UBj, Mm + 1) + 1):
                assert a[i - b.LBi, j - b.LBj] == G[wrap(i), j], (r, i, j)


def test_reduce_diag_and_host_clock():
    """Host-side helpers of main3d.py: the reduction of the tile-local diag vectors (diag.F:398-420) and the
    clock handed to ana_srflux (caldate with TIME_REF = 0; pinned against the reference in test_ref_pinning.py)."""
    from roms_trunk_mgh_amd import main3d
    a = np.array([10.0, 2.0, 3.0, 0.5, 27.0, 0.10, 0.04, 0.05, 0.01, 7, 8, 9])
    b = np.array([20.0, 1.0, 5.0, 0.7, 26.0, 0.30, 0.10, 0.15, 0.05, 17, 18, 19])
    r = main3d.reduce_diag([a, b])
    assert list(r[0:3]) == [30.0, 3.0, 8.0] and list(r[3:5]) == [0.7, 27.0]
    assert list(r[5:12]) == list(b[5:12])                      # MAXLOC: the tile with the larger Courant number
    assert main3d.host_clock(0.0) == (1.0, 0.0)
    yd, hr = main3d.host_clock(150.0 / 86400.0)                # one BENCHMARK step later
    assert abs(yd - (1.0 + 150.0 / 86400.0)) < 1e-12 and abs(hr - 150.0 / 3600.0) < 1e-12
    yd, hr = main3d.host_clock(10.5)
    assert abs(yd - 11.5) < 1e-12 and hr == 12.0
    with pytest.raises(ValueError):
        main3d.host_clock(400.0)
