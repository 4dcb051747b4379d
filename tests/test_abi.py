"""CPU: the C-ABI library loads without a GPU and exports every symbol
include/roms_hip.h declares; struct layouts match the ctypes mirror; the HIP
path fails loudly (no fallback) when no device is present."""
import ctypes
import os
import re

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
