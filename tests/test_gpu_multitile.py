"""GPU, multi-process: the N>1 device path.  One process per tile (2 and 4 tiles, all on
the one GPU of the test box), halos through the library's host-relay transport over gloo.
After 3 full steps every tile's owned AND ghost points must equal the single-tile HIP run
bit for bit -- the reference's acceptance rule "identical results across tilings" -- which
exercises the general (multi-tile) branch of every kernel, the pack/unpack kernels, the
neighbour table incl. the periodic Nghost+1 rule and the corner messages of the one-phase exchange
(2x2: each tile's W and E, and its two diagonal neighbours, are the same rank -- four messages per pair,
paired by order as RCCL does).
The RCCL calls themselves need one GPU per rank and run only in bench.py --gpus N."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import util
from roms_trunk_mgh_amd import ana, main3d

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single(config, nsteps, variant=""):
    from roms_trunk_mgh_amd import hip
    opts = set(variant.split("+")) if variant else set()
    kw = dict(NT=6, overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"}) if "mpdata" in opts else {}
    if "mask" in opts:
        kw["mask"] = "island"
    if "dif4" in opts:                   # biharmonic mixing (three ghost points with UV_VIS4)
        kw.setdefault("overrides", {}).update({"ts_dif4": 1, "uv_vis4": 1, "tnu4": 1.0e10, "visc4": 2.0e10})
    if "basin" in opts:
        kw.setdefault("overrides", {})["EWperiodic"] = False
    if "wet" in opts:                    # WET_DRY on the beach bathymetry of ana.py (the shoreline crosses tile edges)
        kw.setdefault("overrides", {}).update({"wet_dry": 1, "beach": 1, "zeta_amp": 0.3})
    if "classic" in opts:                # without SPLINES_VVISC / SPLINES_VDIFF: the tridiagonal systems for u, v, t themselves
        kw.setdefault("overrides", {}).update({"splines_vdiff": 0, "splines_vvisc": 0})
    if "gls" in opts:                    # GLS_MIXING (k-epsilon, Kantha-Clayson, N2S2_HORAVG, RI_SPLINES)
        kw.setdefault("overrides", {})["gls"] = "k-epsilon"
    if "my25" in opts:                   # MY25_MIXING (Kantha-Clayson, N2S2_HORAVG, RI_SPLINES)
        kw.setdefault("overrides", {})["gls"] = "my25"
    if "geouv" in opts:                  # UV_VIS2 with MIX_GEO_UV (uv3dmix2_geo.h)
        kw.setdefault("overrides", {}).update({"uv_vis2": 2, **({"visc2": 50.0} if config == "SEAMOUNT" else {})})
    st = ana.make_tile(config, perturb=1.0, **kw)
    if "river" in opts:                  # point sources (LuvSrc) in the walls and, with a mask, on the island's coast
        util.river_sources(st, "all" if "wells" in opts else "both" if "mask" in opts else "walls")
    be = hip.RomsHip(st)
    m = main3d.Main3D(be, physics=("physics" in opts), diagnostics=("physics" in opts))
    m.initial()
    m.run(nsteps)
    be.to_host()
    be.close()
    return st, m.s.nnew - 1


@pytest.mark.parametrize("ntI,ntJ,config,variant", [(2, 1, "BENCHMARK_TINY", ""), (1, 2, "UPWELLING", ""),
                                                    (2, 2, "SEAMOUNT", ""), (4, 1, "BENCHMARK_TINY", ""),
                                                    # ragged tiles (64 = 22 + 21 + 21 columns), three different neighbours
                                                    (3, 1, "BENCHMARK_TINY", ""),
                                                    # three ghost points, MPDATA's extended ranges across tile edges
                                                    (2, 2, "BENCHMARK_TINY", "mpdata"),
                                                    # bulk fluxes, KPP, wvelocity, diag on every tile
                                                    (2, 2, "BENCHMARK_TINY", "physics"),
                                                    # MASKING: an island across the tile corner, a headland on the wall
                                                    (2, 2, "BENCHMARK_TINY", "mask"), (2, 1, "UPWELLING", "mask"),
                                                    # no periodic direction: physical edges on the outer tile sides, corners
                                                    (2, 2, "UPWELLING", "basin"), (2, 1, "BENCHMARK_TINY", "basin+physics"),
                                                    # MPDATA on a basin with land, across tile edges
                                                    (2, 2, "BENCHMARK_TINY", "mpdata+basin+mask"),
                                                    # the GLS closure: smoothed shear, five-point advection of tke / gls and
                                                    # the Akv / Akt edge rule of gls_corstep.F across tile edges
                                                    (2, 2, "UPWELLING", "gls"), (2, 1, "BENCHMARK_TINY", "gls+basin+mask"),
                                                    (2, 2, "UPWELLING", "my25"), (1, 2, "BENCHMARK_TINY", "my25+basin+mask"),
                                                    # UV_VIS2 rotated to geopotentials: the LDS patches of k_uv3dmix2_geo against tile edges
                                                    (2, 2, "SEAMOUNT", "geouv"), (2, 1, "BENCHMARK_TINY", "geouv+basin+mask"), (3, 1, "BENCHMARK_TINY", "geouv+wet"),
                                                    # WET_DRY: masks, their fast-time sum and the drying shoreline across
                                                    # tile edges
                                                    (2, 2, "UPWELLING", "wet"), (2, 2, "UPWELLING", "wet+basin+mask"),
                                                    # biharmonic mixing across tile edges
                                                    (2, 2, "BENCHMARK_TINY", "dif4"), (2, 2, "BENCHMARK_TINY", "dif4+basin+mask"),
                                                    # point sources (LuvSrc): rivers in the walls and on the island's coast
                                                    (2, 2, "UPWELLING", "river+basin+mask"), (2, 1, "BENCHMARK_TINY", "river+physics"),
                                                    (2, 2, "BENCHMARK_TINY", "river+mpdata+basin+mask"),
                                                    (2, 2, "UPWELLING", "river+wells+basin+mask"), (2, 2, "BENCHMARK_TINY", "river+wells+mpdata"),
                                                    (2, 2, "UPWELLING", "river+wells+wet+basin+mask"),
                                                    # without SPLINES_VVISC / SPLINES_VDIFF
                                                    (2, 2, "UPWELLING", "classic+river+wells+basin+mask"), (2, 1, "BENCHMARK_TINY", "classic+physics"),
                                                    # BASELINE.json configurations 4 and 5 at FULL size (2048x256x30): the
                                                    # 512-column tiles of the 8-GPU run (4x1), both tile rows (2x2), the
                                                    # deferred-flux step2d path and, with six MPDATA tracers, three ghost points
                                                    (4, 1, "BENCHMARK3", "physics+slim"), (2, 2, "BENCHMARK3", "physics+slim"),
                                                    (2, 2, "BENCHMARK3", "mpdata+slim")])
def test_tiled_hip_equals_single_hip(tmp_path, ntI, ntJ, config, variant):
    slim = "slim" in variant
    nsteps = 2 if slim else 3
    world = ntI * ntJ
    ref, lev = _single(config, nsteps, variant)     # before the children start: at most `world` + 1 GPU processes
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_gpu_worker.py"), str(r), str(world), str(ntI),
                               str(ntJ), config, str(nsteps), str(port), str(tmp_path), variant], env=env)
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    rb = ref.b
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"tile{r}.npz"))
        Istr, Iend, Jstr, Jend, LBi, LBj = [int(x) for x in d["bounds"]]
        for name in ("zeta", "ubar", "vbar", "u", "v", "t", "Huon", "W", "Hz", "Akv", "tke", "rmask_wet", "umask_wet",
                     "vmask_wet", "pmask_wet", "rmask_wet_avg"):
            a = d[name]
            ni, nj = a.shape[0], a.shape[1]
            i0, j0 = LBi - rb.LBi, LBj - rb.LBj
            full = ref[name]
            if slim and name in ("u", "v"):
                full = full[:, :, :, lev]
            elif slim and name == "t":
                full = full[:, :, :, lev, :]
            want = full[i0:i0 + ni, j0:j0 + nj]
            own = (slice(Istr - LBi, Iend - LBi + 1), slice(Jstr - LBj, Jend - LBj + 1))
            assert np.array_equal(a[own], want[own]), (name, r, float(np.abs(a[own] - want[own]).max()))
            if name in ("zeta", "t", "Hz", "W"):      # rho-type: every ghost point is defined
                iv = min(ni, rb.Lm + rb.NghostPoints - LBi + 1)
                jv = min(nj, rb.Mm + 1 - LBj + 1)
                assert np.array_equal(a[:iv, :jv], want[:iv, :jv]), (name, r, "ghost points differ")
