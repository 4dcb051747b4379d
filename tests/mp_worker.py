"""Worker for the multi-process CPU tests: each rank runs the CPU oracle on one
tile and swaps halos through the Python mirror of mp_exchange over gloo."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_rank(rank, world, ntI, ntJ, config, nsteps, port, outdir, perturb=1.0, variant=""):
    import torch
    import torch.distributed as dist
    import oracle
    from roms_trunk_mgh_amd import ana, halo, main3d
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opts = set(variant.split("+")) if variant else set()
    kw = dict(NT=6, overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"}) if "mpdata" in opts else {}
    if "hsimt" in opts:
        kw = dict(overrides={"Hadv": "HSIMT", "Vadv": "HSIMT"})
    if "mask" in opts:
        kw["mask"] = "island"
    if "dif4" in opts:                   # biharmonic mixing (three ghost points with UV_VIS4)
        kw.setdefault("overrides", {}).update({"ts_dif4": 1, "uv_vis4": 1, "tnu4": 1.0e10, "visc4": 2.0e10})
    if "basin" in opts:                  # no periodic direction
        kw.setdefault("overrides", {})["EWperiodic"] = False
    if "classic" in opts:                # without SPLINES_VVISC / SPLINES_VDIFF: the tridiagonal systems for u, v, t themselves
        kw.setdefault("overrides", {}).update({"splines_vdiff": 0, "splines_vvisc": 0})
    if "gls" in opts:                    # GLS_MIXING (k-epsilon, Kantha-Clayson, N2S2_HORAVG, RI_SPLINES)
        kw.setdefault("overrides", {})["gls"] = "k-epsilon"
    if "my25" in opts:                   # MY25_MIXING (Kantha-Clayson, N2S2_HORAVG, RI_SPLINES)
        kw.setdefault("overrides", {})["gls"] = "my25"
    if "geouv" in opts:                  # UV_VIS2 with MIX_GEO_UV (uv3dmix2_geo.h)
        kw.setdefault("overrides", {}).update({"uv_vis2": 2, **({"visc2": 50.0} if config == "SEAMOUNT" else {})})
    if "wet" in opts:                    # WET_DRY on the beach bathymetry of ana.py (the shoreline crosses tile edges)
        kw.setdefault("overrides", {}).update({"wet_dry": 1, "beach": 1, "zeta_amp": 0.3})
    st = ana.make_tile(config, ntileI=ntI, ntileJ=ntJ, tile=rank, perturb=perturb, **kw)
    if "river" in opts:                  # point sources (LuvSrc) in the walls and, with a mask, on the island's coast
        import util
        util.river_sources(st, "all" if "wells" in opts else "both" if "mask" in opts else "walls")
    b = st.b
    ni, nj = st.ni, st.nj
    sr = halo.gloo_sendrecv(dist, torch)

    HOOK = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_int)

    def hook(ptr, nk, gtype):
        A = np.ctypeslib.as_array(ptr, shape=(nk * nj * ni,)).reshape((ni, nj, nk), order="F")
        halo.exchange(A, b, rank, sr)

    cb = HOOK(hook)
    lib = oracle.lib()
    lib.oracle_set_exchange_hook.argtypes = [HOOK]
    lib.oracle_set_exchange_hook(cb)
    m = main3d.Main3D(oracle.Oracle(st))
    m.initial()
    m.run(nsteps)
    lib.oracle_set_exchange_hook(HOOK(0))
    np.savez(os.path.join(outdir, f"tile{rank}.npz"),
             bounds=np.array([b.Istr, b.Iend, b.Jstr, b.Jend, b.LBi, b.LBj]),
             **{k: st[k] for k in ("zeta", "ubar", "vbar", "u", "v", "t", "Huon", "W", "Hz", "Akv", "tke", "rmask_wet", "umask_wet",
                                 "vmask_wet", "pmask_wet", "rmask_wet_avg")})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    run_rank(int(a[1]), int(a[2]), int(a[3]), int(a[4]), a[5], int(a[6]), int(a[7]), a[8],
             variant=a[9] if len(a) > 9 else "")
