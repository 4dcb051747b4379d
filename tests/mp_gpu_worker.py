"""Worker for the multi-tile GPU tests: each rank runs the HIP library on one tile
(all ranks may share one GPU) and the halo exchange goes through the library's
host-relay transport (roms_hip_set_halo_relay) over gloo -- the same pack/unpack
kernels, neighbour table and phase order as the RCCL transport."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_rank(rank, world, ntI, ntJ, config, nsteps, port, outdir, variant=""):
    import torch
    import torch.distributed as dist
    from roms_trunk_mgh_amd import ana, hip, main3d
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    st = ana.make_tile(config, ntileI=ntI, ntileJ=ntJ, tile=rank, perturb=1.0, **kw)
    if "river" in opts:                  # point sources (LuvSrc) in the walls and, with a mask, on the island's coast
        import util
        util.river_sources(st, "all" if "wells" in opts else "both" if "mask" in opts else "walls")
    b = st.b
    ndev = torch.cuda.device_count()
    if "rccl" in opts:
        # the RCCL transport: one device per rank (RCCL refuses two ranks on one device); with ONE rank the
        # library runs in loopback (the tile is its own W/E neighbour, roms_hip.h)
        assert world <= max(ndev, 1)
        import ctypes
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            assert hip.load().roms_hip_get_unique_id(buf) == 0
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        be = hip.RomsHip(st, rank=rank, device=rank, nccl_unique_id=bytes(t.numpy().tobytes()))
    else:
        be = hip.RomsHip(st, rank=rank, device=rank % max(ndev, 1), nccl_unique_id=None)
        be.set_halo_relay_gloo(dist, torch)
    m = main3d.Main3D(be, physics=("physics" in opts), diagnostics=("physics" in opts))
    m.initial()
    m.run(nsteps)
    be.to_host()
    be.close()
    out = {k: st[k] for k in ("zeta", "ubar", "vbar", "u", "v", "t", "Huon", "W", "Hz", "Akv", "tke", "rmask_wet", "umask_wet",
                                 "vmask_wet", "pmask_wet", "rmask_wet_avg")}
    if "slim" in opts:      # full-size grids: only the newest time level of the 3-D prognostic fields
        lev = m.s.nnew - 1
        out["u"], out["v"], out["t"] = st["u"][:, :, :, lev], st["v"][:, :, :, lev], st["t"][:, :, :, lev, :]
    np.savez(os.path.join(outdir, f"tile{rank}.npz"),
             bounds=np.array([b.Istr, b.Iend, b.Jstr, b.Jend, b.LBi, b.LBj]), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    run_rank(int(a[1]), int(a[2]), int(a[3]), int(a[4]), a[5], int(a[6]), int(a[7]), a[8], a[9] if len(a) > 9 else "")
