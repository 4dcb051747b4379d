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
    kw = dict(NT=6, overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"}) if variant == "mpdata" else {}
    st = ana.make_tile(config, ntileI=ntI, ntileJ=ntJ, tile=rank, perturb=1.0, **kw)
    b = st.b
    ndev = torch.cuda.device_count()
    be = hip.RomsHip(st, rank=rank, device=rank % max(ndev, 1), nccl_unique_id=None)
    be.set_halo_relay_gloo(dist, torch)
    m = main3d.Main3D(be, physics=(variant == "physics"), diagnostics=(variant == "physics"))
    m.initial()
    m.run(nsteps)
    be.to_host()
    be.close()
    np.savez(os.path.join(outdir, f"tile{rank}.npz"),
             bounds=np.array([b.Istr, b.Iend, b.Jstr, b.Jend, b.LBi, b.LBj]),
             **{k: st[k] for k in ("zeta", "ubar", "vbar", "u", "v", "t", "Huon", "W", "Hz")})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    run_rank(int(a[1]), int(a[2]), int(a[3]), int(a[4]), a[5], int(a[6]), int(a[7]), a[8], a[9] if len(a) > 9 else "")
