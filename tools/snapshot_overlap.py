"""Developer tool (GPU box): cost of an asynchronous snapshot of the prognostic fields on BENCHMARK3 --
ms/step of 12 steps without a snapshot, with a synchronous sync_to_host after step 4, and with
snapshot_begin after step 4 / snapshot_end after step 12."""
import os
import sys
import time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
from roms_trunk_mgh_amd import ana, hip, main3d  # noqa: E402

names = ["zeta", "ubar", "vbar", "u", "v", "t"]
st = ana.make_tile("BENCHMARK3", perturb=1.0)
nbytes = sum(st[n].nbytes for n in names)
be = hip.RomsHip(st)
m = main3d.Main3D(be, physics=True, diagnostics=True)
m.initial()
m.run(3)
be.snapshot_begin(names); be.snapshot_end()          # page-locks the host arrays once
for mode in ("none", "sync", "async", "none", "sync", "async"):
    be.sync()
    t0 = time.perf_counter()
    m.run(4)
    if mode == "sync":
        be.to_host(names)
    elif mode == "async":
        be.snapshot_begin(names)
    m.run(8)
    if mode == "async":
        be.snapshot_end()
    be.sync()
    dt = time.perf_counter() - t0
    print(f"{mode:6s} 12 steps {1e3*dt:8.2f} ms  ({1e3*dt/12:.3f} ms/step)  snapshot {nbytes/1e9:.2f} GB", flush=True)
be.close()
