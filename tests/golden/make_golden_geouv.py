"""Generates tests/golden/ref_geouv.npz from the REFERENCE's own uv3dmix2 built with MIX_GEO_UV instead of MIX_S_UV
(uv3dmix2_geo.h; oracle/_ref/UPWELLING_GEOUV, SEAMOUNT_GEOUV, UPWELLING_MASK_GEOUV, UPWELLING_MASK_WET_GEOUV built by
oracle/build_ref.sh) on the state of tests/util.prepared_state with uv_vis2 = 2: channel, the seamount (steep slopes,
closed basin, visc2 = 50 m2/s), the island grid, the island grid with WET_DRY masks.  Stored per case: u, v at
every second point of three levels, rufrc, rvfrc in full, and a SHA-256 of the whole arrays.  Run in this container:

    python tests/golden/make_golden_geouv.py
"""
import hashlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    "channel": dict(config="UPWELLING", mask=None, wet=False, overrides={}),
    "seamount": dict(config="SEAMOUNT", mask=None, wet=False, overrides={"visc2": 50.0}),
    "island": dict(config="UPWELLING", mask="island", wet=False, overrides={}),
    "island_wet": dict(config="UPWELLING", mask="island", wet=True, overrides={"wet_dry": 1}),
}
NAMES = ["u", "v", "rufrc", "rvfrc"]


def prepare(case):
    import util
    c = CASES[case]
    st = util.prepared_state(c["config"], overrides=dict(c["overrides"], uv_vis2=2), mask=c["mask"], wet=c["wet"] or None)
    return st, util.step_idx(iic=5, nrhs=1, nnew=2)


def results(st, s, prefix):
    out = {}
    N = st.b.N
    for name in NAMES:
        a = st[name]
        if a.ndim == 4:
            a = a[..., s.nnew - 1]
            out[f"{prefix}/{name}_levels"] = a[::2, ::2][:, :, sorted({0, N // 2, N - 1})].copy()
        else:
            out[f"{prefix}/{name}"] = a.copy()
        out[f"{prefix}/{name}_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest())
    return out


def child(case):
    from oracle import ref
    st, s = prepare(case)
    ref.Ref(st).call("uv3dmix2", s)
    np.savez_compressed(os.path.join(HERE, f"_geouv_{case}.npz"), **results(st, s, case))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        out = {}
        for case in CASES:           # one process per case: each loads another build of the reference
            subprocess.run([sys.executable, os.path.abspath(__file__), case], check=True)
            f = os.path.join(HERE, f"_geouv_{case}.npz")
            out.update(dict(np.load(f)))
            os.remove(f)
        np.savez_compressed(os.path.join(HERE, "ref_geouv.npz"), **out)
        print(os.path.getsize(os.path.join(HERE, "ref_geouv.npz")) // 1024, "KiB")
