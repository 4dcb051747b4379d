"""Generates tests/golden/ref_dif4_<CONFIG>.npz from the REFERENCE's own biharmonic operators -- t3dmix4_s_tile
(UPWELLING) / t3dmix4_geo_tile (SEAMOUNT) and uv3dmix4_s_tile, oracle/_ref/<APP>_DIF4 (+ UPWELLING_MASK_DIF4) built
by oracle/build_ref.sh from the application's options plus TS_DIF4 and UV_VIS4.  Variants: the periodic channel, a
basin with closed edges, a basin whose tracer / momentum conditions are "gradient" (the other branch of the rule for
the first operator's result on an edge), and for UPWELLING the island grid (MASKING).  Stored: t, u, v (nnew) at three
levels in full, rufrc, rvfrc, and a SHA-256 of all levels (the comparison is bit for bit).  Run in this container:

    python tests/golden/make_golden_dif4.py
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

CONFIGS = {"UPWELLING": ["periodic", "closed", "open", "mask"], "SEAMOUNT": ["periodic", "closed", "open"]}
DIF4 = {"ts_dif4": 1, "uv_vis4": 1, "tnu4": 2.0e7, "visc4": 4.0e7}


def input_state(config, variant):
    import util
    from roms_trunk_mgh_amd import abi
    ov = dict(DIF4)
    if variant in ("closed", "open"):
        ov["EWperiodic"] = False
    st = util.prepared_state(config, overrides=ov, mask="island" if variant == "mask" else None)
    if variant == "open":
        for sd in ("west", "east", "south", "north"):
            for var in ("u", "v", "t"):
                st.p.lbc[abi.LBS[sd]][abi.LBV[var]] = abi.LBC["Gra"]
    return st


def results(st, s, kernel):
    out = {}
    N = st.b.N
    ks = [0, N // 2, N - 1]

    def put(name, a):
        out[name + "_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest())

    if kernel == "t3dmix4":
        a = st["t"][:, :, :, s.nnew - 1, :]
        out["t_levels"] = a[:, :, ks, :].copy()
        put("t", a)
    else:
        for name in ("u", "v"):
            a = st[name][:, :, :, s.nnew - 1]
            out[name + "_levels"] = a[:, :, ks].copy()
            put(name, a)
        for name in ("rufrc", "rvfrc"):
            out[name] = st[name].copy()
    return out


def child(config, variant, path):
    """One process per variant: the reference keeps one set of bounds (and one build) per process."""
    import util
    from oracle import ref
    out = {}
    s = util.step_idx()
    for kernel in ("t3dmix4", "uv3dmix4"):
        st = input_state(config, variant)
        ref.Ref(st).call(kernel, s)
        for k, v in results(st, s, kernel).items():
            out[f"{variant}__{kernel}__{k}"] = v
    np.savez(path, **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], sys.argv[2], sys.argv[3])
    else:
        import tempfile
        for c, variants in CONFIGS.items():
            merged = {}
            with tempfile.TemporaryDirectory() as td:
                for v in variants:
                    part = os.path.join(td, v + ".npz")
                    subprocess.run([sys.executable, os.path.abspath(__file__), c, v, part], check=True)
                    merged.update(np.load(part))
            np.savez_compressed(os.path.join(HERE, f"ref_dif4_{c}.npz"), **merged)
            print(c, os.path.getsize(os.path.join(HERE, f"ref_dif4_{c}.npz")) // 1024, "KiB")
