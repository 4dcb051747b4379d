"""Generates tests/golden/ref_atm_UPWELLING.npz from the REFERENCE's own prsgrd built with -DATM_PRESS
(oracle/_ref/UPWELLING_ATM, _ATM_PG31, _ATM_PJ; oracle/build_ref.sh): the air-pressure term of prsgrd32.h:264-266,
prsgrd31.h:213-215 / :294-296 and prsgrd40.h:194-196 on the state of tests/util.prepared_state with the pressure
field of tests/ref_worker.atm_pressure.  Stored per variant: every second point of three levels of ru, rv and a
SHA-256 of the whole arrays.  One child process per variant (one reference library each).  Run in this container:

    python tests/golden/make_golden_atm.py
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
VARIANTS = {"dj": 0, "pg31": 1, "pj": 3}


def prepare(variant):
    import ref_worker
    import util
    st = util.prepared_state("UPWELLING", overrides={"tnu2": 300.0, "visc2": 800.0, "atm_press": 1})
    st.p.pgf = VARIANTS[variant]
    ref_worker.atm_pressure(st)
    st["Zt_avg1"] *= 1.3
    return st, util.step_idx()


def results(st, variant):
    out = {}
    N = st.b.N
    for name in ("ru", "rv"):
        a = st[name][:, :, :, 0]
        out[f"{variant}/{name}_levels"] = a[::2, ::2][:, :, [1, N // 2, N]].copy()
        out[f"{variant}/{name}_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest())
    return out


def child(variant):
    from oracle import ref
    st, s = prepare(variant)
    ref.Ref(st).call("prsgrd", s)
    np.savez_compressed(os.path.join(HERE, f"_atm_{variant}.npz"), **results(st, variant))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        out = {}
        for v in VARIANTS:
            subprocess.run([sys.executable, os.path.abspath(__file__), v], check=True)
            part = os.path.join(HERE, f"_atm_{v}.npz")
            out.update(dict(np.load(part)))
            os.remove(part)
        np.savez_compressed(os.path.join(HERE, "ref_atm_UPWELLING.npz"), **out)
        print(os.path.getsize(os.path.join(HERE, "ref_atm_UPWELLING.npz")) // 1024, "KiB")
