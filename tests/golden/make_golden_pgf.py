"""Generates tests/golden/ref_pgf_<CONFIG>.npz from the REFERENCE's own prsgrd31_tile (prsgrd31.h: the standard
density Jacobian and, with WJ_GRADP, the weighted one) -- oracle/_ref/<APP>_PG31, <APP>_WJ built by
oracle/build_ref.sh from the application's options without DJ_GRADPS.  Stored: ru, rv(nrhs) at three levels in full
and a SHA-256 of all levels (the comparison is bit for bit).  Run in this container:

    python tests/golden/make_golden_pgf.py
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

CONFIGS = ["UPWELLING", "SEAMOUNT"]
VARIANTS = {"STANDARD": 1, "WJ_GRADP": 2, "PJ_GRADP": 3}


def input_state(config, variant):
    import util
    st = util.prepared_state(config)
    st.p.pgf = VARIANTS[variant]
    return st


def results(st, s):
    out = {}
    N = st.b.N
    for name in ("ru", "rv"):
        a = st[name][:, :, :, s.nrhs - 1]
        out[name + "_levels"] = a[:, :, [1, N // 2, N]].copy()            # k = 1, N/2, N (index 0 is the 2-D term)
        out[name + "_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest())
    return out


def child(config):
    import util
    from oracle import ref
    out = {}
    s = util.step_idx()
    for variant in VARIANTS:
        st = input_state(config, variant)
        ref.Ref(st).call("prsgrd", s)
        for k, v in results(st, s).items():
            out[f"{variant}__{k}"] = v
    np.savez_compressed(os.path.join(HERE, f"ref_pgf_{config}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for c in CONFIGS:
            subprocess.run([sys.executable, os.path.abspath(__file__), c], check=True)
            print(c, os.path.getsize(os.path.join(HERE, f"ref_pgf_{c}.npz")) // 1024, "KiB")
