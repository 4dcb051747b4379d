"""Generates tests/golden/ref_ini_<CONFIG>.npz from the REFERENCE's own ini_zeta + ini_fields (ini_fields.F, the
first-step initialisation of main3d.F:269-283; oracle/_ref built by oracle/build_ref.sh): for five boundary-condition
tables and both index branches, the 2-D results (zeta, ubar, vbar, Zt_avg1) in full for two cases and a SHA-256 of
every other result (u, v, t; the 2-D fields of the other cases) -- the comparison is bit for bit, so a digest carries the same information as the array.  Run in this
container:

    python tests/golden/make_golden_ini.py
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
sys.path.insert(0, HERE)

CONFIGS = [("UPWELLING", None), ("UPWELLING", "island"), ("BENCHMARK_TINY", None)]
FULL = ("zeta", "ubar", "vbar", "Zt_avg1")
DIGEST = ("u", "v", "t")


def tag(config, mask):
    return config + ("_MASK" if mask else "")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest()      # a + 0.0: one sign of zero


def results(st, key="closed:1"):
    """2-D results in full for the two cases of the reference's first step that matter most, digests otherwise."""
    full = key in ("closed:1", "cha_fla_rad:1")
    out = {name: st[name].copy() for name in FULL} if full else {}
    out.update({name + "_sha256": np.array(digest(st[name])) for name in DIGEST + (() if full else FULL)})
    return out


def child(config, mask):
    from oracle import ref
    from ref_worker import ini_cases
    out = {}
    for key, st, s in ini_cases(config, mask):
        for kind in ("ini_zeta", "ini_fields"):
            ref.Ref(st).bc(kind, s, 0, 0)
        for name, val in results(st, key).items():
            out[key.replace(":", "__") + "__" + name] = val
    np.savez_compressed(os.path.join(HERE, f"ref_ini_{tag(config, mask)}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None)
    else:
        for c, m in CONFIGS:
            subprocess.run([sys.executable, os.path.abspath(__file__), c, m or "-"], check=True)
            print(tag(c, m), os.path.getsize(os.path.join(HERE, f"ref_ini_{tag(c, m)}.npz")) // 1024, "KiB")
