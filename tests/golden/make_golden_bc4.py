"""Generates tests/golden/ref_bc4_<CONFIG>.npz from the REFERENCE's own boundary-condition routines on a BASIN (no
periodic direction: tests/ref_worker.basin_state): for every condition, set on all four edges at once, the boundary
lines of the variable after the call -- three columns (Istr-1, Istr, Iend+1) and three rows (Jstr-1, Jstr, Jend+1),
which contain the four corners (3-D variables: three levels in full and a SHA-256 of all levels).  Run in this container:

    python tests/golden/make_golden_bc4.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = [("UPWELLING", None), ("UPWELLING", "island"), ("BENCHMARK_TINY", None)]


def tag(config, mask):
    return config + ("_MASK" if mask else "")


def lines(st, var):
    """2-D variables: the lines in full.  3-D variables: levels 1, N/2, N in full and a SHA-256 of all of them."""
    import hashlib
    b = st.b
    a = st[var]
    cols = np.stack([a[i - b.LBi] for i in (b.Istr - 1, b.Istr, b.Iend + 1)])
    rows = np.stack([a[:, j - b.LBj] for j in (b.Jstr - 1, b.Jstr, b.Jend + 1)])
    if var in ("u", "v", "t"):
        h = hashlib.sha256(np.ascontiguousarray(cols + 0.0).tobytes() + np.ascontiguousarray(rows + 0.0).tobytes()).hexdigest()
        lev = [0, b.N // 2, b.N - 1]
        return cols[:, :, lev], rows[:, :, lev], h
    return cols, rows, ""


def child(config, mask):
    from oracle import ref
    from ref_worker import basin_state, basin_cases
    st0 = basin_state(config, mask)
    out = {}
    for key, kind, var, st, s, nout, itrc in basin_cases(st0):
        ref.Ref(st).bc(kind, s, nout, itrc)
        cols, rows, sha = lines(st, var)
        k = key.replace(":", "__")
        out[k + "__cols"], out[k + "__rows"], out[k + "__sha256"] = cols, rows, np.array(sha)
    np.savez_compressed(os.path.join(HERE, f"ref_bc4_{tag(config, mask)}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None)
    else:
        for c, m in CONFIGS:
            subprocess.run([sys.executable, os.path.abspath(__file__), c, m or "-"], check=True)
            print(tag(c, m), os.path.getsize(os.path.join(HERE, f"ref_bc4_{tag(c, m)}.npz")) // 1024, "KiB")
