"""Generates tests/golden/ref_kpp_BENCHMARK_TINY[_MASK].npz from the REFERENCE's own lmd_vmix (lmd_vmix_tile +
lmd_skpp_tile + lmd_finish_tile; oracle/_ref/BENCHMARK and BENCHMARK_MASK built by oracle/build_ref.sh) on the
stratified state of tests/util.kpp_state -- boundary layers ending inside the top layer, in the next few layers and
deep ones in one state, salinity diffusivities at levels 0 and N different from the temperature's.  (The KPP vectors
of ref_BENCHMARK_TINY.npz come from prepared_state, whose pden, bvf, alpha and beta are zero.)  Stored: hsbl in full,
Akv / Akt / ghats at six levels in full and a SHA-256 of all levels.  Run in this container:

    python tests/golden/make_golden_kpp.py
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

CASES = [None, "island"]
LEVELS = [0, 1, 5, 15, 27, 29, 30]          # W-levels 0..N of the 30-level grid


def tag(mask):
    return "BENCHMARK_TINY" + ("_MASK" if mask else "")


def results(st):
    out = {"hsbl": st["hsbl"].copy()}
    for name in ("Akv", "Akt", "ghats"):
        a = st[name]
        out[name + "_levels"] = a[:, :, LEVELS].copy()
        out[name + "_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest())
    return out


def child(mask):
    import util
    from oracle import ref
    st = util.kpp_state("BENCHMARK_TINY", mask=mask)
    ref.Ref(st).physics("lmd_vmix", util.step_idx())
    np.savez_compressed(os.path.join(HERE, f"ref_kpp_{tag(mask)}.npz"), **results(st))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(None if sys.argv[1] == "-" else sys.argv[1])
    else:
        for m in CASES:
            subprocess.run([sys.executable, os.path.abspath(__file__), m or "-"], check=True)
            print(tag(m), os.path.getsize(os.path.join(HERE, f"ref_kpp_{tag(m)}.npz")) // 1024, "KiB")
