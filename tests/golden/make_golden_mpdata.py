"""Generates tests/golden/ref_mpdata_<CONFIG>.npz from the REFERENCE's own mpdata_adiff_tile
(ROMS/Nonlinear/mpdata_adiff.F compiled with flang into oracle/_ref, see oracle/build_ref.sh).
Inputs: tests/util.prepared_state + util.mpdata_private_arrays (deterministic).  Stored: the
reference outputs Ua, Va, Wa at three levels (bottom, middle, top) and the SHA-256 of the full
output arrays, so the fixture stays small while every element is still checked.  A second fixture
(ref_mpdata_<CONFIG>_MASK.npz) comes from the MASKING build on the island grid.

    python tests/golden/make_golden_mpdata.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CONFIG = "BENCHMARK_TINY"


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a + 0.0).tobytes())     # +0.0: -0.0 and 0.0 hash alike
    return h.hexdigest()


def main(mask=None):
    import util
    from oracle import ref
    st = util.prepared_state(CONFIG, overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"}, mask=mask)
    oHz, Ta0, t3 = util.mpdata_private_arrays(st)
    nis, njs, N = Ta0.shape
    Ta = Ta0.copy(order="F")
    Ua = np.zeros((nis, njs, N), order="F")
    Va = np.zeros((nis, njs, N), order="F")
    Wa = np.zeros((nis, njs, N + 1), order="F")
    ref.Ref(st).mpdata_adiff(oHz, t3, Ta, Ua, Va, Wa)
    ks = [0, N // 2, N - 2]
    np.savez_compressed(os.path.join(HERE, f"ref_mpdata_{CONFIG}{'_MASK' if mask else ''}.npz"),
                        input_sha256=np.array(sha(oHz, Ta0, t3)), levels=np.array(ks),
                        Ua=Ua[:, :, ks], Va=Va[:, :, ks], Wa=Wa[:, :, [k + 1 for k in ks]], Ta=Ta[:, :, ks],
                        output_sha256=np.array(sha(Ta, Ua, Va, Wa)))


if __name__ == "__main__":
    import subprocess
    if len(sys.argv) > 1:
        main(None if sys.argv[1] == "-" else sys.argv[1])
    else:                       # one process per build of the reference (it keeps module state)
        for m in ("-", "island"):
            subprocess.run([sys.executable, os.path.abspath(__file__), m], check=True)
