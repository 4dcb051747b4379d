"""Generates tests/golden/ref_mask_<CONFIG>.npz from the REFERENCE's own Fortran built with -DMASKING
(oracle/_ref/<APP>_MASK/libref.so, oracle/build_ref.sh): the kernels of the path that the reference can run
here, on a grid with an island and a headland (roms_trunk_mgh_amd.ana.island_mask).  Run in this container:

    python tests/golden/make_golden_mask.py

Inputs: tests/util.prepared_state(config, mask="island") with the detuning of make_golden.py (a checksum of
all input arrays is stored); outputs: every horizontal-plane stack a kernel changed."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

CONFIGS = ["BENCHMARK_TINY", "UPWELLING"]
KERNELS = ["set_depth", "set_massflux", "set_zeta", "rho_eos", "prsgrd", "t3dmix2", "uv3dmix2"]


def input_state(config):
    import util
    from make_golden import OVERRIDES
    st = util.prepared_state(config, overrides=OVERRIDES[config], mask="island")
    st["Zt_avg1"] *= 1.3
    st["u"] *= 1.1
    return st


def child(config):
    import util
    from make_golden import checksum
    from oracle import ref
    from roms_trunk_mgh_amd import abi
    st0 = input_state(config)
    assert st0.p.masking == 1
    out = {"input_sha256": np.array(checksum(st0))}
    s = util.step_idx()
    for k in KERNELS:
        st = st0.copy()
        ref.Ref(st).call(k, s)
        for name, kind, _ in abi.FIELDS:
            a, a0 = st[name], st0[name]
            if np.array_equal(a, a0):
                continue
            nplane = a.shape[0] * a.shape[1] * (a.shape[2] if a.ndim > 3 else 1)
            fa = a.reshape((nplane, -1), order="F")
            f0 = a0.reshape((nplane, -1), order="F")
            for q in range(fa.shape[1]):
                if not np.array_equal(fa[:, q], f0[:, q]):
                    out[f"{k}__{name}__{q}"] = fa[:, q].copy()
    np.savez_compressed(os.path.join(HERE, f"ref_mask_{config}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for c in CONFIGS:
            subprocess.run([sys.executable, os.path.abspath(__file__), c], check=True)
            print(c, os.path.getsize(os.path.join(HERE, f"ref_mask_{c}.npz")) // 1024, "KiB")
