"""Generates tests/golden/ref_iso_<CONFIG>.npz from the REFERENCE's own isopycnal tracer mixing -- t3dmix2_iso_tile and
t3dmix4_iso_tile, oracle/_ref/<APP>_ISO (+ UPWELLING_MASK_ISO) built by oracle/build_ref.sh from the application's
options with MIX_ISO_TS as the tracer mixing choice and TS_DIF4 added.  Inputs: tests/ref_worker.iso_state (potential
density from rho_eos, a weakly and a strongly stratified band so that both branches of MAX(drho, eps) are taken).
Variants: periodic channel, closed basin, basin with "gradient" tracer edges, island grid (UPWELLING).  Stored: t(nnew)
at three levels in full and a SHA-256 of all levels (the comparison is bit for bit).  Run in this container:

    python tests/golden/make_golden_iso.py
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
KERNELS = ("t3dmix2", "t3dmix4")


def input_state(config, variant):
    import ref_worker
    return ref_worker.iso_state(config, basin=variant if variant in ("closed", "open") else None,
                                mask="island" if variant == "mask" else None)


def results(st, s, kernel):
    N = st.b.N
    a = st["t"][:, :, :, s.nnew - 1, :]
    return {"t_levels": a[:, :, [0, N // 2, N - 1], :].copy(),
            "t_sha256": np.array(hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest())}


def child(config, variant, path):
    """One process per variant: the reference keeps one set of bounds (and one build) per process."""
    import util
    from oracle import ref
    out = {}
    s = util.step_idx()
    for kernel in KERNELS:
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
            np.savez_compressed(os.path.join(HERE, f"ref_iso_{c}.npz"), **merged)
            print(c, os.path.getsize(os.path.join(HERE, f"ref_iso_{c}.npz")) // 1024, "KiB")
