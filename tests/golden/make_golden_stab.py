"""Generates tests/golden/ref_stab.npz from the REFERENCE's own t3dmix2 / t3dmix4 built with -DTS_MIX_STABILITY (and,
the minstrat_* variants, with -DTS_MIX_MIN_STRAT)
(oracle/_ref/UPWELLING_STAB_DIF4, SEAMOUNT_STAB_DIF4, UPWELLING_STAB_ISO, SEAMOUNT_STAB_ISO; oracle/build_ref.sh):
3/4 t(nrhs) + 1/4 t(nstp) in every tracer difference of t3dmix2_s.h / t3dmix2_geo.h / t3dmix2_iso.h and of the first
operator of t3dmix4_s.h / t3dmix4_geo.h / t3dmix4_iso.h, on the states of tests/util.prepared_state and
tests/ref_worker.iso_state at nrhs = 3, nstp = 1, nnew = 2.  Stored per variant and operator: every second point of
three levels of t(nnew) of both tracers and a SHA-256 of the whole array.  One child process per variant (one
reference library each).  Run in this container:

    python tests/golden/make_golden_stab.py
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
# variant -> (configuration, isopycnal, option)
VARIANTS = {"s": ("UPWELLING", False, "ts_mix_stability"), "geo": ("SEAMOUNT", False, "ts_mix_stability"),
            "iso_upw": ("UPWELLING", True, "ts_mix_stability"), "iso_sea": ("SEAMOUNT", True, "ts_mix_stability"),
            # TS_MIX_MIN_STRAT (oracle/_ref/UPWELLING_MINSTRAT_ISO, SEAMOUNT_MINSTRAT_ISO): the slope scale of the
            # isopycnal operators bounded by strat_min * dz (t3dmix2_iso.h:313-316, t3dmix4_iso.h:361-364, :679-682)
            "minstrat_upw": ("UPWELLING", True, "ts_mix_min_strat"), "minstrat_sea": ("SEAMOUNT", True, "ts_mix_min_strat")}
KERNELS = ("t3dmix2", "t3dmix4")


def prepare(variant):
    import ref_worker
    import util
    config, iso, opt = VARIANTS[variant]
    if iso:
        st = ref_worker.iso_state(config, extra={opt: 1})
    else:
        st = util.prepared_state(config, overrides=dict(ref_worker.DIF4, tnu2=300.0, **{opt: 1}))
    assert getattr(st.p, opt) == 1
    return st, util.step_idx(nstp=1, nnew=2, nrhs=3)


def results(st, variant, kernel):
    N = st.b.N
    a = st["t"][:, :, :, 1, :]                       # t(nnew = 2)
    return {f"{variant}/{kernel}/t_levels": a[::2, ::2][:, :, [0, N // 2, N - 1]].copy(),
            f"{variant}/{kernel}/t_sha256": np.array(hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest())}


def child(variant):
    from oracle import ref
    out = {}
    for k in KERNELS:
        st, s = prepare(variant)
        ref.Ref(st).call(k, s)
        out.update(results(st, variant, k))
    np.savez_compressed(os.path.join(HERE, f"_stab_{variant}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        out = {}
        for v in VARIANTS:
            subprocess.run([sys.executable, os.path.abspath(__file__), v], check=True)
            part = os.path.join(HERE, f"_stab_{v}.npz")
            out.update(dict(np.load(part)))
            os.remove(part)
        np.savez_compressed(os.path.join(HERE, "ref_stab.npz"), **out)
        print(os.path.getsize(os.path.join(HERE, "ref_stab.npz")) // 1024, "KiB")
