"""Generates tests/golden/ref_gls_<CONFIG>[_MASK].npz from the REFERENCE's own gls_prestep / gls_corstep
(gls_prestep.F, gls_corstep.F, tkebc_im.F; oracle/_ref/UPWELLING_GLS, UPWELLING_MASK_GLS and BENCHMARK_GLS built by
oracle/build_ref.sh: KANTHA_CLAYSON + N2S2_HORAVG + RI_SPLINES, and CANUTO_A with the plain shear) on the state of
tests/util.gls_state, for the k-epsilon and the k-kl (Mellor-Yamada 2.5: the wall function) parameter sets, both
kernels, iic = 5.  Stored per case: tke / gls (three time levels) and, for gls_corstep, Akv, Akt, Akk, Akp, Lscale at
every second point of three W-levels, and a SHA-256 of the whole arrays.  Run in this container:

    python tests/golden/make_golden_gls.py
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

CASES = [("UPWELLING", None), ("UPWELLING", "island"), ("BENCHMARK_TINY", None)]
SETS = ["k-epsilon", "k-kl"]
KERNELS = ["gls_prestep", "gls_corstep"]
NAMES = ["tke", "gls", "Akv", "Akt", "Akk", "Akp", "Lscale"]


def tag(config, mask):
    return config + ("_MASK" if mask else "")


def levels(N):
    return sorted({0, N // 2, N - 1})


def prepare(config, gset, kernel, mask):
    """The input state of a case and its step indices."""
    import util
    st = util.gls_state(config, gls=gset, mask=mask)
    s = util.step_idx(iic=5)
    if kernel == "gls_corstep":                 # as gls_prestep leaves the nnew level: Hz-weighted
        hzw = np.zeros_like(st["Akv"])
        hzw[:, :, 1:-1] = 0.5 * (st["Hz"][:, :, :-1] + st["Hz"][:, :, 1:])
        hzw[:, :, 0] = hzw[:, :, 1]
        hzw[:, :, -1] = hzw[:, :, -2]
        for n in ("tke", "gls"):
            st[n][:, :, :, s.nnew - 1] = hzw * st[n][:, :, :, s.nstp - 1]
    return st, s


def results(st, prefix):
    """Every second point of three W-levels in full (what the HIP path is compared with) and a SHA-256 of the whole
    array (what the oracle must reproduce bit for bit); gls_prestep changes tke and gls only."""
    out = {}
    lv = levels(st.b.N)
    for name in (NAMES if "corstep" in prefix else NAMES[:2]):
        a = st[name]
        out[f"{prefix}/{name}_levels"] = a[::2, ::2][:, :, lv].copy()
        out[f"{prefix}/{name}_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest())
    return out


def child(config, mask, family="gls"):
    """family "my25": the MY25_MIXING builds (oracle/_ref/UPWELLING_MY25, UPWELLING_MASK_MY25: KANTHA_CLAYSON +
    N2S2_HORAVG + RI_SPLINES; BENCHMARK_MY25: the plain closure) -- my25_prestep.F, my25_corstep.F -> ref_my25_*.npz"""
    from oracle import ref
    out = {}
    for gset in (SETS if family == "gls" else ["my25"]):
        for kernel in KERNELS:
            st, s = prepare(config, gset, kernel, mask)
            ref.Ref(st).gls(kernel, s)
            out.update(results(st, f"{gset}/{kernel}"))
    np.savez_compressed(os.path.join(HERE, f"ref_{family}_{tag(config, mask)}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], None if sys.argv[2] == "-" else sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "gls")
    else:
        for fam in ("gls", "my25"):
            for c, m in CASES:
                subprocess.run([sys.executable, os.path.abspath(__file__), c, m or "-", fam], check=True)
                print(fam, tag(c, m), os.path.getsize(os.path.join(HERE, f"ref_{fam}_{tag(c, m)}.npz")) // 1024, "KiB")
