"""Generates tests/golden/ref_wet_<CONFIG>_MASK.npz from the REFERENCE's own Fortran built with -DWET_DRY
(oracle/_ref/UPWELLING_MASK_WET[_DIF4|_PG31], BENCHMARK_MASK_WET; oracle/build_ref.sh): the WET_DRY blocks of
set_depth.F, prsgrd32.h / prsgrd31.h, t3dmix2_s.h / t3dmix2_geo.h, uv3dmix2_s.h, t3dmix4_s.h, uv3dmix4_s.h,
bulk_flux.F, mpdata_adiff.F, ini_fields.F and the six boundary-condition files, on the WET_DRY state of
tests/util.prepared_state(wet=True) -- an island grid with synthetic wet/dry masks holding every value
wetdry_mask_tile can produce.  (wetdry.F itself cannot be built here: it reaches mod_sources -> mod_netcdf.)
Stored per case: every second point of three levels of each changed field and a SHA-256 of the whole arrays; for
the boundary-condition cases the SHA-256 only.  Run in this container:

    python tests/golden/make_golden_wet.py
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

CONFIGS = ["UPWELLING", "BENCHMARK_TINY"]
DIF4 = {"ts_dif4": 1, "uv_vis4": 1, "tnu4": 2.0e7, "visc4": 4.0e7}
BC_CODES = {"zetabc": ("zeta", ["Clo", "Cha", "Rad"]), "u2dbc": ("ubar", ["Clo", "Fla", "Shc", "Red"]),
            "v2dbc": ("vbar", ["Clo", "Fla", "Shc", "Red"]), "u3dbc": ("u", ["Clo", "Gra", "Rad"]),
            "v3dbc": ("v", ["Clo", "Gra", "Rad"])}


def levels(n):
    return sorted({0, n // 2, n - 1})


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a + 0.0).tobytes()).hexdigest()


def kernel_cases(config):
    """(label, state, step indices, op): op = a kernel name of the common call interface, "physics:bulk_flux", or
    "ini" (ini_zeta then ini_fields)."""
    import util
    ov = {"tnu2": 300.0, "visc2": 800.0}
    st = util.prepared_state(config, overrides=ov, mask="island", wet=True)
    st["Zt_avg1"] *= 1.3
    st["u"] *= 1.1
    st["h"][7 - st.b.LBi, 9 - st.b.LBj] = 0.0            # set_depth.F:168-172
    s = util.step_idx()
    for k in ("set_depth", "prsgrd", "t3dmix2", "uv3dmix2"):
        yield k, st.copy(), s, k
    if config == "UPWELLING":
        st4 = util.prepared_state(config, overrides=dict(DIF4), mask="island", wet=True)
        for k in ("t3dmix4", "uv3dmix4"):
            yield k, st4.copy(), s, k
        st31 = util.prepared_state(config, overrides=ov, mask="island", wet=True)
        st31.p.pgf = 1
        yield "prsgrd31", st31, s, "prsgrd"
        import ref_worker
        util.WET = True
        try:
            for key, sti, si in ref_worker.ini_cases(config, "island"):
                if key in ("closed:1", "gradient:2", "cha_fla_rad:1"):
                    yield "ini/" + key, sti, si, "ini"
        finally:
            util.WET = False
    else:
        stb = util.prepared_state(config, overrides=ov, mask="island", wet=True)
        stb["Vwind"] += 0.3 * stb["Uwind"] - 2.0
        stb["rain"] += 2.0e-5
        yield "bulk_flux", stb, s, "physics:bulk_flux"


def bc_cases(config):
    """(label, kind, state, step indices, nout, itrc) on a basin with shallow stretches along the edges."""
    import ref_worker
    import util
    util.WET = True
    try:
        st0 = ref_worker.basin_state(config, "island")
        for key, kind, var, st, s, nout, itrc in ref_worker.basin_cases(st0):
            k, code, q = key.split(":")
            if k in BC_CODES and code in BC_CODES[k][1] and q in ("0", "2"):
                yield "bc/" + key, kind, var, st, s, nout, itrc
    finally:
        util.WET = False


def results(st, st0, label):
    from roms_trunk_mgh_amd import abi
    out = {}
    for name, _, _ in abi.FIELDS:
        a, a0 = st[name], st0[name]
        if np.array_equal(a, a0):
            continue
        sub = a[::2, ::2]
        if a.ndim >= 3 and a.shape[2] > 3:
            sub = sub[:, :, levels(a.shape[2])]
        out[f"{label}/{name}_levels"] = sub.copy()
        out[f"{label}/{name}_sha256"] = np.array(sha(a))
    return out


def run_ref(st, s, op):
    from oracle import ref
    r = ref.Ref(st)
    if op.startswith("physics:"):
        r.physics(op.split(":")[1], s)
    elif op == "ini":
        r.bc("ini_zeta", s, 0, 0)
        r.bc("ini_fields", s, 0, 0)
    else:
        r.call(op, s)


def child(config, what):
    """One process per set of bounds (the reference allocates its module arrays once): the channel cases, the basin
    cases of the boundary conditions."""
    from oracle import ref
    out = {}
    if what == "kernels":
        for label, st, s, op in kernel_cases(config):
            st0 = st.copy()
            run_ref(st, s, op)
            res = results(st, st0, label)
            assert res, label
            out.update(res)
        np.savez_compressed(os.path.join(HERE, f"ref_wet_{config}_MASK.npz"), **out)
    else:
        for label, kind, var, st, s, nout, itrc in bc_cases(config):
            st0 = st.copy()
            ref.Ref(st).bc(kind, s, nout, itrc)
            assert not np.array_equal(st[var], st0[var]), label
            out[f"{label}/{var}_sha256"] = np.array(sha(st[var]))
        np.savez_compressed(os.path.join(HERE, f"ref_wet_bc_{config}_MASK.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], sys.argv[2])
    else:
        for c, w in [(c, "kernels") for c in CONFIGS] + [("UPWELLING", "bc")]:
            subprocess.run([sys.executable, os.path.abspath(__file__), c, w], check=True)
        for f in sorted(os.listdir(HERE)):
            if f.startswith("ref_wet_"):
                print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
