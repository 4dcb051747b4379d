"""Golden vectors produced by the REFERENCE's own Fortran (tests/golden/
make_golden.py, from oracle/_ref) -- they travel with the repo, the reference does
not.  CPU: the C oracle reproduces them bit for bit.  GPU (-m gpu): so does the
HIP path through the C ABI."""
import os

import numpy as np
import pytest

import util

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = ["BENCHMARK_TINY", "UPWELLING", "SEAMOUNT"]


def _load(config):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(HERE, "golden", f"ref_{config}.npz"))
    st0 = mg.input_state(config)
    assert mg.checksum(st0) == str(g["input_sha256"]), "seeded inputs changed: regenerate the fixtures"
    return g, st0, mg


def _check(st, st0, g, kernel, tol=0.0):
    keys = [k for k in g.files if k.startswith(kernel + "__")]
    assert keys, kernel
    worst = 0.0
    for key in keys:
        _, name, q = key.split("__")
        a = st[name]
        nplane = a.shape[0] * a.shape[1] * (a.shape[2] if a.ndim > 3 else 1)
        got = a.reshape((nplane, -1), order="F")[:, int(q)]
        want = g[key]
        scale = max(float(np.abs(want).max()), 1e-300)
        worst = max(worst, float(np.abs(got - want).max()) / scale)
    assert worst <= tol, (kernel, worst)


def _check_diag(v, report):
    """v = the 12-vector of diag (roms_hip.h); report = what the reference printed for one tile:
    avgke/volume, avgpe/volume, their sum, volume, (Ci,Cj,Ck), Cu, Cv, Cw, maxspeed -- 7 digits."""
    mine = [v[1] / v[0], v[2] / v[0], v[1] / v[0] + v[2] / v[0], v[0], v[9], v[10], v[11], v[6], v[7], v[8], v[3]]
    for q, (a, b) in enumerate(zip(mine, report)):
        if q in (4, 5, 6):
            assert int(a) == int(b), (q, a, b)
        else:
            assert abs(a - b) <= 6e-7 * abs(b), (q, a, b)


@pytest.mark.parametrize("config", CONFIGS)
def test_oracle_reproduces_reference_vectors(config):
    import oracle
    from roms_trunk_mgh_amd import bounds as B
    g, st0, mg = _load(config)
    # tile bounds and barotropic filter weights from the reference's get_tile / set_weights
    mine = st0.b.as_dict()
    for k, v in zip(g["bounds_names"], g["bounds_values"]):
        assert mine[str(k)] == int(v), k
    assert int(g["nfast"]) == st0.p.nfast
    n2 = 2 * st0.p.ndtfast
    assert max(abs(g["weight1"][i] - st0.p.weight1[i]) for i in range(n2)) < 1e-15
    assert max(abs(g["weight2"][i] - st0.p.weight2[i]) for i in range(n2)) < 1e-15
    s = util.step_idx()
    for k in mg.KERNELS + mg.PHYSICS:
        if k == "uv3dmix2" and config == "SEAMOUNT":
            continue
        if k in ("bulk_flux", "lmd_vmix") and not config.startswith("BENCHMARK"):
            continue
        st = st0.copy()
        oracle.Oracle(st).call(k, s)
        _check(st, st0, g, k, tol=1e-13 if k in ("bulk_flux", "lmd_vmix") else 0.0)
    # diagnostics: wvelocity bit for bit, then diag on that state against the reference's printed report
    st = st0.copy()
    o = oracle.Oracle(st)
    o.call("wvelocity", s)
    _check(st, st0, g, "wvelocity", tol=0.0)
    if "diag_report" in g.files:
        _check_diag(o.diag(s), g["diag_report"])


@pytest.mark.gpu
@pytest.mark.parametrize("config", CONFIGS)
def test_hip_reproduces_reference_vectors(config):
    from roms_trunk_mgh_amd import hip
    g, st0, mg = _load(config)
    s = util.step_idx()
    for k in mg.KERNELS + mg.PHYSICS:
        if k == "uv3dmix2" and config == "SEAMOUNT":
            continue
        if k in ("bulk_flux", "lmd_vmix") and not config.startswith("BENCHMARK"):
            continue
        st = st0.copy()
        h = hip.RomsHip(st)
        try:
            h.call(k, s)
            h.to_host()
        finally:
            h.close()
        _check(st, st0, g, k, tol={"bulk_flux": 1e-11, "lmd_vmix": 1e-10}.get(k, 1e-13))
    st = st0.copy()
    h = hip.RomsHip(st)
    try:
        h.call("wvelocity", s)
        d = h.diag(s)
        h.to_host()
    finally:
        h.close()
    _check(st, st0, g, "wvelocity", tol=0.0)
    if "diag_report" in g.files:
        _check_diag(d, g["diag_report"])


def _load_mask(config):
    import importlib.util
    sys_path_add = os.path.join(HERE, "golden")
    import sys
    if sys_path_add not in sys.path:
        sys.path.insert(0, sys_path_add)
    spec = importlib.util.spec_from_file_location("make_golden_mask", os.path.join(HERE, "golden", "make_golden_mask.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from make_golden import checksum
    g = np.load(os.path.join(HERE, "golden", f"ref_mask_{config}.npz"))
    st0 = mg.input_state(config)
    assert checksum(st0) == str(g["input_sha256"]), "seeded inputs changed: regenerate the fixtures"
    return g, st0, mg


@pytest.mark.parametrize("config", ["BENCHMARK_TINY", "UPWELLING"])
def test_oracle_reproduces_masked_reference_vectors(config):
    """MASKING: outputs of the reference built with -DMASKING on a grid with an island and a headland
    (tests/golden/make_golden_mask.py) -- rho_eos, prsgrd32, t3dmix2 (geo / s), uv3dmix2_s and the glue kernels."""
    import oracle
    g, st0, mg = _load_mask(config)
    s = util.step_idx()
    for k in mg.KERNELS:
        st = st0.copy()
        oracle.Oracle(st).call(k, s)
        _check(st, st0, g, k, tol=0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["BENCHMARK_TINY", "UPWELLING"])
def test_hip_reproduces_masked_reference_vectors(config):
    from roms_trunk_mgh_amd import hip
    g, st0, mg = _load_mask(config)
    s = util.step_idx()
    for k in mg.KERNELS:
        st = st0.copy()
        h = hip.RomsHip(st)
        try:
            h.call(k, s)
            h.to_host()
        finally:
            h.close()
        _check(st, st0, g, k, tol=1e-13)


@pytest.mark.parametrize("config,mask", [("UPWELLING", None), ("UPWELLING", "island"), ("BENCHMARK_TINY", None)])
def test_oracle_reproduces_reference_boundary_conditions(config, mask):
    """The boundary rows the reference's zetabc / u2dbc / v2dbc / u3dbc / v3dbc / t3dbc _tile left (90 cases:
    closed, gradient, clamped, Chapman implicit, Flather, radiation; tests/golden/make_golden_bc.py) vs the oracle."""
    import importlib.util
    import sys
    import oracle
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    spec = importlib.util.spec_from_file_location("make_golden_bc", os.path.join(gd, "make_golden_bc.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from make_golden import checksum
    g = np.load(os.path.join(gd, f"ref_bc_{mg.tag(config, mask)}.npz"))
    st0 = mg.input_state(config, mask)
    assert checksum(st0) == str(g["input_sha256"]), "seeded inputs changed: regenerate the fixtures"
    n = 0
    for key, kind, var, st, s, nout, itrc in mg.cases(st0):
        oracle.Oracle(st).bc(kind, s, nout, itrc)
        assert np.array_equal(mg.rows(st, var), g[key]), key
        n += 1
    assert n == 90


@pytest.mark.parametrize("config,mask", [("UPWELLING", None), ("UPWELLING", "island"), ("BENCHMARK_TINY", None)])
def test_oracle_reproduces_reference_first_step_initialisation(config, mask):
    """ini_zeta + ini_fields of the reference (ini_fields.F; tests/golden/make_golden_ini.py) vs the oracle: five
    boundary-condition tables x two index branches, every output bit for bit (arrays or their SHA-256)."""
    import sys
    import oracle
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_ini as mi
    from ref_worker import ini_cases
    g = np.load(os.path.join(gd, f"ref_ini_{mi.tag(config, mask)}.npz"))
    n = 0
    for key, st, s in ini_cases(config, mask):
        o = oracle.Oracle(st)
        o.call("ini_zeta", s)
        o.call("ini_fields", s)
        for name, val in mi.results(st, key).items():
            want = g[key.replace(":", "__") + "__" + name]
            assert (str(val) == str(want)) if name.endswith("_sha256") else np.array_equal(val, want), (key, name)
        n += 1
    assert n == 10


@pytest.mark.gpu
@pytest.mark.parametrize("config,mask", [("UPWELLING", None), ("UPWELLING", "island"), ("BENCHMARK_TINY", None)])
def test_hip_reproduces_reference_first_step_initialisation(config, mask):
    """roms_hip_ini_zeta + roms_hip_ini_fields against the committed outputs of the reference's ini_fields.F, bit for
    bit (only +, *, / and the mask multiplies: no tolerance needed)."""
    import sys
    from roms_trunk_mgh_amd import hip
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_ini as mi
    from ref_worker import ini_cases
    g = np.load(os.path.join(gd, f"ref_ini_{mi.tag(config, mask)}.npz"))
    for key, st, s in ini_cases(config, mask):
        h = hip.RomsHip(st)
        try:
            h.call("ini_zeta", s)
            h.call("ini_fields", s)
            h.to_host()
        finally:
            h.close()
        for name, val in mi.results(st, key).items():
            want = g[key.replace(":", "__") + "__" + name]
            assert (str(val) == str(want)) if name.endswith("_sha256") else np.array_equal(val, want), (key, name)


def _pgf_check(config, backend):
    import sys
    import util
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_pgf as mp
    g = np.load(os.path.join(gd, f"ref_pgf_{config}.npz"))
    s = util.step_idx()
    for variant in mp.VARIANTS:
        st = mp.input_state(config, variant)
        backend(st, s)
        for k, v in mp.results(st, s).items():
            want = g[f"{variant}__{k}"]
            assert (str(v) == str(want)) if k.endswith("_sha256") else np.array_equal(v, want), (variant, k)


@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT"])
def test_oracle_reproduces_reference_prsgrd31(config):
    """prsgrd31_tile of the reference (standard and weighted Jacobian; tests/golden/make_golden_pgf.py) vs the oracle."""
    import oracle
    _pgf_check(config, lambda st, s: oracle.Oracle(st).call("prsgrd", s))


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT"])
def test_hip_reproduces_reference_prsgrd31(config):
    """k_prsgrd31 against the committed outputs of the reference's prsgrd31.h, bit for bit."""
    from roms_trunk_mgh_amd import hip

    def run(st, s):
        h = hip.RomsHip(st)
        try:
            h.call("prsgrd", s)
            h.to_host()
        finally:
            h.close()
    _pgf_check(config, run)


def _dif4_check(config, backend):
    import sys
    import util
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_dif4 as md
    g = np.load(os.path.join(gd, f"ref_dif4_{config}.npz"))
    s = util.step_idx()
    for variant in md.CONFIGS[config]:
        for kernel in ("t3dmix4", "uv3dmix4"):
            st = md.input_state(config, variant)
            backend(st, s, kernel)
            for k, v in md.results(st, s, kernel).items():
                want = g[f"{variant}__{kernel}__{k}"]
                assert (str(v) == str(want)) if k.endswith("_sha256") else np.array_equal(v, want), (variant, kernel, k)


@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT"])
def test_oracle_reproduces_reference_biharmonic_mixing(config):
    """t3dmix4_s / t3dmix4_geo / uv3dmix4_s of the reference (periodic channel, closed and open basins, island grid;
    tests/golden/make_golden_dif4.py) vs the oracle, bit for bit."""
    import oracle
    _dif4_check(config, lambda st, s, kernel: oracle.Oracle(st).call(kernel, s))


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT"])
def test_hip_reproduces_reference_biharmonic_mixing(config):
    """The HIP biharmonic kernels against the committed outputs of the reference's Fortran, bit for bit."""
    from roms_trunk_mgh_amd import hip

    def run(st, s, kernel):
        h = hip.RomsHip(st)
        try:
            h.call(kernel, s)
            h.to_host()
        finally:
            h.close()
    _dif4_check(config, run)


def _iso_check(config, backend):
    import sys
    import util
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_iso as mi
    g = np.load(os.path.join(gd, f"ref_iso_{config}.npz"))
    s = util.step_idx()
    for variant in mi.CONFIGS[config]:
        for kernel in mi.KERNELS:
            st = mi.input_state(config, variant)
            backend(st, s, kernel)
            for k, v in mi.results(st, s, kernel).items():
                want = g[f"{variant}__{kernel}__{k}"]
                assert (str(v) == str(want)) if k.endswith("_sha256") else np.array_equal(v, want), (variant, kernel, k)


@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT"])
def test_oracle_reproduces_reference_isopycnal_mixing(config):
    """t3dmix2_iso / t3dmix4_iso of the reference (tests/golden/make_golden_iso.py) vs the oracle, bit for bit."""
    import oracle
    _iso_check(config, lambda st, s, kernel: oracle.Oracle(st).call(kernel, s))


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT"])
def test_hip_reproduces_reference_isopycnal_mixing(config):
    """k_t3dmix_geo<MODE, ISO> against the committed outputs of the reference's t3dmix2_iso.h / t3dmix4_iso.h."""
    from roms_trunk_mgh_amd import hip

    def run(st, s, kernel):
        h = hip.RomsHip(st)
        try:
            h.call(kernel, s)
            h.to_host()
        finally:
            h.close()
    _iso_check(config, run)


def _kpp_check(mask, backend, tol):
    import sys
    import util
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_kpp as mk
    g = np.load(os.path.join(gd, f"ref_kpp_{mk.tag(mask)}.npz"))
    st = util.kpp_state("BENCHMARK_TINY", mask=mask)
    backend(st, util.step_idx())
    for k, v in mk.results(st).items():
        want = g[k]
        if k.endswith("_sha256"):
            if tol == 0.0:
                assert str(v) == str(want), k
        else:
            scale = max(float(np.abs(want).max()), 1e-300)
            assert float(np.abs(v - want).max()) <= tol * scale, (k, float(np.abs(v - want).max()) / scale)
    hs = st.interior("hsbl")
    assert float(hs.max()) > -1.0 and float(hs.min()) < -100.0          # shallow and deep boundary layers


@pytest.mark.parametrize("mask", [None, "island"])
def test_oracle_reproduces_reference_kpp_on_stratified_state(mask):
    """lmd_vmix of the reference on util.kpp_state (tests/golden/make_golden_kpp.py), with and without MASKING, vs
    the oracle: bit for bit on this host (same libm)."""
    import oracle
    _kpp_check(mask, lambda st, s: oracle.Oracle(st).call("lmd_vmix", s), 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("mask", [None, "island"])
def test_hip_reproduces_reference_kpp_on_stratified_state(mask):
    """k_lmd_vmix against the reference's vectors directly (device pow / exp: 1e-10 of each field's maximum)."""
    from roms_trunk_mgh_amd import hip

    def run(st, s):
        h = hip.RomsHip(st)
        try:
            h.call("lmd_vmix", s)
            h.to_host()
        finally:
            h.close()
    _kpp_check(mask, run, 1e-10)


def _gls_check(config, mask, backend, tol, family="gls"):
    import sys
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_gls as mg
    g = np.load(os.path.join(gd, f"ref_{family}_{mg.tag(config, mask)}.npz"))
    for gset in (mg.SETS if family == "gls" else ["my25"]):
        for kernel in mg.KERNELS:
            st, s = mg.prepare(config, gset, kernel, mask)
            st0 = st.copy()
            backend(st, kernel, s)
            for k, v in mg.results(st, f"{gset}/{kernel}").items():
                want = g[k]
                if k.endswith("_sha256"):
                    if tol == 0.0:
                        assert str(v) == str(want), k
                else:
                    scale = max(float(np.abs(want).max()), 1e-300)
                    assert float(np.abs(v - want).max()) <= tol * scale, (k, float(np.abs(v - want).max()) / scale)
            changed = ["tke", "gls"] if kernel == "gls_prestep" else [n for n in mg.NAMES if family == "gls" or n != "Akp"]
            assert all(not np.array_equal(st[n], st0[n]) for n in changed), (gset, kernel)


GLS_CASES = [("UPWELLING", None), ("UPWELLING", "island"), ("BENCHMARK_TINY", None)]


@pytest.mark.parametrize("config,mask", GLS_CASES)
def test_oracle_reproduces_reference_gls(config, mask):
    """gls_prestep / gls_corstep of the reference's GLS builds (tests/golden/make_golden_gls.py: Kantha-Clayson with
    N2S2_HORAVG and RI_SPLINES, with and without MASKING; Canuto A with the plain shear) vs the oracle: bit for bit on
    this host (same libm)."""
    import oracle
    _gls_check(config, mask, lambda st, k, s: oracle.Oracle(st).call(k, s), 0.0)


@pytest.mark.parametrize("config,mask", GLS_CASES)
def test_oracle_reproduces_reference_my25(config, mask):
    """my25_prestep / my25_corstep of the reference's MY25_MIXING builds (Kantha-Clayson with N2S2_HORAVG and RI_SPLINES,
    with and without MASKING; the plain closure with the Galperin functions and the plain shear) vs the oracle
    (gls_mixing = 2): bit for bit on this host."""
    import oracle
    _gls_check(config, mask, lambda st, k, s: oracle.Oracle(st).call(k, s), 0.0, family="my25")


@pytest.mark.gpu
@pytest.mark.parametrize("config,mask", GLS_CASES)
def test_hip_reproduces_reference_gls(config, mask):
    """k_gls_prestep / k_gls_corstep against the reference's vectors directly (device pow: 1e-10 of each field's
    maximum)."""
    from roms_trunk_mgh_amd import hip

    def run(st, kernel, s):
        h = hip.RomsHip(st)
        try:
            h.call(kernel, s)
            h.to_host()
        finally:
            h.close()
    _gls_check(config, mask, run, 1e-10)
    _gls_check(config, mask, run, 1e-10, family="my25")          # MY25_MIXING: the my25_prestep / my25_corstep vectors


@pytest.mark.parametrize("config,mask", [("UPWELLING", None), ("UPWELLING", "island"), ("BENCHMARK_TINY", None)])
def test_oracle_reproduces_reference_boundary_conditions_on_a_basin(config, mask):
    """The boundary lines (columns, rows, corners) the reference's six routines left on a grid without a periodic
    direction, every condition on all four edges at once (tests/golden/make_golden_bc4.py) vs the oracle."""
    import sys
    import oracle
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_bc4 as mb
    from ref_worker import basin_state, basin_cases
    g = np.load(os.path.join(gd, f"ref_bc4_{mb.tag(config, mask)}.npz"))
    n = 0
    for key, kind, var, st, s, nout, itrc in basin_cases(basin_state(config, mask)):
        oracle.Oracle(st).bc(kind, s, nout, itrc)
        cols, rows, sha = mb.lines(st, var)
        k = key.replace(":", "__")
        assert np.array_equal(cols, g[k + "__cols"]) and np.array_equal(rows, g[k + "__rows"]), key
        assert sha == str(g[k + "__sha256"]), key
        n += 1
    assert n == 90


@pytest.mark.parametrize("mask", [None, "island"])
def test_oracle_reproduces_reference_mpdata_adiff(mask):
    """mpdata_adiff_tile: the committed outputs of the reference's Fortran (three levels stored,
    all elements through a SHA-256) vs the C oracle on the same deterministic inputs; the second fixture is the
    MASKING build on the island grid."""
    import hashlib
    import util
    path = os.path.join(HERE, "golden", f"ref_mpdata_BENCHMARK_TINY{'_MASK' if mask else ''}.npz")
    g = np.load(path)

    def sha(*arrays):
        h = hashlib.sha256()
        for a in arrays:
            h.update(np.ascontiguousarray(a + 0.0).tobytes())
        return h.hexdigest()

    st = util.prepared_state("BENCHMARK_TINY", overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"}, mask=mask)
    oHz, Ta0, t3 = util.mpdata_private_arrays(st)
    assert sha(oHz, Ta0, t3) == str(g["input_sha256"]), "fixture inputs changed: regenerate with make_golden_mpdata.py"
    Ta, Ua, Va, Wa = util.oracle_mpdata_adiff(st, oHz, Ta0, t3)
    ks = [int(k) for k in g["levels"]]
    assert np.array_equal(Ua[:, :, ks], g["Ua"]) and np.array_equal(Va[:, :, ks], g["Va"])
    assert np.array_equal(Wa[:, :, [k + 1 for k in ks]], g["Wa"]) and np.array_equal(Ta[:, :, ks], g["Ta"])
    assert sha(Ta, Ua, Va, Wa) == str(g["output_sha256"])


@pytest.mark.parametrize("config", ["BENCHMARK_TINY", "UPWELLING", "SEAMOUNT"])
def test_analytic_setup_reproduces_reference_fields(config):
    """ana.py against the fields the reference's ana_grid + metrics, ana_initial and forcing routines produced
    (committed by make_golden.py): the check of tests/test_ref_pinning.py where the reference is absent."""
    from roms_trunk_mgh_amd import ana
    g = np.load(os.path.join(HERE, "golden", f"ref_{config}.npz"))
    st = ana.make_tile(config, perturb=0.0)
    b = st.b
    reg = (slice(0, b.Lm + b.NghostPoints - b.LBi + 1), slice(0 - b.LBj, b.Mm + 1 - b.LBj + 1))
    if "srflux_field" in g.files:
        import oracle
        from roms_trunk_mgh_amd import main3d
        yd, hr = g["srflux_clock"]
        assert main3d.host_clock(0.3) == (float(yd), float(hr))
        so = st.copy()
        oracle.Oracle(so).ana_srflux(yd, hr)
        assert np.array_equal(so.interior("srflx"), g["srflux_field"][st.I(b.Istr, b.Iend), st.J(b.Jstr, b.Jend)])
    keys = [k for k in g.files if k.startswith("ana__") and k != "ana__T0"]
    assert len(keys) >= 23
    for key in keys:
        name = key[5:]
        want, got = g[key][reg], st[name][reg]
        tol = 4e-16 if name == "h" else 0.0
        assert float(np.abs(got - want).max()) <= tol * float(np.abs(want).max()), name
    own = (st.I(b.Istr, b.Iend), st.J(b.Jstr, b.Jend))
    want, got = g["ana__T0"][own], st["t"][:, :, :, 0, 0][own]
    assert float(np.abs(got - want).max()) <= 4e-16 * float(np.abs(want).max())


def _wet_mod():
    import sys
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_wet as mw
    return mw


def _wet_check(config, backend, tol, only=None):
    mw = _wet_mod()
    g = np.load(os.path.join(HERE, "golden", f"ref_wet_{config}_MASK.npz"))
    seen = 0
    for label, st, s, op in mw.kernel_cases(config):
        if only is not None and not only(label):
            continue
        st0 = st.copy()
        backend(st, s, op)
        res = mw.results(st, st0, label)
        want_keys = {k for k in g.files if k.startswith(label + "/")}
        assert set(res) == want_keys, (label, sorted(set(res) ^ want_keys))
        for k, v in res.items():
            want = g[k]
            if k.endswith("_sha256"):
                if tol == 0.0:
                    assert str(v) == str(want), k
            else:
                scale = max(float(np.abs(want).max()), 1e-300)
                assert float(np.abs(v - want).max()) <= tol * scale, (k, float(np.abs(v - want).max()) / scale)
        seen += 1
    assert seen


@pytest.mark.parametrize("config", ["UPWELLING", "BENCHMARK_TINY"])
def test_oracle_reproduces_reference_wet_dry_kernels(config):
    """WET_DRY: set_depth, prsgrd32 / prsgrd31, t3dmix2 (s / geo), uv3dmix2, t3dmix4, uv3dmix4, bulk_flux, ini_zeta +
    ini_fields of the reference built with -DWET_DRY (tests/golden/make_golden_wet.py) vs the oracle, bit for bit."""
    import oracle

    def run(st, s, op):
        o = oracle.Oracle(st)
        if op == "ini":
            o.call("ini_zeta", s)
            o.call("ini_fields", s)
        else:
            o.call(op.split(":")[-1], s)
    _wet_check(config, run, 0.0)


def test_oracle_reproduces_reference_wet_dry_boundary_conditions():
    """WET_DRY blocks of zetabc.F (:733-827), u2dbc_im.F (:331, :679, :1176-1293), v2dbc_im.F (:333, :682, :1169-1287),
    u3dbc_im.F, v3dbc_im.F on a basin with shallow stretches along the edges: SHA-256 of the reference's outputs."""
    import oracle
    mw = _wet_mod()
    g = np.load(os.path.join(HERE, "golden", "ref_wet_bc_UPWELLING_MASK.npz"))
    n = 0
    for label, kind, var, st, s, nout, itrc in mw.bc_cases("UPWELLING"):
        oracle.Oracle(st).bc(kind, s, nout, itrc)
        assert mw.sha(st[var]) == str(g[f"{label}/{var}_sha256"]), label
        n += 1
    assert n == len(g.files)


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["UPWELLING", "BENCHMARK_TINY"])
def test_hip_reproduces_reference_wet_dry_kernels(config):
    """The same vectors against the HIP path (bit for bit but for bulk_flux, whose exp / log come from another libm)."""
    from roms_trunk_mgh_amd import hip

    def run(st, s, op):
        h = hip.RomsHip(st)
        try:
            if op == "ini":
                h.call("ini_zeta", s)
                h.call("ini_fields", s)
            else:
                h.call(op.split(":")[-1], s)
            h.to_host()
        finally:
            h.close()
    _wet_check(config, run, 1e-12)


def _atm_check(backend, tol):
    import sys
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_atm as ma
    g = np.load(os.path.join(gd, "ref_atm_UPWELLING.npz"))
    for variant in ma.VARIANTS:
        st, s = ma.prepare(variant)
        st_off = st.copy()
        st_off.p = type(st.p).from_buffer_copy(st.p)
        backend(st, s)
        for k, v in ma.results(st, variant).items():
            want = g[k]
            if k.endswith("_sha256"):
                if tol == 0.0:
                    assert str(v) == str(want), k
            else:
                scale = max(float(np.abs(want).max()), 1e-300)
                assert float(np.abs(v - want).max()) <= tol * scale, k
        # the air-pressure term matters
        st_off.p.atm_press = 0
        backend(st_off, s)
        assert not np.array_equal(st_off["ru"], st["ru"]) and not np.array_equal(st_off["rv"], st["rv"])


def test_oracle_reproduces_reference_atm_press():
    """ATM_PRESS: prsgrd32 / prsgrd31 / prsgrd40 of the reference built with the option (make_golden_atm.py)."""
    import oracle
    _atm_check(lambda st, s: oracle.Oracle(st).call("prsgrd", s), 0.0)


@pytest.mark.gpu
def test_hip_reproduces_reference_atm_press():
    from roms_trunk_mgh_amd import hip

    def run(st, s):
        h = hip.RomsHip(st)
        try:
            h.call("prsgrd", s)
            h.to_host()
        finally:
            h.close()
    _atm_check(run, 1e-13)


def _stab_check(backend, tol):
    import sys
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_stab as ms
    g = np.load(os.path.join(gd, "ref_stab.npz"))
    for variant in ms.VARIANTS:
        for kernel in ms.KERNELS:
            st, s = ms.prepare(variant)
            st_off = st.copy()
            st_off.p = type(st.p).from_buffer_copy(st.p)
            backend(st, s, kernel)
            for k, v in ms.results(st, variant, kernel).items():
                want = g[k]
                if k.endswith("_sha256"):
                    if tol == 0.0:
                        assert str(v) == str(want), k
                else:
                    scale = max(float(np.abs(want).max()), 1e-300)
                    assert float(np.abs(v - want).max()) <= tol * scale, k
            # the option matters (the 1/4 t(nstp) part; the stratification bound)
            st_off.p.ts_mix_stability = 0
            st_off.p.ts_mix_min_strat = 0
            backend(st_off, s, kernel)
            assert not np.array_equal(st_off["t"], st["t"]), (variant, kernel)


def test_oracle_reproduces_reference_ts_mix_stability():
    """TS_MIX_STABILITY: t3dmix2 / t3dmix4 along s-surfaces, geopotentials and isopycnals of the reference built with
    the option, and TS_MIX_MIN_STRAT on the isopycnal operators (make_golden_stab.py), whole arrays by SHA-256."""
    import oracle
    _stab_check(lambda st, s, k: oracle.Oracle(st).call(k, s), 0.0)


@pytest.mark.gpu
def test_hip_reproduces_reference_ts_mix_stability():
    from roms_trunk_mgh_amd import hip

    def run(st, s, k):
        h = hip.RomsHip(st)
        try:
            h.call(k, s)
            h.to_host()
        finally:
            h.close()
    _stab_check(run, 1e-13)


def _geouv_check(case, backend, tol):
    import sys
    gd = os.path.join(HERE, "golden")
    if gd not in sys.path:
        sys.path.insert(0, gd)
    import make_golden_geouv as mg
    g = np.load(os.path.join(gd, "ref_geouv.npz"))
    st, s = mg.prepare(case)
    st0 = st.copy()
    backend(st, s)
    for k, v in mg.results(st, s, case).items():
        want = g[k]
        if k.endswith("_sha256"):
            if tol == 0.0:
                assert str(v) == str(want), k
        else:
            scale = max(float(np.abs(want).max()), 1e-300)
            assert float(np.abs(v - want).max()) <= tol * scale, (k, float(np.abs(v - want).max()) / scale)
    assert all(not np.array_equal(st[n], st0[n]) for n in mg.NAMES), case


GEOUV_CASES = ["channel", "seamount", "island", "island_wet"]


@pytest.mark.parametrize("case", GEOUV_CASES)
def test_oracle_reproduces_reference_uv3dmix2_geo(case):
    """uv3dmix2_geo.h (UV_VIS2 with MIX_GEO_UV, uv_vis2 = 2) of the reference built with the option
    (tests/golden/make_golden_geouv.py: channel, seamount, island grid, island grid with WET_DRY) vs the oracle: bit for bit."""
    import oracle
    _geouv_check(case, lambda st, s: oracle.Oracle(st).call("uv3dmix2", s), 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GEOUV_CASES)
def test_hip_reproduces_reference_uv3dmix2_geo(case):
    """k_uv3dmix2_geo against the reference's vectors directly (same operations in the same order: 1e-13)."""
    from roms_trunk_mgh_amd import hip

    def run(st, s):
        h = hip.RomsHip(st)
        try:
            h.call("uv3dmix2", s)
            h.to_host()
        finally:
            h.close()
    _geouv_check(case, run, 1e-13)
