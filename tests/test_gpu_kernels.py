"""-m gpu: per-kernel parity of the HIP path (through the C ABI) against the CPU
oracle on the same seeded inputs.  Tolerance: 1e-12 relative to the field's
max-norm per call (the north_star bound is 1e-10 relative RMS after 100 steps;
with identical operation order and no FMA contraction the observed difference
is 0 or a few ulp)."""
import os

import numpy as np
import pytest

import util
from roms_trunk_mgh_amd import hip

pytestmark = pytest.mark.gpu
TOL = 1e-12

CONFIGS = ["BENCHMARK_TINY", "UPWELLING", "SEAMOUNT"]


def _run_pair(config, kernel, s, prep=None, NT=None, overrides=None):
    import oracle
    st0 = util.prepared_state(config, NT=NT, overrides=overrides)
    if prep:
        prep(st0)
    st_o, st_h = st0.copy(), st0.copy()
    oracle.Oracle(st_o).call(kernel, s)
    h = hip.RomsHip(st_h)
    try:
        h.call(kernel, s)
        h.to_host()
    finally:
        h.close()
    return st_h, st_o, st0


def _detune(st):
    """make every glue kernel's inputs inconsistent with its current outputs"""
    st["Zt_avg1"] *= 1.3
    st["u"] *= 1.1
    st["v"] *= 0.9
    st["Huon"] *= 1.05
    st["Hvom"] *= 0.95


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("kernel", ["set_depth", "set_massflux", "omega", "set_zeta"])
def test_glue_kernels(config, kernel):
    st_h, st_o, st0 = _run_pair(config, kernel, util.step_idx(), prep=_detune)
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    changed = util.compare_states(st_o, st0)
    assert changed, "kernel did not modify anything: test is vacuous"


@pytest.mark.parametrize("config", CONFIGS)
def test_step3d_t(config):
    st_h, st_o, st0 = _run_pair(config, "step3d_t", util.step_idx(), prep=util.hz_weighted_tnew)
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    assert util.max_rel_diff(st_o["t"], st0["t"]) > 1e-6


@pytest.mark.parametrize("hadv,vadv", [("C2", "C2"), ("C4", "C4"), ("A4", "A4"), ("U3", "SPLINES"),
                                       ("A4", "SPLINES"), ("U3", "C4")])
def test_step3d_t_schemes(hadv, vadv):
    ov = {"Hadv": hadv, "Vadv": vadv}
    st_h, st_o, st0 = _run_pair("UPWELLING", "step3d_t", util.step_idx(), prep=util.hz_weighted_tnew,
                                overrides=ov)
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs


@pytest.mark.parametrize("config", CONFIGS)
def test_prsgrd(config):
    st_h, st_o, st0 = _run_pair(config, "prsgrd", util.step_idx())
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    assert util.max_rel_diff(st_o["ru"], st0["ru"]) > 1e-6


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("pgf", ["STANDARD", "WJ_GRADP", "PJ_GRADP"])
def test_prsgrd31(config, pgf):
    """prsgrd31.h (standard / weighted density Jacobian, prsgrd.F:24-25) or prsgrd40.h (PJ_GRADP) instead of prsgrd32.h; N = 40 as well."""
    for ov in ({"pgf": pgf}, {"pgf": pgf, "N": 40}):
        st_h, st_o, st0 = _run_pair(config, "prsgrd", util.step_idx(), overrides=ov)
        diffs = util.compare_states(st_h, st_o)
        assert all(v <= TOL for v in diffs.values()), diffs
        assert util.max_rel_diff(st_o["ru"], st0["ru"]) > 1e-6 and util.max_rel_diff(st_o["rv"], st0["rv"]) > 1e-6


@pytest.mark.parametrize("config", CONFIGS)
def test_rho_eos(config):
    st_h, st_o, st0 = _run_pair(config, "rho_eos", util.step_idx())
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    assert util.max_rel_diff(st_o["rho"], st0["rho"]) > 1e-6


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("iic", [1, 2, 5])
def test_pre_step3d(config, iic):
    st_h, st_o, st0 = _run_pair(config, "pre_step3d", util.step_idx(iic=iic))
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    assert util.max_rel_diff(st_o["t"], st0["t"]) > 1e-6
    assert util.max_rel_diff(st_o["u"], st0["u"]) > 1e-6


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("kernel", ["t3dmix2", "uv3dmix2", "rhs3d_tile"])
def test_rhs_pieces(config, kernel):
    ov = {"tnu2": 300.0, "visc2": 800.0}      # make the mixing terms non-trivial everywhere
    st_h, st_o, st0 = _run_pair(config, kernel, util.step_idx(), overrides=ov)
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    assert util.compare_states(st_o, st0), "kernel did not modify anything"


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("iic", [1, 4])
def test_rhs3d_driver(config, iic):
    st_h, st_o, st0 = _run_pair(config, "rhs3d", util.step_idx(iic=iic))
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("iic", [1, 2, 5])
def test_step3d_uv(config, iic):
    st_h, st_o, st0 = _run_pair(config, "step3d_uv", util.step_idx(iic=iic))
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    assert util.max_rel_diff(st_o["Huon"], st0["Huon"]) > 1e-6


def _idx2d(iif, pred, iic, first=False):
    """time indices as main3d.F:597-662 sets them (indx1 = 1)"""
    if pred:
        return util.step_idx(iic=iic, iif=iif, pred=1, kstp=1 if iif == 1 else 2, knew=3, krhs=1)
    return util.step_idx(iic=iic, iif=iif, pred=0, knew=2, kstp=1, krhs=3)


@pytest.mark.parametrize("config", CONFIGS)
@pytest.mark.parametrize("iif,pred,iic", [(1, 1, 1), (1, 1, 2), (1, 1, 7), (1, 0, 7), (5, 1, 7), (5, 0, 7)])
def test_step2d(config, iif, pred, iic):
    def prep(st):
        # prepared_state leaves the AM3 history terms and the 3-D forcing of the barotropic mode at zero: give the
        # corrector's 8/12 and 1/12 weights (step2d_LF_AM3.h:823-836, :2150-2255) and the coupling (:1884-2065)
        # something to act on
        b = st.b
        ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None]
        jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :]
        w = np.sin(2.0 * np.pi * 3 * ii / b.Lm + 0.4) * np.cos(np.pi * 2 * jj / b.Mm)
        for lev in range(2):
            st["rzeta"][:, :, lev] = (1.0 + 0.3 * lev) * 1.0e-2 * w
            st["rubar"][:, :, lev] = (1.0 - 0.2 * lev) * 3.0e-1 * w
            st["rvbar"][:, :, lev] = (1.0 + 0.1 * lev) * 2.0e-1 * np.roll(w, 5, axis=0)
        st["rufrc"][:] = 4.0e-1 * np.roll(w, 3, axis=0)
        st["rvfrc"][:] = 2.5e-1 * np.roll(w, 9, axis=0)
    st_h, st_o, st0 = _run_pair(config, "step2d", _idx2d(iif, pred, iic), prep=prep)
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    assert util.max_rel_diff(st_o["ubar"], st0["ubar"]) > 1e-9


@pytest.mark.parametrize("config", CONFIGS)
def test_step2d_last_predictor(config):
    from roms_trunk_mgh_amd import ana
    nfast = ana.make_params(ana.CONFIGS[config], ana.CONFIGS[config]["NAT"]).nfast
    st_h, st_o, st0 = _run_pair(config, "step2d", _idx2d(nfast + 1, 1, 7))
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs


@pytest.mark.parametrize("config", CONFIGS)
def test_step2d_loop(config):
    import oracle
    st0 = util.prepared_state(config)
    st_o, st_h = st0.copy(), st0.copy()
    s1, s2 = util.step_idx(iic=4), util.step_idx(iic=4)
    i_o = oracle.Oracle(st_o).step2d_loop(s1, 1)
    h = hip.RomsHip(st_h)
    try:
        i_h = h.step2d_loop(s2, 1)
        h.to_host()
    finally:
        h.close()
    assert i_o == i_h
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-11 for v in diffs.values()), diffs
    assert np.isfinite(st_o["zeta"]).all()


def _detune_forcing(st):
    st["stflux"][:, :, 0] += 1.0e-6
    st["Vwind"] += 0.3 * st["Uwind"] - 2.0
    st["rain"] += 2.0e-5
    if st.b.NT > 1:
        st["stflux"][:, :, 1] = 2.0e-8
        st["btflx"][:, :, 1] = 1.0e-9


@pytest.mark.parametrize("config", CONFIGS)
def test_set_vbc(config):
    """bottom stress (quadratic / linear drag) and kinematic surface fluxes."""
    st_h, st_o, st0 = _run_pair(config, "set_vbc", util.step_idx(), prep=_detune_forcing)
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    assert util.max_rel_diff(st_o["bustr"], st0["bustr"]) > 1e-6


@pytest.mark.parametrize("config", CONFIGS)
def test_bulk_flux(config):
    """COARE 3.0 bulk fluxes; the device log/exp/pow/atan differ from the host's in the last bits,
    three fixed-point iterations amplify that a little: tolerance 1e-11 of each field's maximum."""
    st_h, st_o, st0 = _run_pair(config, "bulk_flux", util.step_idx(), prep=_detune_forcing)
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-11 for v in diffs.values()), diffs
    for name in ("sustr", "svstr", "lhflx", "shflx", "lrflx", "stflux"):
        assert util.max_rel_diff(st_o[name], st0[name]) > 1e-6, name


def test_lmd_vmix():
    """KPP vertical mixing (lmd_vmix_tile + lmd_skpp + lmd_finish), BENCHMARK option set.  The device
    pow/exp differ from the host's in the last bits: tolerance 1e-10 of each field's maximum."""
    import oracle
    # util.kpp_state: stratified (rho_eos run first), boundary layers inside the top layer and deep ones, surface
    # diffusivities of salinity different from the temperature's
    st0 = util.kpp_state("BENCHMARK_TINY")
    st_o, st_h = st0.copy(), st0.copy()
    oracle.Oracle(st_o).call("lmd_vmix", util.step_idx())
    h = hip.RomsHip(st_h)
    try:
        h.call("lmd_vmix", util.step_idx())
        h.to_host()
    finally:
        h.close()
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-10 for v in diffs.values()), diffs
    for name in ("Akv", "Akt", "ghats", "hsbl"):
        assert util.max_rel_diff(st_o[name], st0[name]) > 1e-6, name
    # both regimes of the boundary-layer search must occur in the test state
    hs = st_o.interior("hsbl")
    assert float(hs.max()) > float(hs.min())


@pytest.mark.parametrize("config", CONFIGS)
def test_wvelocity_and_diag(config):
    """The two diagnostics of every step.  wvelocity: bit for bit.  diag: the three sums are
    accumulated in the reference's order (k per column, then j, then i) and the Courant maximum is found
    in its loop order, so the twelve numbers are equal as well."""
    import oracle
    st0 = util.prepared_state(config)
    _detune(st0)
    s = util.step_idx()
    st_o, st_h = st0.copy(), st0.copy()
    o = oracle.Oracle(st_o)
    o.call("wvelocity", s)
    d_o = o.diag(s)
    h = hip.RomsHip(st_h)
    try:
        h.call("wvelocity", s)
        d_h = h.diag(s)
        h.to_host()
    finally:
        h.close()
    diffs = util.compare_states(st_h, st_o)
    assert all(v == 0.0 for v in diffs.values()), diffs
    assert float(np.abs(st_o["wvel"]).max()) > 0.0
    assert d_o[5] > 0.0 and d_o[8] > 0.0, d_o              # a Courant maximum with a vertical part exists
    assert np.array_equal(d_h, d_o), (d_h, d_o)


@pytest.mark.parametrize("tdays", [0.0, 0.3, 0.5, 200.25])
def test_ana_srflux(tdays):
    """ana_srflux, ALBEDO branch (analytic forcing): night (zero) and day points, two seasons.
    sin/cos/pow of the device differ from the host's in the last bits: 1e-13 of the field maximum."""
    import oracle
    from roms_trunk_mgh_amd import main3d
    st0 = util.prepared_state("BENCHMARK_TINY")
    st0["srflx"][...] = -1.0
    yd, hr = main3d.host_clock(tdays)
    st_o, st_h = st0.copy(), st0.copy()
    oracle.Oracle(st_o).ana_srflux(yd, hr)
    h = hip.RomsHip(st_h)
    try:
        h.ana_srflux(yd, hr)
        h.to_host(["srflx"])
    finally:
        h.close()
    x, y = st_h.interior("srflx"), st_o.interior("srflx")
    assert float(y.min()) >= 0.0
    if tdays != 0.0:
        assert float(y.max()) > 0.0
    assert float(np.abs(x - y).max()) <= 1e-13 * max(float(y.max()), 1e-5)
    assert np.array_equal(x == 0.0, y == 0.0)           # same night side


def test_diag_at_rest():
    """All velocities zero: no Courant number exceeds zero, so the location stays (0,0,0) as in the
    reference's strict comparison."""
    import oracle
    st0 = util.prepared_state("SEAMOUNT")
    for n in ("u", "v", "wvel", "ubar", "vbar"):
        st0[n][...] = 0.0
    s = util.step_idx()
    d_o = oracle.Oracle(st0.copy()).diag(s)
    h = hip.RomsHip(st0.copy())
    try:
        d_h = h.diag(s)
    finally:
        h.close()
    assert list(d_o[5:12]) == [0.0] * 7
    assert np.array_equal(d_h, d_o), (d_h, d_o)


@pytest.mark.parametrize("N", [40, 64])
@pytest.mark.parametrize("kernel", ["step3d_t", "pre_step3d", "step3d_uv", "omega", "rhs3d"])
def test_more_than_32_levels(kernel, N):
    """Column kernels keep Thomas / spline arrays in registers under full unrolling: instantiated for
    N <= 16, 32, 48, 64 (the two larger ones spill into AGPRs: slower per cell, same results)."""
    prep = util.hz_weighted_tnew if kernel == "step3d_t" else None
    st_h, st_o, st0 = _run_pair("UPWELLING", kernel, util.step_idx(iic=5), prep=prep, overrides={"N": N})
    assert st0.b.N == N
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs


@pytest.mark.parametrize("config", CONFIGS)
def test_step2d_loop_reads_metric_arrays_when_they_depend_on_i(config):
    """The barotropic kernel takes its grid metrics from a per-row table when all fifteen arrays are independent of
    i (every analytic grid of this repository) and from the arrays otherwise -- likewise, as a second group, the
    resting depth and the viscosity coefficients: one perturbed value of pm must select the general kernel, one of h
    the metrics-only kernel, and all of them must agree with the oracle bit for bit."""
    import oracle
    from roms_trunk_mgh_amd import hip, main3d
    for perturbed in (None, "pm", "h"):
        st0 = util.prepared_state(config)
        if perturbed:
            st0[perturbed][st0.I(7, 7), st0.J(9, 9)] *= 1.0 + 1.0e-6
        st_o, st_h = st0.copy(), st0.copy()
        s_o, s_h = util.step_idx(iic=5), util.step_idx(iic=5)
        oracle.Oracle(st_o).step2d_loop(s_o, 1)
        h = hip.RomsHip(st_h)
        try:
            h.step2d_loop(s_h, 1)
            state = h.row_metrics_state()
            h.to_host()
        finally:
            h.close()
        # 3: metrics, depth and viscosity from the row table; 1: metrics only (SEAMOUNT's depth depends on i);
        # 2: the arrays
        want = {None: 1 if config == "SEAMOUNT" else 3, "pm": 2, "h": 1}[perturbed]
        assert state == want, (perturbed, state)
        diffs = util.compare_states(st_h, st_o)
        assert all(v <= TOL for v in diffs.values()), (perturbed, diffs)
        assert util.compare_states(st_o, st0)


@pytest.mark.parametrize("config,kernel,mpdata", [("BENCHMARK_TINY", "pre_step3d", False), ("UPWELLING", "pre_step3d", False),
                                                  ("SEAMOUNT", "pre_step3d", False), ("UPWELLING", "step3d_t", True)])
def test_semi_implicit_vertical_mixing(config, kernel, mpdata):
    """lambda < 1 (mod_scalars.F:724-729; the shipped applications use 1): the explicit share dt (1 - lambda) of the
    vertical viscous / diffusive fluxes in pre_step3d.F:838, :918, :1023 -- with lambda = 1 the predictor kernels skip
    that flux altogether, this is the other branch -- and the implicit share lambda dt of the classic tridiagonal
    MPDATA tracers keep (step3d_t.F:1436; the spline-form operators of SPLINES_VDIFF / SPLINES_VVISC, which all three
    applications define, do not contain lambda)."""
    import oracle
    ov = {"Hadv": "MPDATA", "Vadv": "MPDATA"} if mpdata else None

    def prep(st):
        st.p = type(st.p).from_buffer_copy(st.p)
        st.p.lambda_ = 0.75
        if kernel == "step3d_t":
            util.hz_weighted_tnew(st)
    st_h, st_o, st0 = _run_pair(config, kernel, util.step_idx(iic=5), prep=prep, overrides=ov)
    assert st_o.p.lambda_ == 0.75
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= TOL for v in diffs.values()), diffs
    # and it is a different answer from lambda = 1
    st_1 = st0.copy()
    st_1.p = type(st0.p).from_buffer_copy(st0.p)          # copy() shares the parameter block
    st_1.p.lambda_ = 1.0
    oracle.Oracle(st_1).call(kernel, util.step_idx(iic=5))
    assert util.compare_states(st_o, st_1)


@pytest.mark.gpu
def test_set_vbc_log_layer_drag():
    """UV_LOGDRAG (roms_params_t.uv_drag = 3): HIP vs oracle; the device log() is not the host's: 1e-14."""
    import oracle
    import ref_worker
    from roms_trunk_mgh_amd import hip
    st0 = ref_worker.logdrag_state("UPWELLING")
    st_o, st_h = st0.copy(), st0.copy()
    s = util.step_idx()
    oracle.Oracle(st_o).call("set_vbc", s)
    h = hip.RomsHip(st_h)
    try:
        h.call("set_vbc", s)
        h.to_host()
    finally:
        h.close()
    for n in ("bustr", "bvstr", "stflx", "btflx"):
        assert util.max_rel_diff(st_h[n], st_o[n]) <= 1e-14, n
    assert util.max_rel_diff(st_o["bustr"], st0["bustr"]) > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("config,drag", [("UPWELLING", 1), ("BENCHMARK_TINY", 2), ("UPWELLING", 3)])
def test_set_vbc_limited_bottom_stress(config, drag):
    """LIMIT_BSTRESS (roms_params_t.limit_bstress) with each of the three drag laws: HIP vs oracle."""
    import oracle
    from roms_trunk_mgh_amd import hip
    st0 = util.prepared_state(config, overrides={"limit_bstress": 1, "uv_drag": drag})
    st0["rdrag"] *= 60.0
    st0["rdrag2"] *= 4000.0
    st0["ZoBot"][:] = 0.6 * (st0["z_r"][:, :, 0] - st0["z_w"][:, :, 0])
    st_o, st_h = st0.copy(), st0.copy()
    s = util.step_idx()
    oracle.Oracle(st_o).call("set_vbc", s)
    h = hip.RomsHip(st_h)
    try:
        h.call("set_vbc", s)
        h.to_host()
    finally:
        h.close()
    for n in ("bustr", "bvstr"):
        assert util.max_rel_diff(st_h[n], st_o[n]) <= (1e-14 if drag == 3 else 0.0), n
    st_n = st0.copy()
    st_n.p = type(st0.p).from_buffer_copy(st0.p)
    st_n.p.limit_bstress = 0
    oracle.Oracle(st_n).call("set_vbc", s)
    assert util.max_rel_diff(st_n["bustr"], st_o["bustr"]) > 1e-3        # the limit is reached somewhere


@pytest.mark.gpu
@pytest.mark.parametrize("mask", [None, "island"])
def test_bulk_flux_eminusp(mask):
    """EMINUSP (roms_params_t.eminusp): evap and the surface salt flux of bulk_flux.F:883-899, with their exchanges."""
    import oracle
    st0 = util.prepared_state("BENCHMARK_TINY", overrides={"eminusp": 1}, mask=mask)
    st0["rain"] += 2.0e-5
    st_o, st_h = st0.copy(), st0.copy()
    s = util.step_idx()
    oracle.Oracle(st_o).call("bulk_flux", s)
    h = hip.RomsHip(st_h)
    try:
        h.call("bulk_flux", s)
        h.to_host()
    finally:
        h.close()
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-11 for v in diffs.values()), diffs
    assert float(np.abs(st_o["evap"]).max()) > 1e-6 and util.max_rel_diff(st_o["stflux"][:, :, 1], st0["stflux"][:, :, 1]) > 1e-6


@pytest.mark.parametrize("iif,pred", [(1, 1), (5, 1), (5, 0)])
def test_step2d_flather_with_press_compensate(iif, pred):
    """ATM_PRESS + PRESS_COMPENSATE: the air-pressure term in the Flather value (u2dbc_im.F:264-272; pinned in the
    oracle against the reference built with both options), one barotropic call on a basin with Chapman / Flather edges."""
    import oracle
    from roms_trunk_mgh_amd import abi
    import ref_worker
    st0 = util.prepared_state("UPWELLING", overrides={"EWperiodic": False, "atm_press": 1, "press_compensate": 1})
    ref_worker.atm_pressure(st0)
    for sd in ("west", "east", "south", "north"):
        st0.p.lbc[abi.LBS[sd]][abi.LBV["zeta"]] = abi.LBC["Cha"]
        st0.p.lbc[abi.LBS[sd]][abi.LBV["ubar"]] = abi.LBC["Fla"]
        st0.p.lbc[abi.LBS[sd]][abi.LBV["vbar"]] = abi.LBC["Fla"]
    rng = np.random.default_rng(4)
    for name in ("zeta_bry", "ubar_bry", "vbar_bry"):
        st0[name][:] = 1.0e-2 * rng.standard_normal(st0[name].shape)
    s = _idx2d(iif, pred, 7)
    st_o, st_h, st_n = st0.copy(), st0.copy(), st0.copy()
    st_n.p = type(st0.p).from_buffer_copy(st0.p)
    st_n.p.press_compensate = 0
    oracle.Oracle(st_o).call("step2d", s)
    oracle.Oracle(st_n).call("step2d", s)
    h = hip.RomsHip(st_h)
    try:
        h.call("step2d", s)
        h.to_host()
    finally:
        h.close()
    assert not util.compare_states(st_h, st_o)
    assert not np.array_equal(st_o["ubar"], st_n["ubar"]) and not np.array_equal(st_o["vbar"], st_n["vbar"])
