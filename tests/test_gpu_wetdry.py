"""-m gpu: WET_DRY (wetting and drying), HIP vs CPU oracle through the C ABI.
The oracle's WET_DRY blocks of the buildable files are pinned bit for bit against the reference built with -DWET_DRY
(tests/test_ref_pinning.py, tests/test_golden.py); wetdry.F and the blocks in step2d / step3d_uv / pre_step3d have the
known-answer tests of tests/test_wetdry.py (which run on the HIP path too under -m gpu)."""
import numpy as np
import pytest

import util
from roms_trunk_mgh_amd import abi, ana, hip, main3d

pytestmark = pytest.mark.gpu
TOL = 1e-12
BEACH = {"wet_dry": 1, "beach": 1, "zeta_amp": 0.3}


def _pair(st0, fn):
    import oracle
    st_o, st_h = st0.copy(), st0.copy()
    fn(oracle.Oracle(st_o))
    h = hip.RomsHip(st_h)
    try:
        fn(h)
        h.to_host()
    finally:
        h.close()
    return st_h, st_o


def _wet_state(config, mask="island", overrides=None, basin=False, NT=None):
    ov = dict(overrides or {})
    if basin:
        ov["EWperiodic"] = False
    st = util.prepared_state(config, overrides=ov, mask=mask, wet=True, NT=NT)
    st["h"][7 - st.b.LBi, 9 - st.b.LBj] = 0.0
    return st


@pytest.mark.parametrize("config", ["UPWELLING", "BENCHMARK_TINY", "SEAMOUNT"])
@pytest.mark.parametrize("kernel", ["set_depth", "prsgrd", "t3dmix2", "uv3dmix2", "pre_step3d", "rhs3d", "step3d_uv",
                                    "ini_zeta", "ini_fields", "wetdry"])
def test_wet_kernels_vs_oracle(config, kernel):
    """Each kernel that has a WET_DRY block, on a state whose wet/dry masks hold 0, 1, 2 and -1: bit for bit."""
    if kernel == "uv3dmix2" and config == "SEAMOUNT":
        pytest.skip("no UV_VIS2 in SEAMOUNT")
    st0 = _wet_state(config, overrides={"tnu2": 300.0} if config == "SEAMOUNT" else {"tnu2": 300.0, "visc2": 800.0})
    s = util.step_idx(iic=4)
    if kernel.startswith("ini"):
        s = util.step_idx(iic=1, iif=1, pred=0, kstp=1, krhs=1, knew=1)
        s.nstp, s.nnew, s.nrhs = 1, 2, 1
        # shallow stretches so that the Dcrit floor of ini_zeta acts
        st0["h"][:, :6][::3] = 0.16
    st_h, st_o = _pair(st0, lambda be: be.call(kernel, s))
    diffs = util.compare_states(st_h, st_o)
    assert not diffs, diffs
    assert util.compare_states(st_o, st0), "kernel did not modify anything: test is vacuous"
    # ... and the wet/dry masks mattered
    st_n = st0.copy()
    st_n.p = type(st0.p).from_buffer_copy(st0.p)
    st_n.p.wet_dry = 0
    import oracle
    # (pre_step3d's only block is the solar source term, which BENCHMARK alone has)
    if kernel != "wetdry" and not (kernel == "pre_step3d" and config != "BENCHMARK_TINY"):
        oracle.Oracle(st_n).call(kernel, s)
        assert util.compare_states(st_n, st_o), "WET_DRY made no difference: test is vacuous"


@pytest.mark.parametrize("pgf", ["STANDARD", "WJ_GRADP"])
def test_wet_prsgrd31(pgf):
    st0 = _wet_state("UPWELLING", overrides={"pgf": pgf})
    st_h, st_o = _pair(st0, lambda be: be.call("prsgrd", util.step_idx()))
    assert not util.compare_states(st_h, st_o)


def test_wet_pj_gradp_is_refused():
    """PJ_GRADP with WET_DRY does not compile in the reference (prsgrd40.h:98-100): refused by library and oracle."""
    import oracle
    st = _wet_state("UPWELLING", overrides={"pgf": "PJ_GRADP"})
    with pytest.raises(RuntimeError):
        oracle.Oracle(st.copy()).call("prsgrd", util.step_idx())
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError) as e:
            h.call("prsgrd", util.step_idx())
        assert "PJ_GRADP" in str(e.value)
    finally:
        h.close()


def test_wet_without_masking_is_refused():
    st = util.prepared_state("UPWELLING")
    st.p.wet_dry = 1
    st.p.masking = 0
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError) as e:
            h.call("set_depth", util.step_idx())
        assert "masking" in str(e.value)
    finally:
        h.close()


@pytest.mark.parametrize("kernel", ["t3dmix4", "uv3dmix4"])
@pytest.mark.parametrize("basin", [False, True])
def test_wet_biharmonic(kernel, basin):
    st0 = _wet_state("UPWELLING", overrides={"ts_dif4": 1, "uv_vis4": 1, "tnu4": 2.0e7, "visc4": 4.0e7}, basin=basin)
    st_h, st_o = _pair(st0, lambda be: be.call(kernel, util.step_idx()))
    assert not util.compare_states(st_h, st_o)


def test_wet_bulk_flux():
    st0 = _wet_state("BENCHMARK_TINY")
    st0["Vwind"] += 0.3 * st0["Uwind"] - 2.0
    st0["rain"] += 2.0e-5
    st_h, st_o = _pair(st0, lambda be: be.call("bulk_flux", util.step_idx()))
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-12 for v in diffs.values()), diffs


def test_wet_mpdata_step3d_t():
    st0 = _wet_state("BENCHMARK_TINY", overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"}, NT=3)
    util.hz_weighted_tnew(st0)
    st_h, st_o = _pair(st0, lambda be: be.call("step3d_t", util.step_idx(iic=4)))
    assert not util.compare_states(st_h, st_o)


def _idx2d(iif, pred, iic):
    if pred:
        return util.step_idx(iic=iic, iif=iif, pred=1, kstp=1 if iif == 1 else 2, knew=3, krhs=1)
    return util.step_idx(iic=iic, iif=iif, pred=0, knew=2, kstp=1, krhs=3)


def _prep2d(st):
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
    # shallow patches: some cells fall dry in this call (zeta of prepared_state lies within +-0.13)
    st["h"][((ii // 4) % 3 == 0) & ((jj // 5) % 4 == 1)] = 0.16
    st["rmask_wet_avg"][:] = np.floor(3.0 * (1.0 + w))


@pytest.mark.parametrize("basin,lbc", [(False, None), (True, None), (True, "open")])
@pytest.mark.parametrize("iif,pred,iic", [(1, 1, 1), (1, 1, 7), (1, 0, 7), (5, 1, 7), (5, 0, 7), (-1, 1, 7)])
def test_wet_step2d(basin, lbc, iif, pred, iic):
    """One barotropic call with WET_DRY: the masks, zeta with its Dcrit rule, the wet/dry factor of ubar / vbar and of
    rufrc / ru(:,:,0,nstp), the boundary rules (closed walls; Chapman / Flather / Shchepetkin on a basin) -- bit for bit.
    iif = -1: the call after the last fast step (the masks of the baroclinic step)."""
    st0 = _wet_state("UPWELLING", overrides={"visc2": 800.0}, basin=basin)
    _prep2d(st0)
    if lbc == "open":
        for sd, u, v in (("west", "Shc", "Fla"), ("east", "Fla", "Shc"), ("south", "Fla", "Shc"), ("north", "Shc", "Fla")):
            st0.p.lbc[abi.LBS[sd]][abi.LBV["zeta"]] = abi.LBC["Cha"]
            st0.p.lbc[abi.LBS[sd]][abi.LBV["ubar"]] = abi.LBC[u]
            st0.p.lbc[abi.LBS[sd]][abi.LBV["vbar"]] = abi.LBC[v]
        rng = np.random.default_rng(3)
        for name in ("zeta_bry", "ubar_bry", "vbar_bry"):
            st0[name][:] = 1.0e-2 * rng.standard_normal(st0[name].shape)
    if iif == -1:
        iif = st0.p.nfast + 1
    s = _idx2d(iif, pred, iic)
    st_h, st_o = _pair(st0, lambda be: be.call("step2d", s))
    diffs = util.compare_states(st_h, st_o)
    assert not diffs, diffs
    assert not np.array_equal(st_o["umask_wet"], st0["umask_wet"])


@pytest.mark.parametrize("variant", ["channel", "island", "basin"])
def test_wet_three_steps_bitwise(variant):
    """Three whole steps on the drying beach: every field of the state equal bit for bit."""
    ov = dict(BEACH)
    if variant == "basin":
        ov["EWperiodic"] = False

    def run(be):
        m = main3d.Main3D(be)
        m.initial()
        m.run(3)
    st0 = ana.make_tile("UPWELLING", perturb=1.0, overrides=ov, mask="island" if variant == "island" else None)
    st_h, st_o = _pair(st0, run)
    diffs = util.compare_states(st_h, st_o)
    assert not diffs, diffs
    assert 0 < st_o["rmask_wet"].sum() < st_o["rmask_wet"].size


def test_wet_100_steps():
    """100 steps of the drying beach, HIP vs oracle: the wet/dry masks identical, the prognostic fields within the
    north-star bound."""
    from roms_trunk_mgh_amd.state import rel_rms

    def run(be):
        m = main3d.Main3D(be)
        m.initial()
        m.run(100)
        run.m = m
    st0 = ana.make_tile("UPWELLING", perturb=1.0, overrides=BEACH)
    st_h, st_o = _pair(st0, run)
    m = run.m
    for name in ("rmask_wet", "umask_wet", "vmask_wet", "pmask_wet", "rmask_wet_avg"):
        assert np.array_equal(st_h[name], st_o[name]), name
    out = {"zeta": rel_rms(st_h.interior("zeta")[..., m.indx1 - 1], st_o.interior("zeta")[..., m.indx1 - 1], 1e-3)}
    for name in ("u", "v"):
        out[name] = rel_rms(st_h.interior(name)[..., m.s.nnew - 1], st_o.interior(name)[..., m.s.nnew - 1], 1e-4)
    for it in range(st_o.b.NT):
        out[f"t{it+1}"] = rel_rms(st_h.interior("t")[..., m.s.nnew - 1, it], st_o.interior("t")[..., m.s.nnew - 1, it], 1e-3)
    assert all(v <= 1e-10 for v in out.values()), out
