"""-m gpu: biharmonic lateral mixing (TS_DIF4, UV_VIS4; SURVEY.md section 8f-4 "other selectable numerics").  HIP
against the CPU oracle through the C ABI; the oracle's t3dmix4 / uv3dmix4 are pinned bit for bit against the
reference's Fortran (tests/test_ref_pinning.py, tests/test_golden.py); the 2-D operator inside step2d
(step2d_LF_AM3.h:1474-1740) belongs to an unpinned routine and is checked against the oracle only."""
import numpy as np
import pytest

import util
from roms_trunk_mgh_amd import abi, ana, hip, main3d
from roms_trunk_mgh_amd.state import rel_rms

pytestmark = pytest.mark.gpu
DIF4 = {"UPWELLING": {"ts_dif4": 1, "uv_vis4": 1, "tnu4": 2.0e7, "visc4": 4.0e7},          # t3dmix4_s
        "SEAMOUNT": {"ts_dif4": 1, "uv_vis4": 1, "tnu4": 1.0e8, "visc4": 1.0e8},           # t3dmix4_geo
        "BENCHMARK_TINY": {"ts_dif4": 1, "uv_vis4": 1, "tnu4": 1.0e10, "visc4": 2.0e10}}   # t3dmix4_geo, curvilinear terms


def _state(config, variant, extra=None):
    ov = dict(DIF4[config], **(extra or {}))
    if variant in ("closed", "open"):
        ov["EWperiodic"] = False
    st = util.prepared_state(config, overrides=ov, mask="island" if variant == "mask" else None)
    if variant == "open":
        for sd in ("west", "east", "south", "north"):
            for var in ("ubar", "vbar", "u", "v", "t"):
                st.p.lbc[abi.LBS[sd]][abi.LBV[var]] = abi.LBC["Gra"]
    assert st.b.NghostPoints == 3
    return st


@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT", "BENCHMARK_TINY"])
@pytest.mark.parametrize("variant", ["periodic", "closed", "open", "mask"])
@pytest.mark.parametrize("kernel", ["t3dmix4", "uv3dmix4", "rhs3d", "step2d"])
def test_biharmonic_kernels(config, variant, kernel):
    import oracle
    st0 = _state(config, variant)
    st_o, st_h = st0.copy(), st0.copy()
    preds = [(5, 1, 0)] if kernel != "step2d" else [(5, 1, 1), (5, 2, 1), (5, 2, 0)]
    for iic, iif, pred in preds:
        s = util.step_idx(iic=iic, iif=iif, pred=pred, knew=3 if pred else 2, krhs=1 if pred else 3)
        oracle.Oracle(st_o).call(kernel, s)
        h = hip.RomsHip(st_h)
        try:
            h.call(kernel, s)
            h.to_host()
        finally:
            h.close()
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-12 for v in diffs.values()), diffs
    assert util.compare_states(st_o, st0)


def test_biharmonic_term_is_in_step2d():
    """the 2-D operator changes the barotropic right-hand side: the same call without UV_VIS4 gives another ubar"""
    import oracle
    st4 = _state("UPWELLING", "periodic")
    st2 = st4.copy()
    st2.p = type(st4.p).from_buffer_copy(st4.p)
    st2.p.uv_vis4 = 0
    s = util.step_idx(iic=5, iif=2, pred=1, knew=3, krhs=1)
    out = []
    for st in (st4, st2):
        h = hip.RomsHip(st)
        try:
            h.call("step2d", s)
            h.to_host()
        finally:
            h.close()
        out.append(st["ubar"][:, :, 2].copy())
    assert util.max_rel_diff(out[0], out[1]) > 1e-9


def test_uv_vis4_needs_three_ghost_points():
    st = ana.make_tile("UPWELLING", perturb=1.0)           # two ghost points
    st.p.uv_vis4 = 1
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError) as e:
            h.call("uv3dmix4", util.step_idx())
        assert "NghostPoints = 3" in str(e.value)
    finally:
        h.close()


@pytest.mark.parametrize("config,variant", [("UPWELLING", "periodic"), ("SEAMOUNT", "periodic"), ("UPWELLING", "closed"),
                                            ("BENCHMARK_TINY", "mask")])
def test_100_steps_with_biharmonic_mixing(config, variant):
    import oracle
    ov = dict(DIF4[config])
    if variant == "closed":
        ov["EWperiodic"] = False
    st_o = ana.make_tile(config, perturb=1.0 if config != "SEAMOUNT" else 0.0, overrides=ov,
                         mask="island" if variant == "mask" else None)
    st_h = st_o.copy()
    mo = main3d.Main3D(oracle.Oracle(st_o))
    mo.initial()
    mo.run(100)
    be = hip.RomsHip(st_h)
    try:
        mh = main3d.Main3D(be)
        mh.initial()
        mh.run(100)
        be.to_host()
    finally:
        be.close()
    s = mo.s
    out = {"zeta": rel_rms(st_h.interior("zeta")[..., mo.indx1 - 1], st_o.interior("zeta")[..., mo.indx1 - 1], 1e-3)}
    for name in ("u", "v"):
        out[name] = rel_rms(st_h.interior(name)[..., s.nnew - 1], st_o.interior(name)[..., s.nnew - 1], 1e-4)
    for it in range(st_o.b.NT):
        out[f"t{it+1}"] = rel_rms(st_h.interior("t")[..., s.nnew - 1, it], st_o.interior("t")[..., s.nnew - 1, it], 1e-3)
    assert np.isfinite(st_h["t"]).all() and np.isfinite(st_o["t"]).all()
    assert all(v <= 1e-10 for v in out.values()), out          # north-star bound


# ---- MIX_ISO_TS: tracer mixing along isopycnals, harmonic and biharmonic (t3dmix2_iso.h, t3dmix4_iso.h) ----
@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT", "BENCHMARK_TINY"])
@pytest.mark.parametrize("variant", ["periodic", "closed", "open", "mask"])
@pytest.mark.parametrize("kernel", ["t3dmix2", "t3dmix4", "rhs3d"])
def test_isopycnal_kernels(config, variant, kernel):
    import oracle
    import ref_worker
    st0 = ref_worker.iso_state(config, basin=variant if variant in ("closed", "open") else None,
                               mask="island" if variant == "mask" else None, extra=DIF4[config])
    st_o, st_h = st0.copy(), st0.copy()
    s = util.step_idx(iic=5)
    oracle.Oracle(st_o).call(kernel, s)
    h = hip.RomsHip(st_h)
    try:
        h.call(kernel, s)
        h.to_host()
    finally:
        h.close()
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-12 for v in diffs.values()), diffs
    assert util.max_rel_diff(st_o["t"], st0["t"]) > 1e-6


@pytest.mark.parametrize("config,dif4", [("BENCHMARK_TINY", False), ("UPWELLING", True)])
def test_100_steps_with_isopycnal_mixing(config, dif4):
    import oracle
    ov = {"mix_iso_ts": 1}
    if dif4:
        ov.update(DIF4[config])
    else:
        ov["tnu2"] = 200.0
    st_o = ana.make_tile(config, perturb=1.0, overrides=ov)
    st_h = st_o.copy()
    mo = main3d.Main3D(oracle.Oracle(st_o))
    mo.initial()
    mo.run(100)
    be = hip.RomsHip(st_h)
    try:
        mh = main3d.Main3D(be)
        mh.initial()
        mh.run(100)
        be.to_host()
    finally:
        be.close()
    s = mo.s
    out = {"zeta": rel_rms(st_h.interior("zeta")[..., mo.indx1 - 1], st_o.interior("zeta")[..., mo.indx1 - 1], 1e-3)}
    for name in ("u", "v"):
        out[name] = rel_rms(st_h.interior(name)[..., s.nnew - 1], st_o.interior(name)[..., s.nnew - 1], 1e-4)
    for it in range(st_o.b.NT):
        out[f"t{it+1}"] = rel_rms(st_h.interior("t")[..., s.nnew - 1, it], st_o.interior("t")[..., s.nnew - 1, it], 1e-3)
    assert np.isfinite(st_h["t"]).all() and np.isfinite(st_o["t"]).all()
    assert all(v <= 1e-10 for v in out.values()), out
    assert float(np.abs(st_o["pden"]).max()) > 0.0


# ---- TS_MIX_STABILITY: 3/4 t(nrhs) + 1/4 t(nstp) in every tracer difference of the operators ----
@pytest.mark.parametrize("config,iso", [("UPWELLING", False), ("SEAMOUNT", False), ("BENCHMARK_TINY", False),
                                        ("UPWELLING", True), ("SEAMOUNT", True)])
@pytest.mark.parametrize("variant", ["periodic", "open", "mask"])
@pytest.mark.parametrize("kernel", ["t3dmix2", "t3dmix4"])
def test_ts_mix_stability_kernels(config, iso, variant, kernel):
    """s-surfaces (UPWELLING), geopotentials (SEAMOUNT, BENCHMARK_TINY) and isopycnals, with two distinct time levels
    (nrhs = 3, nstp = 1) and with the model's own nrhs = nstp; the oracle is pinned against the reference built with
    -DTS_MIX_STABILITY (tests/test_ref_pinning.py::test_ts_mix_stability_matches_reference_build)."""
    import oracle
    import ref_worker
    if iso:
        st0 = ref_worker.iso_state(config, basin=variant if variant == "open" else None,
                                   mask="island" if variant == "mask" else None,
                                   extra=dict(DIF4[config], ts_mix_stability=1))
    else:
        st0 = _state(config, variant, extra={"tnu2": 300.0, "ts_mix_stability": 1})
    assert st0.p.ts_mix_stability == 1
    for s in (util.step_idx(iic=5, nstp=1, nnew=2, nrhs=3), util.step_idx(iic=5)):
        st_o, st_h, st_p = st0.copy(), st0.copy(), st0.copy()
        oracle.Oracle(st_o).call(kernel, s)
        st_p.p = type(st0.p).from_buffer_copy(st0.p)
        st_p.p.ts_mix_stability = 0
        oracle.Oracle(st_p).call(kernel, s)
        h = hip.RomsHip(st_h)
        try:
            h.call(kernel, s)
            h.to_host()
        finally:
            h.close()
        assert np.array_equal(st_h["t"], st_o["t"]), util.compare_states(st_h, st_o)
        assert util.max_rel_diff(st_o["t"], st0["t"]) > 1e-6
        if s.nrhs != s.nstp:
            assert util.max_rel_diff(st_o["t"], st_p["t"]) > 1e-9       # the option acts


@pytest.mark.parametrize("config,iso", [("UPWELLING", False), ("SEAMOUNT", False), ("BENCHMARK_TINY", True)])
def test_100_steps_with_ts_mix_stability(config, iso):
    import oracle
    ov = dict(DIF4[config], ts_mix_stability=1)
    if iso:
        ov["mix_iso_ts"] = 1
    st_o = ana.make_tile(config, perturb=1.0 if config != "SEAMOUNT" else 0.0, overrides=ov)
    st_h = st_o.copy()
    assert st_o.p.ts_mix_stability == 1
    mo = main3d.Main3D(oracle.Oracle(st_o))
    mo.initial()
    mo.run(100)
    be = hip.RomsHip(st_h)
    try:
        mh = main3d.Main3D(be)
        mh.initial()
        mh.run(100)
        be.to_host()
    finally:
        be.close()
    s = mo.s
    out = {"zeta": rel_rms(st_h.interior("zeta")[..., mo.indx1 - 1], st_o.interior("zeta")[..., mo.indx1 - 1], 1e-3)}
    for name in ("u", "v"):
        out[name] = rel_rms(st_h.interior(name)[..., s.nnew - 1], st_o.interior(name)[..., s.nnew - 1], 1e-4)
    for it in range(st_o.b.NT):
        out[f"t{it+1}"] = rel_rms(st_h.interior("t")[..., s.nnew - 1, it], st_o.interior("t")[..., s.nnew - 1, it], 1e-3)
    assert np.isfinite(st_h["t"]).all() and np.isfinite(st_o["t"]).all()
    assert all(v <= 1e-10 for v in out.values()), out          # north-star bound


# ---- TS_MIX_MIN_STRAT: the slope scale of the isopycnal operators bounded by strat_min * dz ----
@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT", "BENCHMARK_TINY"])
@pytest.mark.parametrize("variant", ["periodic", "open", "mask"])
@pytest.mark.parametrize("kernel", ["t3dmix2", "t3dmix4"])
def test_ts_mix_min_strat_kernels(config, variant, kernel):
    """the oracle is pinned against the reference built with -DTS_MIX_MIN_STRAT
    (tests/test_ref_pinning.py::test_ts_mix_min_strat_matches_reference_build)"""
    import oracle
    import ref_worker
    st0 = ref_worker.iso_state(config, basin=variant if variant == "open" else None,
                               mask="island" if variant == "mask" else None,
                               extra=dict(DIF4[config], ts_mix_min_strat=1))
    assert st0.p.ts_mix_min_strat == 1
    s = util.step_idx(iic=5)
    st_o, st_h, st_p = st0.copy(), st0.copy(), st0.copy()
    oracle.Oracle(st_o).call(kernel, s)
    st_p.p = type(st0.p).from_buffer_copy(st0.p)
    st_p.p.ts_mix_min_strat = 0
    oracle.Oracle(st_p).call(kernel, s)
    h = hip.RomsHip(st_h)
    try:
        h.call(kernel, s)
        h.to_host()
    finally:
        h.close()
    assert np.array_equal(st_h["t"], st_o["t"]), util.compare_states(st_h, st_o)
    assert util.max_rel_diff(st_o["t"], st_p["t"]) > 1e-9           # the option acts
