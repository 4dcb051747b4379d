"""-m gpu: harmonic viscosity rotated to geopotential surfaces (UV_VIS2 with MIX_GEO_UV, uv3dmix2_geo.h:116-756;
roms_params_t.uv_vis2 = 2; SURVEY.md section 8f-4 "other selectable numerics").  k_uv3dmix2_geo through the C ABI against
the CPU oracle, whose uv3dmix2_geo is pinned bit for bit against four reference builds with the option
(tests/test_golden.py, ref_geouv.npz).  The kernel evaluates the reference's expressions in the reference's order:
the comparison is for identical bits."""
import numpy as np
import pytest

import util
from roms_trunk_mgh_amd import ana, hip, main3d
from roms_trunk_mgh_amd.state import rel_rms

pytestmark = pytest.mark.gpu
VISC = {"SEAMOUNT": {"visc2": 50.0}}          # the application itself has no UV_VIS2 (visc2 = 0)


def _state(config, variant):
    ov = {"uv_vis2": 2, **VISC.get(config, {})}
    if variant in ("closed", "mask_closed"):
        ov["EWperiodic"] = False
    return util.prepared_state(config, overrides=ov, mask="island" if variant.startswith("mask") else None,
                               wet=(variant == "wet") or None)


@pytest.mark.parametrize("config", ["UPWELLING", "SEAMOUNT", "BENCHMARK_TINY"])
@pytest.mark.parametrize("variant", ["periodic", "closed", "mask", "mask_closed", "wet"])
@pytest.mark.parametrize("kernel", ["uv3dmix2", "rhs3d"])
def test_uv3dmix2_geo_kernel(config, variant, kernel):
    import oracle
    st0 = _state(config, variant)
    assert st0.p.uv_vis2 == 2
    st_o, st_h = st0.copy(), st0.copy()
    s = util.step_idx(iic=5, iif=1, pred=0, knew=2, krhs=3)
    oracle.Oracle(st_o).call(kernel, s)
    h = hip.RomsHip(st_h)
    try:
        h.call(kernel, s)
        h.to_host()
    finally:
        h.close()
    for name in ("u", "v", "rufrc", "rvfrc"):
        assert np.array_equal(st_h[name], st_o[name]), (name, float(np.abs(st_h[name] - st_o[name]).max()))
    assert not np.array_equal(st_o["u"], st0["u"]) and not np.array_equal(st_o["rvfrc"], st0["rvfrc"])


def test_geo_differs_from_s_surfaces():
    """the switch reaches the kernel: on the sloping levels of SEAMOUNT the rotated operator gives another u"""
    out = []
    for vis in (1, 2):
        st = util.prepared_state("SEAMOUNT", overrides={"uv_vis2": vis, "visc2": 50.0})
        s = util.step_idx(iic=5, iif=1, pred=0, knew=2, krhs=3)
        h = hip.RomsHip(st)
        try:
            h.call("uv3dmix2", s)
            h.to_host()
        finally:
            h.close()
        out.append(st["u"][:, :, :, s.nnew - 1].copy())
    assert util.max_rel_diff(out[0], out[1]) > 1e-9


@pytest.mark.parametrize("config,mask,basin,physics", [("SEAMOUNT", None, False, False), ("BENCHMARK_TINY", "island", True, True),
                                                       ("UPWELLING", None, False, False)])
def test_100_steps_geo_uv(config, mask, basin, physics):
    import oracle
    ov = {"uv_vis2": 2, **VISC.get(config, {})}
    if basin:
        ov["EWperiodic"] = False
    st_o = ana.make_tile(config, perturb=1.0 if config != "SEAMOUNT" else 0.0, overrides=ov, mask=mask)
    st_h = st_o.copy()
    mo = main3d.Main3D(oracle.Oracle(st_o), physics=physics, diagnostics=physics)
    mo.initial()
    mo.run(100)
    be = hip.RomsHip(st_h)
    try:
        mh = main3d.Main3D(be, physics=physics, diagnostics=physics)
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
    assert np.isfinite(st_h["u"]).all() and np.isfinite(st_o["u"]).all()
    assert all(x <= 1e-10 for x in out.values()), out          # north-star bound
