"""-m gpu: without SPLINES_VVISC / SPLINES_VDIFF (step3d_uv.F:400-464, :733-797, step3d_t.F:1431-1501): the tridiagonal
systems for u, v and the tracers themselves.  The HIP library against the CPU oracle through the C ABI, kernel by kernel
and over 100-step runs.  PARITY UNPINNED like the rest of step3d_uv / step3d_t (mod_sources -> netCDF); the oracle's
known answers are in tests/test_classic_vertical.py."""
import numpy as np
import pytest

import util
from roms_trunk_mgh_amd import ana, hip, main3d
from roms_trunk_mgh_amd.state import rel_rms

pytestmark = pytest.mark.gpu

CLASSIC = {"splines_vdiff": 0, "splines_vvisc": 0}

VARIANTS = {
    "channel": dict(config="UPWELLING", overrides={}, mask=None, kind=None),
    "basin_mask": dict(config="UPWELLING", overrides={"EWperiodic": False}, mask="island", kind=None),
    "benchmark": dict(config="BENCHMARK_TINY", overrides={}, mask=None, kind=None),
    "seamount": dict(config="SEAMOUNT", overrides={"EWperiodic": False}, mask=None, kind=None),
    "n40": dict(config="UPWELLING", overrides={"N": 40}, mask=None, kind=None),
    "c4_splines": dict(config="UPWELLING", overrides={"Hadv": "C4", "Vadv": "SPLINES"}, mask=None, kind=None),
    "a4": dict(config="UPWELLING", overrides={"Hadv": "A4", "Vadv": "A4"}, mask="island", kind=None),
    "mpdata": dict(config="UPWELLING", overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"}, mask="island", kind=None),
    "rivers": dict(config="UPWELLING", overrides={"EWperiodic": False}, mask="island", kind="all"),
    "wet": dict(config="UPWELLING", overrides={"EWperiodic": False, "wet_dry": 1, "beach": 1, "zeta_amp": 0.3}, mask=None,
                kind=None),
    # only one of the two switches off
    "vdiff_only": dict(config="UPWELLING", overrides={"splines_vvisc": 1}, mask=None, kind=None),
    "vvisc_only": dict(config="UPWELLING", overrides={"splines_vdiff": 1}, mask=None, kind=None),
}


def _ov(v):
    return dict(CLASSIC, **v["overrides"])


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("kernel", ["step3d_uv", "step3d_t"])
def test_classic_kernels(variant, kernel):
    import oracle
    v = VARIANTS[variant]
    st0 = util.prepared_state(v["config"], overrides=_ov(v), mask=v["mask"], wet=bool(v["overrides"].get("wet_dry")))
    if v["kind"]:
        src = util.river_sources(st0, v["kind"])
        q = src.qsrc()
        for n, (i, j, d) in enumerate(zip(src.Isrc, src.Jsrc, src.Dsrc)):
            if int(d) < 2:
                st0["Huon" if int(d) == 0 else "Hvom"][st0.I(i), st0.J(j), :] = q[n]
    if kernel == "step3d_t":
        util.hz_weighted_tnew(st0)
    st_o, st_h = st0.copy(), st0.copy()
    s = util.step_idx(iic=5)
    be_o = oracle.Oracle(st_o)
    be_o.call(kernel, s)
    h = hip.RomsHip(st_h)
    try:
        h.call(kernel, s)
        h.to_host()
    finally:
        h.close()
    diffs = util.compare_states(st_h, st_o)
    assert all(x <= 1e-13 for x in diffs.values()), diffs
    # the switch acts: the spline form gives another answer
    # (MPDATA tracers take the tridiagonal form under SPLINES_VDIFF as well, step3d_t.F:1431)
    acts = not st0.p.splines_vvisc if kernel == "step3d_uv" else (not st0.p.splines_vdiff and v["overrides"].get("Hadv") != "MPDATA")
    if acts:
        st_s = st0.copy()
        st_s.p = type(st0.p).from_buffer_copy(st0.p)
        st_s.p.splines_vdiff = st_s.p.splines_vvisc = 1
        if v["kind"]:
            st_s.sources = st0.sources
        oracle.Oracle(st_s).call(kernel, s)
        assert any(x > 1e-10 for x in util.compare_states(st_s, st_o).values()), "the switch is without effect"


@pytest.mark.parametrize("variant,physics", [("channel", False), ("basin_mask", False), ("benchmark", True), ("seamount", False),
                                             ("c4_splines", False), ("mpdata", False), ("rivers", False), ("wet", False),
                                             ("vdiff_only", False), ("vvisc_only", False)])
def test_100_steps_classic(variant, physics):
    import oracle
    v = VARIANTS[variant]
    st_o = ana.make_tile(v["config"], perturb=1.0 if v["config"] != "SEAMOUNT" else 0.0, overrides=_ov(v), mask=v["mask"])
    if v["kind"]:
        util.river_sources(st_o, v["kind"])
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
    assert np.isfinite(st_h["t"]).all() and np.isfinite(st_o["t"]).all()
    assert all(x <= 1e-10 for x in out.values()), out          # north-star bound


def test_hsimt_without_splines_vdiff_is_refused():
    st = util.prepared_state("UPWELLING", overrides=dict(CLASSIC, Hadv="HSIMT", Vadv="HSIMT"))
    util.hz_weighted_tnew(st)
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError, match="HSIMT without SPLINES_VDIFF"):
            h.call("step3d_t", util.step_idx(iic=5))
    finally:
        h.close()
