"""Point sources / sinks: through u- and v-faces (LuvSrc, rivers) and at cell centres (LwSrc); SURVEY.md section 8f.
PARITY UNPINNED: step2d, step3d_uv, step3d_t, pre_step3d, omega and wetdry USE mod_sources, which needs netCDF, so none
of the reference's source blocks can be built here.  What stands in for the pin: the answers a river has to give --
  * the volume of a closed basin grows by the net discharge times dt, step after step;
  * a river of ambient water leaves a uniform tracer uniform (every piece -- the barotropic and baroclinic velocities
    of the source faces, the corrected mass fluxes, omega, the predictor's and the corrector's tracer flux -- has to be
    consistent for that), for each advection scheme and on a land mask;
  * the tracer content changes by dt * sum_k Huon(source face, k) * Tsrc(k), nothing else;
and the reference's own acceptance rule, identical results on every tiling (tests/test_multitile_gloo.py, "river").
The HIP library against this oracle: tests/test_gpu_sources.py."""
import numpy as np
import pytest

import oracle
import util
from roms_trunk_mgh_amd import ana, main3d


def _river_state(config, same=None, mask=None, kind="walls", basin=True, overrides=None, NT=None):
    ov = dict(overrides or {})
    if basin:
        ov["EWperiodic"] = False
    st = ana.make_tile(config, perturb=0.0, overrides=ov, mask=mask, NT=NT)
    for n in ("sustr", "svstr", "stflx", "btflx", "srflx"):          # the rivers are the only forcing
        st[n][:] = 0.0
    if same is not None:
        for it in range(st.b.NT):
            st["t"][:, :, :, :, it] = same + it
            if mask:
                st["t"][:, :, :, :, it] *= st["rmask"][:, :, None, None]
    src = util.river_sources(st, kind, same_tracer=same if same is not None else False)
    # (river_sources gave the state its own parameter block) BENCHMARK: no solar heating, no non-local KPP transport
    st.p.lmd_nonlocal = 0
    st.p.solar_source = 0
    if same is not None:
        for it in range(st.b.NT):
            src.Tsrc[:, :, it] = same + it
    return st, src


def _interior(st):
    b = st.b
    return (st.I(b.Istr, b.Iend), st.J(b.Jstr, b.Jend))


def _volume(st):
    sl = _interior(st)
    area = 1.0 / (st["pm"] * st["pn"])
    return float((st["Hz"][sl].sum(axis=2) * area[sl]).sum())


@pytest.mark.parametrize("kind", ["walls", "wells", "all"])
def test_basin_volume_grows_by_the_net_discharge(kind):
    st, src = _river_state("UPWELLING", same=14.0, kind=kind)
    mo = main3d.Main3D(oracle.Oracle(st))
    mo.initial()
    mo.run(2)                                   # the forward first step has its own time centring
    # into the basin: the wall faces of util.river_sources are south (in), west (in), east (out); a cell-centred source
    # counts as it stands
    qnet = 0.0
    for i, d, q in zip(src.Isrc, src.Dsrc, src.Qbar):
        qnet += q if int(d) == 2 else (-q if i == st.b.Lm + 1 else q)
    v = [_volume(st)]
    for _ in range(4):
        mo.run(5)
        v.append(_volume(st))
    inc = np.diff(v) / (qnet * st.p.dt * 5)
    assert np.all(np.abs(inc - 1.0) < 1e-7), inc
    assert float(np.abs(st["u"]).max()) > 1e-3           # the rivers drive a flow


@pytest.mark.parametrize("config,overrides,mask,kind,basin", [
    ("UPWELLING", {}, None, "walls", True),                                     # U3 / C4, walls of a basin
    ("UPWELLING", {}, None, "walls", False),                                    # a periodic channel
    ("UPWELLING", {}, "island", "both", True),                                  # land mask: coast faces
    ("SEAMOUNT", {}, None, "walls", True),                                      # A4 / A4
    ("BENCHMARK_TINY", {}, "island", "both", True),                             # curvilinear, spherical
    ("UPWELLING", {"Hadv": "MPDATA", "Vadv": "MPDATA"}, "island", "both", True),
    ("UPWELLING", {"Hadv": "HSIMT", "Vadv": "HSIMT"}, None, "walls", True),
    ("UPWELLING", {"Hadv": "C2", "Vadv": "C2"}, None, "walls", True),
    ("UPWELLING", {"Hadv": "C4", "Vadv": "SPLINES"}, None, "walls", True),
    # cell-centred sources (LwSrc), alone and beside the face sources
    ("UPWELLING", {}, None, "wells", True), ("UPWELLING", {}, "island", "all", True),
    ("BENCHMARK_TINY", {}, None, "all", False), ("SEAMOUNT", {}, None, "wells", True),
    ("UPWELLING", {"Hadv": "MPDATA", "Vadv": "MPDATA"}, "island", "all", True),
    ("UPWELLING", {"Hadv": "HSIMT", "Vadv": "HSIMT"}, None, "all", True)])
def test_a_river_of_ambient_water_keeps_tracers_uniform(config, overrides, mask, kind, basin):
    T0 = 14.0
    st, src = _river_state(config, same=T0, mask=mask, kind=kind, basin=basin, overrides=overrides)
    mo = main3d.Main3D(oracle.Oracle(st))
    mo.initial()
    mo.run(30)
    sl = _interior(st)
    wet = st["rmask"][sl] == 1.0 if mask else np.ones(st["h"][sl].shape, bool)
    for it in range(st.b.NT):
        t = st["t"][sl][..., mo.s.nnew - 1, it][wet]
        assert float(np.abs(t - (T0 + it)).max()) < 2e-11 * (T0 + it), (it, float(np.abs(t - (T0 + it)).max()))
    assert float(np.abs(st["u"]).max()) > 1e-4


def test_tracer_content_changes_by_the_source_flux():
    """Tsrc differs from the ambient value: per step the content of the LtracerSrc tracer changes by
    dt * sum_k Huon(face, k) * Tsrc(k) over the source faces (signs by the side of the water cell), the fluxes being the
    ones step3d_t has just used (Huon / Hvom after step3d_uv)."""
    st, src = _river_state("UPWELLING", same=14.0)
    src.Tsrc[:, :, 0] = 20.0 + np.arange(st.b.N)[None, :] * 0.25           # a warm river, warmer towards the surface
    area = 1.0 / (st["pm"] * st["pn"])
    sl = _interior(st)
    be = oracle.Oracle(st)
    mo = main3d.Main3D(be)
    mo.initial()

    def content():
        return float((st["t"][sl][..., mo.s.nnew - 1, 0] * st["Hz"][sl] * area[sl][..., None]).sum())

    mo.run(2)
    for _ in range(5):
        c0 = content()
        mo.run(1)
        c1 = content()
        flux = 0.0
        sign = [+1.0, +1.0, -1.0]               # southern wall (in), western wall (in), eastern wall (out)
        for q, (i, j, d) in enumerate(zip(src.Isrc, src.Jsrc, src.Dsrc)):
            H = st["Huon" if int(d) == 0 else "Hvom"][st.I(i), st.J(j), :]
            flux += sign[q] * float((H * src.Tsrc[q, :, 0]).sum())
        assert abs((c1 - c0) - st.p.dt * flux) <= 2e-6 * abs(st.p.dt * flux), ((c1 - c0), st.p.dt * flux)


def test_source_table_rules():
    st, src = _river_state("UPWELLING", same=14.0)
    be = oracle.Oracle(st)
    s = util.step_idx()
    # Dsrc is 0, 1 or 2
    bad = type(src)(src.Isrc, src.Jsrc, [3.0] * src.n, src.Qbar, src.Qshape, src.Tsrc, src.LtracerSrc)
    with pytest.raises(RuntimeError):
        be.set_sources(bad)
    # LuvSrc / LwSrc without a table
    oracle.lib().oracle_set_sources(0, None, None, None, None, None, None, None, 0, 0)
    for flag in (1, 2, 3):
        st.p.point_sources = flag
        for k in ("step2d", "step3d_t", "omega"):
            with pytest.raises(RuntimeError):
                be.call(k, s)
