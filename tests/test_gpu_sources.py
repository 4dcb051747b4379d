"""-m gpu: point sources through u- and v-faces (LuvSrc, rivers).  The HIP library against the CPU oracle through the C
ABI (roms_hip_set_sources), kernel by kernel and over whole runs.  The oracle's source blocks are parity-unpinned (the
reference's routines USE mod_sources -> netCDF); its known answers are in tests/test_sources.py, and the first two of
them are repeated here on the device."""
import numpy as np
import pytest

import test_sources as ts
import util
from roms_trunk_mgh_amd import ana, hip, main3d
from roms_trunk_mgh_amd.state import rel_rms

pytestmark = pytest.mark.gpu

VARIANTS = {
    "basin": dict(config="UPWELLING", overrides={"EWperiodic": False}, mask=None, kind="walls"),
    "channel": dict(config="UPWELLING", overrides={}, mask=None, kind="walls"),
    "coast": dict(config="UPWELLING", overrides={"EWperiodic": False}, mask="island", kind="both"),
    "benchmark": dict(config="BENCHMARK_TINY", overrides={}, mask="island", kind="both"),
    "seamount": dict(config="SEAMOUNT", overrides={"EWperiodic": False}, mask=None, kind="walls"),
    "mpdata": dict(config="UPWELLING", overrides={"EWperiodic": False, "Hadv": "MPDATA", "Vadv": "MPDATA"}, mask="island",
                   kind="both"),
    "hsimt": dict(config="UPWELLING", overrides={"EWperiodic": False, "Hadv": "HSIMT", "Vadv": "HSIMT"}, mask=None,
                  kind="walls"),
    "n40": dict(config="UPWELLING", overrides={"EWperiodic": False, "N": 40}, mask=None, kind="walls"),
    # cell-centred sources (LwSrc): alone, beside the face sources, two in one cell
    "wells": dict(config="UPWELLING", overrides={"EWperiodic": False}, mask=None, kind="wells"),
    "all": dict(config="UPWELLING", overrides={"EWperiodic": False}, mask="island", kind="all"),
    "wells_dup": dict(config="BENCHMARK_TINY", overrides={}, mask=None, kind="wells_dup"),
    "mpdata_all": dict(config="UPWELLING", overrides={"EWperiodic": False, "Hadv": "MPDATA", "Vadv": "MPDATA"},
                       mask="island", kind="all"),
    "hsimt_all": dict(config="UPWELLING", overrides={"EWperiodic": False, "Hadv": "HSIMT", "Vadv": "HSIMT"}, mask=None,
                      kind="all"),
    # with WET_DRY: the general barotropic sequence carries both, the output masks count source faces as water
    "wet_all": dict(config="UPWELLING", overrides={"EWperiodic": False, "wet_dry": 1, "beach": 1, "zeta_amp": 0.3},
                    mask="island", kind="all"),
}


def _prepared(variant):
    v = VARIANTS[variant]
    st = util.prepared_state(v["config"], overrides=v["overrides"], mask=v["mask"], wet=bool(v["overrides"].get("wet_dry")))
    src = util.river_sources(st, v["kind"])
    # the mass fluxes of the source faces as step3d_uv leaves them (prepared_state has walls and coasts at rest)
    q = src.qsrc()
    for n, (i, j, d) in enumerate(zip(src.Isrc, src.Jsrc, src.Dsrc)):
        if int(d) < 2:
            st["Huon" if int(d) == 0 else "Hvom"][st.I(i), st.J(j), :] = q[n]
    return st


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("kernel", ["step2d", "step3d_uv", "pre_step3d", "step3d_t", "rhs3d", "omega", "wetdry"])
def test_source_kernels(variant, kernel):
    import oracle
    st0 = _prepared(variant)
    if kernel == "step3d_t":
        util.hz_weighted_tnew(st0)
    luv, lw = bool(st0.p.point_sources & 1), bool(st0.p.point_sources & 2)
    if (kernel == "omega" and not lw) or (kernel in ("step3d_uv", "pre_step3d", "rhs3d") and not luv) or \
            (kernel == "wetdry" and not (luv and st0.p.wet_dry)):
        pytest.skip("the kernel has no block for this kind of source")
    st_o, st_h, st_n = st0.copy(), st0.copy(), st0.copy()
    preds = [(5, 1, 0)] if kernel != "step2d" else [(5, 1, 1), (5, 2, 1), (5, 2, 0)]
    be_o = oracle.Oracle(st_o)
    h = hip.RomsHip(st_h)
    try:
        for iic, iif, pred in preds:
            s = util.step_idx(iic=iic, iif=iif, pred=pred, knew=3 if pred else 2, krhs=1 if pred else 3)
            be_o.call(kernel, s)
            h.call(kernel, s)
        h.to_host()
    finally:
        h.close()
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-13 for v in diffs.values()), diffs
    # the sources act: the same call without them gives another state
    st_n.p = type(st0.p).from_buffer_copy(st0.p)
    st_n.p.point_sources = 0
    st_n.sources = None
    be_n = oracle.Oracle(st_n)
    for iic, iif, pred in preds:
        be_n.call(kernel, util.step_idx(iic=iic, iif=iif, pred=pred, knew=3 if pred else 2, krhs=1 if pred else 3))
    assert util.compare_states(st_n, st_o), "sources without effect"


@pytest.mark.parametrize("variant,physics", [("basin", False), ("channel", False), ("coast", False), ("benchmark", True),
                                             ("seamount", False), ("mpdata", False), ("hsimt", False), ("n40", False),
                                             ("wells", False), ("all", False), ("wells_dup", True), ("mpdata_all", False),
                                             ("hsimt_all", False), ("wet_all", False)])
def test_100_steps_with_rivers(variant, physics):
    import oracle
    v = VARIANTS[variant]
    st_o = ana.make_tile(v["config"], perturb=1.0 if v["config"] != "SEAMOUNT" else 0.0, overrides=v["overrides"], mask=v["mask"])
    util.river_sources(st_o, v["kind"])
    st_h = st_o.copy()
    mo = main3d.Main3D(oracle.Oracle(st_o), physics=physics, diagnostics=physics)
    # (N = 40 thins the layers of the shallow shelf until this river's vertical velocity is past the stability limit
    # of the explicit scheme after ~25 steps, in the oracle as on the device: 20 steps there)
    nsteps = 20 if variant == "n40" else 100
    mo.initial()
    mo.run(nsteps)
    be = hip.RomsHip(st_h)
    try:
        mh = main3d.Main3D(be, physics=physics, diagnostics=physics)
        mh.initial()
        mh.run(nsteps)
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


@pytest.mark.parametrize("config,overrides,mask,kind", [("UPWELLING", {}, None, "walls"), ("UPWELLING", {}, "island", "both"),
                                                        ("BENCHMARK_TINY", {}, "island", "both"),
                                                        ("UPWELLING", {"Hadv": "MPDATA", "Vadv": "MPDATA"}, "island", "both"),
                                                        ("UPWELLING", {}, None, "wells"), ("UPWELLING", {}, "island", "all"),
                                                        ("UPWELLING", {"Hadv": "MPDATA", "Vadv": "MPDATA"}, None, "all")])
def test_device_river_of_ambient_water_keeps_tracers_uniform(config, overrides, mask, kind):
    """the known answers of tests/test_sources.py on the device: uniform tracers stay uniform, the basin's volume grows
    by the net discharge"""
    T0 = 14.0
    st, src = ts._river_state(config, same=T0, mask=mask, kind=kind, basin=True, overrides=overrides)
    be = hip.RomsHip(st)
    try:
        mh = main3d.Main3D(be)
        mh.initial()
        mh.run(2)
        be.to_host()
        v0 = ts._volume(st)
        mh.run(28)
        be.to_host()
        v1 = ts._volume(st)
    finally:
        be.close()
    sl = ts._interior(st)
    wet = st["rmask"][sl] == 1.0 if mask else np.ones(st["h"][sl].shape, bool)
    for it in range(st.b.NT):
        t = st["t"][sl][..., mh.s.nnew - 1, it][wet]
        assert float(np.abs(t - (T0 + it)).max()) < 2e-11 * (T0 + it)
    if kind in ("walls", "wells") or (kind == "all" and not mask):
        qnet = 0.0
        for i, d, q in zip(src.Isrc, src.Dsrc, src.Qbar):
            qnet += q if int(d) == 2 else (-q if i == st.b.Lm + 1 else q)
        assert abs((v1 - v0) / (qnet * st.p.dt * 28) - 1.0) < 1e-7


def test_river_discharge_changes_between_steps():
    """set_data refreshes Qbar / Qsrc / Tsrc every step (set_data.F:124-160): roms_hip_set_sources again with new
    values (the same faces: the device table is updated in place, captured LOOP_2D graphs stay valid)"""
    import oracle
    st_o = ana.make_tile("UPWELLING", perturb=1.0, overrides={"EWperiodic": False})
    src = util.river_sources(st_o, "walls")
    st_h = st_o.copy()
    q0 = src.Qbar.copy()
    be_o = oracle.Oracle(st_o)
    be_h = hip.RomsHip(st_h)
    try:
        mo, mh = main3d.Main3D(be_o), main3d.Main3D(be_h)
        mo.initial()
        mh.initial()
        for n in range(12):
            src.Qbar[:] = q0 * (1.0 + 0.1 * n)
            src.Tsrc[:, :, 0] += 0.05
            be_o.set_sources(src)
            be_h.set_sources(src)
            mo.run(1)
            mh.run(1)
        be_h.to_host()
    finally:
        be_h.close()
    diffs = util.compare_states(st_h, st_o)
    assert all(v <= 1e-11 for v in diffs.values()), diffs
