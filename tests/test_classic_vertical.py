"""Without SPLINES_VVISC / SPLINES_VDIFF (3 of the reference's 31 three-dimensional applications): the implicit vertical
viscosity / diffusion as tridiagonal systems for u, v and the tracers themselves (step3d_uv.F:400-464, :733-797,
step3d_t.F:1431-1501).  PARITY UNPINNED like the rest of step3d_uv / step3d_t (mod_sources -> netCDF).  Known answers of
the operator on the oracle; the HIP library against the oracle in tests/test_gpu_classic_vertical.py."""
import numpy as np
import pytest

import oracle
import test_sources as ts
import util
from roms_trunk_mgh_amd import ana, main3d

CLASSIC = {"splines_vdiff": 0, "splines_vvisc": 0}


def test_vertical_diffusion_conserves_the_column_and_smooths_it():
    """no flow, no surface or bottom flux: step3d_t is the implicit vertical diffusion alone -- the thickness-weighted
    column sum of every tracer is unchanged, its extremes do not grow and a step in the profile is smoothed"""
    st = util.prepared_state("UPWELLING", overrides=dict(CLASSIC, EWperiodic=False))
    for n in ("Huon", "Hvom", "W", "stflx", "btflx", "srflx"):
        st[n][:] = 0.0
    N = st.b.N
    st["Akt"][:] = 5.0e-3
    prof = np.where(np.arange(N) < N // 2, 10.0, 14.0)              # a step at mid-depth
    for it in range(st.b.NT):
        st["t"][:, :, :, 2, it] = prof[None, None, :] + it
        st["t"][:, :, :, 1, it] = (prof[None, None, :] + it) * st["Hz"]          # t(nnew) as pre_step3d leaves it
    sl = ts._interior(st)
    before = (st["t"][sl][..., 1, 0]).sum(axis=2)                   # sum_k Hz t
    oracle.Oracle(st).call("step3d_t", util.step_idx(iic=5, nstp=1, nnew=2, nrhs=1))
    t = st["t"][sl][..., 1, 0]
    after = (t * st["Hz"][sl]).sum(axis=2)
    assert float(np.abs(after - before).max()) <= 1e-12 * float(np.abs(before).max())
    assert t.min() >= 10.0 - 1e-12 and t.max() <= 14.0 + 1e-12
    assert float(np.abs(np.diff(t, axis=2)).max()) < 4.0 - 1e-3    # the step is smoothed


@pytest.mark.parametrize("config,overrides,mask,kind", [("UPWELLING", {}, None, "all"), ("UPWELLING", {}, "island", "all"),
                                                        ("SEAMOUNT", {}, None, "walls"),
                                                        ("UPWELLING", {"Hadv": "MPDATA", "Vadv": "MPDATA"}, None, "all"),
                                                        ("UPWELLING", {"Hadv": "C4", "Vadv": "SPLINES"}, None, "walls")])
def test_uniform_tracers_stay_uniform_without_the_spline_operators(config, overrides, mask, kind):
    """rivers and wells of ambient water: the constancy test of tests/test_sources.py with both switches off (the
    thickness-weighted tracer through the vertical advection, the sources' term without 1/Hz, step3d_t.F:1341-1343)"""
    T0 = 14.0
    st, _ = ts._river_state(config, same=T0, mask=mask, kind=kind, basin=True, overrides=dict(overrides, **CLASSIC))
    assert st.p.splines_vdiff == 0 and st.p.splines_vvisc == 0
    mo = main3d.Main3D(oracle.Oracle(st))
    mo.initial()
    mo.run(30)
    sl = ts._interior(st)
    wet = st["rmask"][sl] == 1.0 if mask else np.ones(st["h"][sl].shape, bool)
    for it in range(st.b.NT):
        t = st["t"][sl][..., mo.s.nnew - 1, it][wet]
        assert float(np.abs(t - (T0 + it)).max()) < 2e-11 * (T0 + it)
    assert float(np.abs(st["u"]).max()) > 1e-4


def test_classic_and_spline_operators_agree_to_discretisation_error():
    """100 steps of UPWELLING with either form: the same flow to a per cent or two, not the same bits"""
    res = {}
    for spl in (1, 0):
        st = ana.make_tile("UPWELLING", perturb=1.0, overrides={"splines_vdiff": spl, "splines_vvisc": spl})
        mo = main3d.Main3D(oracle.Oracle(st))
        mo.initial()
        mo.run(100)
        assert np.isfinite(st["t"]).all()
        res[spl] = st
    du = util.max_rel_diff(res[0]["u"], res[1]["u"])
    dtr = util.max_rel_diff(res[0]["t"], res[1]["t"])
    assert 1e-6 < du < 0.05 and 1e-8 < dtr < 5e-3, (du, dtr)
