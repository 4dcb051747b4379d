"""Open boundary conditions on the southern / northern edges: Chapman implicit for the
free surface (zetabc.F:489, :638), Flather for the normal barotropic velocity (v2dbc_im.F:216, :565) with the
Chapman-type rule the reference gives the tangential one on such an edge (u2dbc_im.F:912, :1070), gradient, clamped
and implicit upstream radiation for every variable (zetabc.F:408, u2dbc_im.F:833, v2dbc_im.F:138, u3dbc_im.F:381,
v3dbc_im.F:97, t3dbc_im.F:364), selected per variable and side through roms_params_t.lbc as LBC(:,isFsur..isTvar,ng) does.

CPU (oracle): a free-surface bump in a flat channel leaves through Chapman / Flather edges and is kept by closed
walls; clamped edges hold the prescribed values.  GPU (-m gpu): the same runs and every boundary kernel, HIP
against the oracle."""
import math

import numpy as np
import pytest

import util
from roms_trunk_mgh_amd import abi, ana, main3d

H0 = 150.0
# the usual open-ocean set of roms_*.in: LBC(isFsur) = Cha, LBC(isUbar) = LBC(isVbar) = Fla, 3-D variables Rad
OPEN = {"zeta": "Cha", "ubar": "Fla", "vbar": "Fla", "u": "Rad", "v": "Rad", "t": "Rad"}
RADI = {v: "Rad" for v in OPEN}               # implicit upstream radiation for every variable
# explicit Chapman for the free surface with the Shchepetkin (Mason et al., 2010) condition for the barotropic velocity
CHE_SHC = {"zeta": "Che", "ubar": "Shc", "vbar": "Shc", "u": "Rad", "v": "Rad", "t": "Rad"}
GRAD = {v: "Gra" for v in OPEN}
CLAMP = {v: "Cla" for v in OPEN}


def set_lbc(st, table, sides=("south", "north")):
    for sd in sides:
        for var, code in table.items():
            st.p.lbc[abi.LBS[sd]][abi.LBV[var]] = abi.LBC[code]


def _bump_channel(table):
    """UPWELLING's channel made flat, f = 0, homogeneous, with a free-surface ridge along the middle of the
    channel: two gravity waves, one towards each of the S/N edges."""
    st = ana.make_tile("UPWELLING", perturb=0.0, overrides=dict(dt=100.0, ndtfast=20, theta_s=0.0, theta_b=0.0))
    st["h"][:] = H0
    st["f"][:] = 0.0
    st["fomn"][:] = 0.0
    st["t"][:, :, :, :, 0] = 14.0
    st["t"][:, :, :, :, 1] = 35.0
    for name in ("sustr", "svstr", "bustr", "bvstr", "stflx", "btflx", "srflx", "ubar", "vbar", "u", "v"):
        st[name][:] = 0.0
    b = st.b
    y = (np.arange(b.LBj, b.UBj + 1) - 0.5) / b.Mm
    z0 = 0.02 * np.exp(-((y - 0.5) / 0.08) ** 2)[None, :] * np.ones((st.ni, 1))
    for lev in range(3):
        st["zeta"][:, :, lev] = z0
    st["Zt_avg1"][:] = z0
    if table:
        set_lbc(st, table)
    return st


def _run(kind, st, nsteps):
    if kind == "hip":
        from roms_trunk_mgh_amd import hip
        be = hip.RomsHip(st)
    else:
        import oracle
        be = oracle.Oracle(st)
    m = main3d.Main3D(be)
    m.initial()
    m.run(nsteps)
    if kind == "hip":
        be.to_host()
        be.close()
    return m


BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("which", ["cha_fla_rad", "radiation", "che_shc_rad"])
def test_wave_leaves_through_open_edges(kind, which):
    """c = sqrt(g H) = 38 m/s: the two waves need 40 km / c = 1040 s to reach the edges; after 2400 s an open
    channel is (almost) at rest, a closed one still holds the energy."""
    nsteps = 24
    left = {}
    tables = {"cha_fla_rad": OPEN, "radiation": RADI, "che_shc_rad": CHE_SHC}
    for name, table in (("open", tables[which]), ("closed", None)):
        st = _bump_channel(table)
        e0 = float((st.interior("Zt_avg1") ** 2).sum())
        _run(kind, st, nsteps)
        left[name] = float((st.interior("Zt_avg1") ** 2).sum()) / e0
        assert np.isfinite(st["zeta"]).all() and np.isfinite(st["u"]).all()
    assert left["closed"] > 0.3, left
    # Chapman + Flather absorb the wave almost completely; radiation alone (no RADIATION_2D, and the free surface
    # of the southern edge differenced towards the boundary row, zetabc.F:424) lets about a quarter back
    # Chapman explicit + Shchepetkin absorb it as well
    assert left["open"] < {"cha_fla_rad": 0.02, "radiation": 0.35, "che_shc_rad": 0.05}[which] * left["closed"], left


@pytest.mark.parametrize("kind", BACKENDS)
def test_clamped_edges_hold_the_boundary_data(kind):
    st = _bump_channel(CLAMP)
    b = st.b
    rng = np.random.default_rng(3)
    st["zeta_bry"][:] = 1.0e-3 * rng.standard_normal(st["zeta_bry"].shape)
    st["ubar_bry"][:] = 1.0e-3 * rng.standard_normal(st["ubar_bry"].shape)
    st["vbar_bry"][:] = 1.0e-3 * rng.standard_normal(st["vbar_bry"].shape)
    st["u_bry"][:] = 1.0e-3 * rng.standard_normal(st["u_bry"].shape)
    st["v_bry"][:] = 1.0e-3 * rng.standard_normal(st["v_bry"].shape)
    st["t_bry"][:] = 14.0 + rng.standard_normal(st["t_bry"].shape)
    m = _run(kind, st, 3)
    s = m.s
    I = st.I(b.Istr, b.Iend)
    for jb in (b.Jstr - 1, b.Jend + 1):          # rho-type boundary rows
        J = st.J(jb)
        assert np.array_equal(st["t"][I, J, :, s.nnew - 1, :], st["t_bry"][I, J])
    # (u and v on their boundary rows are the clamped values with the vertical mean replaced by the barotropic
    # one, step3d_uv.F:1126-1180, 1344-1400: their deviation from the vertical mean is the data's)
    Hzu = 0.5 * (st["Hz"][1:, :, :] + st["Hz"][:-1, :, :])
    for jb in (b.Jstr - 1, b.Jend + 1):
        J = st.J(jb)
        Iu = st.I(b.IstrU, b.Iend)
        w = Hzu[Iu.start - 1:Iu.stop - 1, J, :]
        dev = lambda a: a - (a * w).sum(-1, keepdims=True) / w.sum(-1, keepdims=True)
        assert np.allclose(dev(st["u"][Iu, J, :, s.nnew - 1]), dev(st["u_bry"][Iu, J]), rtol=0, atol=1e-15)


# -------------------------------------------------------------------------------------------- GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("config", ["BENCHMARK_TINY", "UPWELLING", "SEAMOUNT"])
@pytest.mark.parametrize("table", [OPEN, RADI, GRAD, CLAMP, CHE_SHC], ids=["cha_fla_rad", "radiation", "gradient", "clamped", "che_shc_rad"])
@pytest.mark.parametrize("kernel", ["step2d", "step3d_uv", "step3d_t", "pre_step3d"])
def test_hip_kernels_with_open_edges(config, table, kernel):
    import oracle
    from roms_trunk_mgh_amd import hip
    st0 = util.prepared_state(config)
    set_lbc(st0, table)
    rng = np.random.default_rng(5)
    for name in ("zeta_bry", "ubar_bry", "vbar_bry", "u_bry", "v_bry"):
        st0[name][:] = 1.0e-2 * rng.standard_normal(st0[name].shape)
    st0["t_bry"][:] = st0["t"][:, :, :, 0, :] * (1.0 + 1.0e-3 * rng.standard_normal(st0["t_bry"].shape))
    if kernel == "step3d_t":
        util.hz_weighted_tnew(st0)
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


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["BENCHMARK_TINY", "UPWELLING"])
def test_hip_100_steps_with_open_edges(config):
    import oracle
    from roms_trunk_mgh_amd import hip
    from roms_trunk_mgh_amd.state import rel_rms
    st_o = ana.make_tile(config, perturb=1.0)
    set_lbc(st_o, OPEN)
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
