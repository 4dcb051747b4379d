"""Known-answer tests that do NOT come from reading the reference's Fortran: analytic solutions of the
equations the path discretises, run through the whole step (main3d sequencing) on a flat-bottom, uniform-
density periodic channel.  They are independent evidence for the routines whose oracle is parity-unpinned
against reference output (step2d, pre_step3d, rhs3d_tile, step3d_uv, step3d_t, omega) -- a
misread coefficient in the C restatement AND in the HIP kernel (written from the same Fortran) would pass
every HIP-vs-oracle test and fail here.

* gravity wave: a standing free-surface wave must oscillate with the C-grid phase speed
  omega = (2/dx) sin(k dx/2) sqrt(g' H), g' = g (1 + rho'/rho0) for the uniform density anomaly rho' = rho - 1000
  (the slow-time force is the 3-D pressure gradient, whose surface pressure is g zeta + (g/rho0) rho' (zeta - z),
  prsgrd32.h:264-269), and keep its amplitude -- constrains g, the total depth, pm/pn,
  dtfast and the LF-AM3 predictor/corrector weights of step2d_LF_AM3.h:770-851, 939-1019, 2098-2255 and the
  2-D/3-D coupling through the fast-time averages (set_weights.F, step3d_uv.F:997-1190).
* geostrophic jet: u = -(g'/f) d(zeta)/dy is a steady state -- constrains sign and size of the Coriolis
  term against the pressure gradient in step2d (:1291-1325) and rhs3d.F:467-505, and the coupling.
* momentum diffusion: a cosine mode of u(z) with no stress at top and bottom decays as
  exp(-Akv m^2 t) -- constrains the spline-form implicit operator of step3d_uv.F:346-400
  (FC/CF/BC with Hz/6, Hz/3 and dt*Akv/Hz).
* tracer wave in a uniform current: translated at U, damped at the rate of the third-order upstream scheme's own
  symbol, damping ~ k^4 -- constrains the U3 fluxes and time stepping of pre_step3d.F / step3d_t.F and the mass
  fluxes of step3d_uv.F.
Each test runs on the oracle (CPU) and, with -m gpu, on the HIP path."""
import math

import numpy as np
import pytest

from roms_trunk_mgh_amd import ana, main3d

G = 9.81
H0 = 150.0


def _channel(dt, ndtfast, f0=0.0, akv=1.0e-5):
    """UPWELLING's Cartesian periodic channel (dx = dy = 1 km, 41 x 80 x 16) made flat and homogeneous."""
    st = ana.make_tile("UPWELLING", perturb=0.0, overrides=dict(dt=dt, ndtfast=ndtfast, theta_s=0.0, theta_b=0.0))
    st["h"][:] = H0
    st["f"][:] = f0
    st["fomn"][:] = f0 / (st["pm"] * st["pn"])
    st["t"][:, :, :, :, 0] = 14.0
    st["t"][:, :, :, :, 1] = 35.0
    for name in ("sustr", "svstr", "bustr", "bvstr", "stflx", "btflx", "srflx"):
        st[name][:] = 0.0
    st["Akv"][:] = akv
    st["Akt"][:] = 1.0e-6
    for name in ("zeta", "ubar", "vbar", "u", "v", "Zt_avg1"):
        st[name][:] = 0.0
    return st


def _backend(kind, st):
    if kind == "hip":
        from roms_trunk_mgh_amd import hip
        return hip.RomsHip(st)
    import oracle
    return oracle.Oracle(st)


BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("kind", BACKENDS)
def test_gravity_wave_phase_speed(kind):
    dt, ndtfast = 20.0, 20
    st = _channel(dt, ndtfast)
    b = st.b
    dx = 1.0 / float(st["pm"][3, 3])
    L = b.Lm * dx
    kx = 2.0 * math.pi / L
    amp = 1.0e-3
    x = (np.arange(b.LBi, b.UBi + 1) - 0.5) * dx
    z0 = amp * np.cos(kx * x)[:, None] * np.ones((1, st.nj))
    for lev in range(3):
        st["zeta"][:, :, lev] = z0
    st["Zt_avg1"][:] = z0
    rho_anom = st.p.R0 - st.p.R0 * st.p.Tcoef * (14.0 - st.p.T0) - 1000.0      # linear EOS, rho_eos.F:700-716
    omega = (2.0 / dx) * math.sin(0.5 * kx * dx) * math.sqrt(G * (1.0 + rho_anom / st.p.rho0) * H0)
    period = 2.0 * math.pi / omega
    nsteps = int(round(2.25 * period / dt))
    be = _backend(kind, st)
    m = main3d.Main3D(be)
    m.initial()
    i0, j0 = 1 - b.LBi, b.Mm // 2 - b.LBj          # antinode of the standing wave
    series = []
    for _ in range(nsteps):
        m.step()
        if kind == "hip":
            be.to_host(["Zt_avg1", "vbar"])
        series.append(float(st["Zt_avg1"][i0, j0]))
    if kind == "hip":
        be.close()
    series = np.array(series)
    t = dt * np.arange(1, nsteps + 1)
    # Zt_avg1 after step n is the fast-time average centred on t_n: least-squares fit of
    # A cos(w t) + B sin(w t) over a scan of w
    a0 = float(z0[i0, j0])
    best = None
    for w in omega * np.linspace(0.95, 1.05, 2001):
        A = np.stack([np.cos(w * t), np.sin(w * t)], 1)
        c = np.linalg.lstsq(A, series, rcond=None)[0]
        r = float(np.sum((A @ c - series) ** 2))
        if best is None or r < best[0]:
            best = (r, w, c)
    _, w_fit, c = best
    assert abs(w_fit / omega - 1.0) < 2.0e-3, (w_fit, omega)
    # amplitude kept (no growth, little damping by the fast-time filter), phase of a wave released at rest,
    # no cross-channel flow
    assert 0.97 < math.hypot(*c) / a0 < 1.005, math.hypot(*c) / a0
    assert abs(math.atan2(c[1], c[0])) < 5.0e-3
    assert float(np.abs(st["vbar"]).max()) < 1e-12


@pytest.mark.parametrize("kind", BACKENDS)
def test_geostrophic_jet_is_steady(kind):
    f0 = -8.26e-5
    dt, ndtfast = 300.0, 30
    st = _channel(dt, ndtfast, f0=f0)
    b = st.b
    dy = 1.0 / float(st["pn"][3, 3])
    Ly = b.Mm * dy
    ly = math.pi / Ly
    amp = 0.05
    y = (np.arange(b.LBj, b.UBj + 1) - 0.5) * dy
    zy = amp * np.cos(ly * y)
    zy[0], zy[-1] = zy[1], zy[-2]                         # closed walls: zero gradient (zetabc)
    z0 = np.ones((st.ni, 1)) * zy[None, :]
    rho_anom = st.p.R0 - st.p.R0 * st.p.Tcoef * (14.0 - st.p.T0) - 1000.0
    g_eff = G * (1.0 + rho_anom / st.p.rho0)             # see the module docstring (prsgrd32.h:264-269)
    uy = (g_eff / f0) * amp * ly * np.sin(ly * y)        # u = -(g'/f) d(zeta)/dy at the u-points (same y)
    u0 = np.ones((st.ni, 1)) * uy[None, :]
    for lev in range(3):
        st["zeta"][:, :, lev] = z0
        st["ubar"][:, :, lev] = u0
    st["Zt_avg1"][:] = z0
    st["u"][:] = u0[:, :, None, None]
    be = _backend(kind, st)
    m = main3d.Main3D(be)
    m.initial()
    m.run(20)
    if kind == "hip":
        be.to_host()
        be.close()
    umax = float(np.abs(uy).max())
    ui = st.interior("u")[..., m.s.nnew - 1]
    want = u0[st.I(b.Istr, b.Iend), st.J(b.Jstr, b.Jend)][:, :, None]
    assert float(np.abs(ui - want).max()) < 2.0e-3 * umax, float(np.abs(ui - want).max()) / umax
    assert float(np.abs(st.interior("v")).max()) < 2.0e-3 * umax
    zi = st.interior("Zt_avg1")
    assert float(np.abs(zi - z0[st.I(b.Istr, b.Iend), st.J(b.Jstr, b.Jend)]).max()) < 2.0e-3 * amp


@pytest.mark.parametrize("kind", BACKENDS)
def test_vertical_momentum_diffusion_mode(kind):
    akv = 1.0e-2
    dt, ndtfast = 300.0, 30
    st = _channel(dt, ndtfast, akv=akv)
    b = st.b
    be = _backend(kind, st)
    m = main3d.Main3D(be)
    m.initial()                                            # z_r of the flat channel
    if kind == "hip":
        be.to_host(["z_r"])
    mz = math.pi / H0
    prof = 0.1 * np.cos(mz * (st["z_r"] + H0))             # zero vertical mean, zero stress at top and bottom
    st["u"][:] = prof[:, :, :, None]
    if kind == "hip":
        be.to_device(["u"])
    nsteps = 60
    m.run(nsteps)
    if kind == "hip":
        be.to_host()
        be.close()
    ui = st.interior("u")[..., m.s.nnew - 1]
    p0 = prof[st.I(b.Istr, b.Iend), st.J(b.Jstr, b.Jend)]
    ratio = float(np.sum(ui * p0) / np.sum(p0 * p0))      # projection on the mode
    rate = -math.log(ratio) / (nsteps * dt)
    assert abs(rate / (akv * mz * mz) - 1.0) < 2.0e-2, (rate, akv * mz * mz)
    # the shape is preserved (it is an eigenmode), nothing leaks into v or the free surface
    assert float(np.abs(ui - ratio * p0).max()) < 2.0e-3 * 0.1
    assert float(np.abs(st.interior("v")).max()) < 1e-10


def _u3_symbol(theta):
    """Eigenvalue (times dx/U) of the third-order upstream-biased flux-form advection operator for the mode
    exp(i k x), U > 0: face value (-S[i-1] + 5 S[i] + 2 S[i+1]) / 6 = centred average minus one sixth of the
    upstream curvature (Shchepetkin & McWilliams 1998, the scheme TS_U3HADVECTION names)."""
    e = complex(math.cos(theta), math.sin(theta))
    face = 0.5 * (1.0 + e) - (e - 2.0 + 1.0 / e) / 6.0
    return -(1.0 - 1.0 / e) * face


@pytest.mark.parametrize("kind", BACKENDS)
def test_tracer_wave_in_uniform_flow_u3(kind):
    """A passive tracer wave S = S0 + A cos(k x) in a uniform zonal current U (flat channel, f = 0, no stress: a
    steady state of the momentum equations) is translated at speed U and, with third-order upstream advection,
    damped at the rate of the scheme's own symbol, Re(lambda) = -(U/dx) theta^4/12 (1 + O(theta^2)) -- constrains
    the 1/6 curvature weights and the upstream selection of the U3 fluxes in pre_step3d.F:364-420 and
    step3d_t.F:596-700, the LF-TR/AB3 time-stepping weights (their error enters at O((U dt/dx)^2) of this), and
    the mass fluxes Huon handed from step3d_uv to the tracers (a wrong flux moves the wave at another speed).
    Two wavelengths give the order: damping over a fixed time scales with k^4."""
    U, A = 0.5, 1.0
    dt, ndtfast, nsteps = 100.0, 20, 120
    out = {}
    for mode in (1, 2):
        st = _channel(dt, ndtfast)
        st.p.Scoef = 0.0                                   # salinity does not enter the density: passive
        b = st.b
        dx = 1.0 / float(st["pm"][3, 3])
        theta = 2.0 * math.pi * mode / b.Lm
        xr = (np.arange(b.LBi, b.UBi + 1) - 0.5) * dx     # rho-points
        k = theta / dx
        S = 35.0 + A * np.cos(k * xr)
        st["t"][:, :, :, :, 1] = S[:, None, None, None]
        st["u"][:] = U
        st["ubar"][:] = U
        be = _backend(kind, st)
        m = main3d.Main3D(be)
        m.initial()
        m.run(nsteps)
        if kind == "hip":
            be.to_host()
            be.close()
        T = nsteps * dt
        Si = st.interior("t")[:, :, :, m.s.nnew - 1, 1]
        xi = xr[st.I(b.Istr, b.Iend)]
        # nothing else moved: the current is steady, the wave stays independent of y and z
        assert float(np.abs(st.interior("u")[..., m.s.nnew - 1] - U).max()) < 1e-9
        assert float(np.abs(Si - Si[:, :1, :1]).max()) < 1e-9
        s1 = Si[:, 3, 5] - 35.0
        c = 2.0 * np.mean(s1 * np.cos(k * xi)), 2.0 * np.mean(s1 * np.sin(k * xi))
        amp, phase = math.hypot(*c), math.atan2(c[1], c[0])
        lam = _u3_symbol(theta) * U / dx
        out[mode] = (-math.log(amp / A) / T, phase / (k * T), lam)
        # translated at the phase speed of the scheme (= U up to its fourth-order dispersion)
        assert abs(phase / T / (-lam.imag) - 1.0) < 1.0e-3, (phase / T, -lam.imag)
        assert abs(phase / (k * T) / U - 1.0) < 1.0e-3
        # damped at the rate of the third-order scheme
        assert abs(out[mode][0] / (-lam.real) - 1.0) < 3.0e-2, (out[mode][0], -lam.real)
    order = math.log2(out[2][0] / out[1][0])
    assert 3.8 < order < 4.05, (order, out)
