"""Synthetic (analytic) problem set-up for the configurations BASELINE.json
names: BENCHMARK1/2/3, UPWELLING, SEAMOUNT.  Host-side, runs once -- this is
the reference's `initial` phase (ROMS/Nonlinear/initial.F:273-571), restated in
numpy only far enough to give the hot path realistic inputs:

  grid      ROMS/Functionals/ana_grid.h   (BENCHMARK :243-248,459-479,674-689,
            867-874,920-926; UPWELLING :384-392,1047-1070; SEAMOUNT :345-350,
            1021-1028) and ROMS/Utility/metrics.F
  s-coord   ROMS/Utility/set_scoord.F (Vstretching 4)
  weights   ROMS/Utility/set_weights.F:3-244 (POWER_LAW)
  IC        ROMS/Functionals/ana_initial.h:523-536, 787-794, 806-825

The per-step physics *outside* the hot path (bulk_flux, lmd_vmix, set_vbc --
SURVEY.md section 8f-1) is replaced by fixed analytic forcing/mixing fields
that are inputs to BOTH the oracle and the HIP path.
"""
import math

import numpy as np

from . import abi
from .bounds import make_bounds
from .state import TileState

# ---------------------------------------------------------------------------
# configuration table (ROMS/External/roms_*.in)
# ---------------------------------------------------------------------------
CONFIGS = {
    # name: Lm, Mm, N, NAT, dt, ndtfast, tnu2, visc2, gamma2, theta_s, theta_b,
    #       Tcline, Hadv, Vadv, app
    "BENCHMARK1": dict(Lm=512, Mm=64, N=30, NAT=2, dt=150.0, ndtfast=20, tnu2=500.0,
                       visc2=5000.0, gamma2=1.0, theta_s=0.0, theta_b=0.0, Tcline=400.0,
                       Hadv="U3", Vadv="C4", app="BENCHMARK"),
    "BENCHMARK2": dict(Lm=1024, Mm=128, N=30, NAT=2, dt=150.0, ndtfast=20, tnu2=500.0,
                       visc2=5000.0, gamma2=1.0, theta_s=0.0, theta_b=0.0, Tcline=400.0,
                       Hadv="U3", Vadv="C4", app="BENCHMARK"),
    "BENCHMARK3": dict(Lm=2048, Mm=256, N=30, NAT=2, dt=150.0, ndtfast=20, tnu2=500.0,
                       visc2=5000.0, gamma2=1.0, theta_s=0.0, theta_b=0.0, Tcline=400.0,
                       Hadv="U3", Vadv="C4", app="BENCHMARK"),
    # shrunken BENCHMARK for CPU-sized parity tests
    "BENCHMARK_TINY": dict(Lm=64, Mm=32, N=30, NAT=2, dt=150.0, ndtfast=20, tnu2=500.0,
                           visc2=5000.0, gamma2=1.0, theta_s=0.0, theta_b=0.0, Tcline=400.0,
                           Hadv="U3", Vadv="C4", app="BENCHMARK"),
    "UPWELLING": dict(Lm=41, Mm=80, N=16, NAT=2, dt=300.0, ndtfast=30, tnu2=0.0,
                      visc2=5.0, gamma2=1.0, theta_s=3.0, theta_b=0.0, Tcline=25.0,
                      Hadv="U3", Vadv="C4", app="UPWELLING"),
    "SEAMOUNT": dict(Lm=49, Mm=48, N=13, NAT=1, dt=60.0, ndtfast=20, tnu2=0.0,
                     visc2=0.0, gamma2=-1.0, theta_s=6.5, theta_b=2.0, Tcline=100.0,
                     Hadv="A4", Vadv="A4", app="SEAMOUNT"),
}

G = 9.81            # mod_scalars.F:431-441
RHO0 = 1025.0
ERADIUS = 6371315.0
DEG2RAD = math.pi / 180.0


def set_weights(ndtfast):
    """ROMS/Utility/set_weights.F:3-244, POWER_LAW filter.  The reference sums in
    real(r16); numpy longdouble is used here (differences <= 1 ulp of double)."""
    LD = np.longdouble
    Falpha, Fbeta, Fgamma = 2.0, 4.0, 0.284     # mod_scalars.F:310-312
    n2 = 2 * ndtfast
    w1 = np.zeros(n2 + 2, dtype=np.float64)     # 1-based
    w2 = np.zeros(n2 + 2, dtype=np.float64)
    nfast = 0
    scale = (Falpha + 1.0) * (Falpha + Fbeta + 1.0) / (
        (Falpha + 2.0) * (Falpha + Fbeta + 2.0) * float(ndtfast))
    gamma = Fgamma * max(0.0, 1.0 - 10.0 / float(ndtfast))
    for _ in range(16):
        nfast = 0
        for i in range(1, n2 + 1):
            cff = LD(scale) * LD(i)
            w1[i] = float(cff ** LD(Falpha) - cff ** LD(Falpha + Fbeta) - LD(gamma) * cff)
            if w1[i] > 0.0:
                nfast = i
            if nfast > 0 and w1[i] < 0.0:
                w1[i] = 0.0
        wsum = LD(0)
        shift = LD(0)
        for i in range(1, nfast + 1):
            wsum += LD(w1[i])
            shift += LD(w1[i]) * LD(i)
        scale = float(LD(scale) * shift / (wsum * LD(ndtfast)))
    for _ in range(ndtfast):
        wsum = LD(0)
        shift = LD(0)
        for i in range(1, nfast + 1):
            wsum += LD(w1[i])
            shift += LD(i) * LD(w1[i])
        shift = shift / wsum
        cff = LD(ndtfast) - shift
        if cff > 1:
            nfast += 1
            for i in range(nfast, 1, -1):
                w1[i] = w1[i - 1]
            w1[1] = 0.0
        elif cff > 0:
            wsum = LD(1) - cff
            for i in range(nfast, 1, -1):
                w1[i] = float(wsum * LD(w1[i]) + cff * LD(w1[i - 1]))
            w1[1] = float(wsum * LD(w1[1]))
        elif cff < -1:
            nfast -= 1
            for i in range(1, nfast + 1):
                w1[i] = w1[i + 1]
            w1[nfast + 1] = 0.0
        elif cff < 0:
            wsum = LD(1) + cff
            for i in range(1, nfast):
                w1[i] = float(wsum * LD(w1[i]) - cff * LD(w1[i + 1]))
            w1[nfast] = float(wsum * LD(w1[nfast]))
    for j in range(1, nfast + 1):
        cff = w1[j]
        for i in range(1, j + 1):
            w2[i] = w2[i] + cff
    wsum = LD(0)
    cff = LD(0)
    for i in range(1, nfast + 1):
        wsum += LD(w1[i])
        cff += LD(w2[i])
    wsum = LD(1) / wsum
    cff = LD(1) / cff
    for i in range(1, nfast + 1):
        w1[i] = float(wsum * LD(w1[i]))
        w2[i] = float(cff * LD(w2[i]))
    return nfast, w1[1:n2 + 1].copy(), w2[1:n2 + 1].copy()


def set_scoord(N, theta_s, theta_b):
    """ROMS/Utility/set_scoord.F, Vstretching == 4 branch."""
    sc_w = np.zeros(N + 1)
    Cs_w = np.zeros(N + 1)
    sc_r = np.zeros(N + 1)      # index 1..N used
    Cs_r = np.zeros(N + 1)
    ds = 1.0 / float(N)

    def C(s):
        if theta_s > 0.0:
            Csur = (1.0 - math.cosh(theta_s * s)) / (math.cosh(theta_s) - 1.0)
        else:
            Csur = -s ** 2
        if theta_b > 0.0:
            return (math.exp(theta_b * Csur) - 1.0) / (1.0 - math.exp(-theta_b))
        return Csur

    sc_w[N] = 0.0
    Cs_w[N] = 0.0
    for k in range(N - 1, 0, -1):
        s = ds * float(k - N)
        sc_w[k] = s
        Cs_w[k] = C(s)
    sc_w[0] = -1.0
    Cs_w[0] = -1.0
    for k in range(1, N + 1):
        s = ds * (float(k - N) - 0.5)
        sc_r[k] = s
        Cs_r[k] = C(s)
    return sc_r, Cs_r, sc_w, Cs_w


def make_params(cfg, NT):
    p = abi.Params()
    p.dt = cfg["dt"]
    p.dtfast = cfg["dt"] / float(cfg["ndtfast"])
    p.g, p.rho0 = G, RHO0
    p.gamma2 = cfg["gamma2"]
    p.lambda_ = 1.0                              # mod_scalars.F:724-729
    p.ndtfast = cfg["ndtfast"]
    nfast, w1, w2 = set_weights(cfg["ndtfast"])
    p.nfast = nfast
    for i in range(2 * cfg["ndtfast"]):
        p.weight1[i] = w1[i]
        p.weight2[i] = w2[i]
    p.Vtransform = 2
    p.hc = cfg["Tcline"]                          # Vtransform 2: hc = Tcline
    sc_r, Cs_r, sc_w, Cs_w = set_scoord(cfg["N"], cfg["theta_s"], cfg["theta_b"])
    for k in range(cfg["N"] + 1):
        p.sc_r[k], p.Cs_r[k], p.sc_w[k], p.Cs_w[k] = sc_r[k], Cs_r[k], sc_w[k], Cs_w[k]
    for it in range(NT):
        p.Hadv[it] = abi.ADV[cfg.get("Hadv_list", [cfg["Hadv"]] * NT)[it]]
        p.Vadv[it] = abi.ADV[cfg.get("Vadv_list", [cfg["Vadv"]] * NT)[it]]
    # the shipped applications are zonal channels (LBC = Per Clo Per Clo); "EWperiodic": False closes the
    # western / eastern edges as well (a basin), per-variable conditions then go into p.lbc
    p.lbc_west = p.lbc_east = abi.LBC_PERIODIC if cfg.get("EWperiodic", True) else abi.LBC_CLOSED
    p.lbc_south = p.lbc_north = abi.LBC_CLOSED
    app = cfg["app"]
    p.R0, p.T0, p.S0 = 1027.0, {"BENCHMARK": 10.0, "UPWELLING": 14.0, "SEAMOUNT": 10.0}[app], \
        {"BENCHMARK": 35.0, "UPWELLING": 35.0, "SEAMOUNT": 32.0}[app]
    p.Tcoef = 1.7e-4
    p.Scoef = {"BENCHMARK": 7.6e-4, "UPWELLING": 0.0, "SEAMOUNT": 7.6e-4}[app]
    p.nonlin_eos = int(app == "BENCHMARK")
    # option switches of the application headers (ROMS/Include/{benchmark,upwelling,seamount}.h)
    p.uv_adv, p.uv_cor = 1, 1
    p.uv_vis2 = int(cfg.get("uv_vis2", app in ("BENCHMARK", "UPWELLING")))     # 2 = UV_VIS2 with MIX_GEO_UV (uv3dmix2_geo.h) instead of MIX_S_UV
    p.curvgrid = int(app == "BENCHMARK")
    p.var_rho_2d = 1                      # globaldefs.h:491-495, always with SOLVE3D
    p.ts_dif2 = int(cfg.get("ts_dif2", 1))
    p.ts_dif4 = int(cfg.get("ts_dif4", 0))
    p.uv_vis4 = int(cfg.get("uv_vis4", 0))
    p.mix_iso_ts = int(cfg.get("mix_iso_ts", 0))
    p.radiation_2d = int(cfg.get("radiation_2d", 0))
    p.mix_geo_ts = int(app in ("BENCHMARK", "SEAMOUNT") and not p.mix_iso_ts)
    p.mix_s_ts = int(app == "UPWELLING" and not p.mix_iso_ts)
    p.salinity = int(app in ("BENCHMARK", "UPWELLING"))
    p.lmd_nonlocal = int(app == "BENCHMARK")
    p.solar_source = int(app == "BENCHMARK")
    p.splines_vdiff = int(cfg.get("splines_vdiff", 1))   # SPLINES_VDIFF / SPLINES_VVISC: 30 of the reference's 31 3-D applications
    p.splines_vvisc = int(cfg.get("splines_vvisc", 1))
    for it in range(NT):
        p.Akt_bak[it] = {"BENCHMARK": 1.0e-5}.get(app, 1.0e-6)
    p.Akv_bak = {"BENCHMARK": 1.0e-4}.get(app, 1.0e-5)
    # WTYPE == 1 (roms_benchmark*.in:392): mod_scalars.F:1502-1512
    p.swfrac_mu1, p.swfrac_mu2, p.swfrac_r1 = 0.35, 23.0, 0.58
    # set_vbc.F: BENCHMARK and SEAMOUNT have UV_QDRAG (rdrg2 = 3.0d-03), UPWELLING UV_LDRAG (rdrg = 3.0d-04)
    p.uv_drag = int(cfg.get("uv_drag", 1 if cfg["app"] == "UPWELLING" else 2))      # 3 = UV_LOGDRAG
    p.limit_bstress = int(cfg.get("limit_bstress", 0))
    p.eminusp = int(cfg.get("eminusp", 0))
    p.Cdb_min, p.Cdb_max = 1.0e-6, 0.5                                  # mod_scalars.F:747-748
    p.blk_ZQ = p.blk_ZT = p.blk_ZW = 10.0       # roms_*.in:382-384
    # GLS_MIXING: cfg["gls"] = one of the parameter sets of roms_*.in (roms_upwelling.in:352-364, :1862-1874);
    # cfg["gls_stability"] = GALPERIN | KANTHA_CLAYSON | CANUTO_A | CANUTO_B (the CPP choice of the application)
    # cfg["gls"] = "my25": MY25_MIXING instead (my25_prestep.F / my25_corstep.F: tke = q2, gls = q2l; of the closure's
    # input parameters only gls_Kmin / gls_Pmin -- the initial values, mod_mixing.F:1464-1473 -- and Akk_bak are read)
    if cfg.get("gls"):
        p.gls_mixing = 2 if cfg["gls"] == "my25" else 1
        (p.gls_p, p.gls_m, p.gls_n, p.gls_Kmin, p.gls_Pmin, p.gls_cmu0, p.gls_c1, p.gls_c2, p.gls_c3m, p.gls_c3p,
         p.gls_sigk, p.gls_sigp) = GLS_SETS[cfg["gls"]]
        p.gls_stability = abi.GLS_STAB[cfg.get("gls_stability", "KANTHA_CLAYSON")]
        p.gls_n2s2_horavg = int(cfg.get("gls_n2s2_horavg", 1))
        p.gls_ri_splines = int(cfg.get("gls_ri_splines", 1))
        p.Akk_bak = p.Akp_bak = 5.0e-6              # AKK_BAK, AKP_BAK (roms_upwelling.in)
        p.Zos = 0.02                                # Zos (roms_upwelling.in:378)
    # WET_DRY: DCRIT of roms_*.in (0.10 m in every input script of the reference, e.g. roms_upwelling.in)
    p.atm_press = int(cfg.get("atm_press", 0))       # ATM_PRESS: Pair in the baroclinic pressure gradient
    p.press_compensate = int(cfg.get("press_compensate", 0))   # ... and (PRESS_COMPENSATE) in the Flather value
    p.ts_mix_min_strat = int(cfg.get("ts_mix_min_strat", 0))   # TS_MIX_MIN_STRAT (with MIX_ISO_TS)
    p.ts_mix_stability = int(cfg.get("ts_mix_stability", 0))   # TS_MIX_STABILITY: 3/4 t(nrhs) + 1/4 t(nstp) in the tracer mixing
    p.wet_dry = int(cfg.get("wet_dry", 0))
    p.Dcrit = float(cfg.get("Dcrit", 0.10))
    return p


# GLS_P, GLS_M, GLS_N, GLS_Kmin, GLS_Pmin, GLS_CMU0, GLS_C1, GLS_C2, GLS_C3M, GLS_C3P, GLS_SIGK, GLS_SIGP
# (the table of roms_upwelling.in:1862-1874)
GLS_SETS = {
    "k-kl":      (0.0, 1.0, 1.0, 5.0e-6, 5.0e-6, 0.5544, 0.9, 0.52, 2.5, 1.0, 1.96, 1.96),
    "k-epsilon": (3.0, 1.5, -1.0, 7.6e-6, 1.0e-12, 0.5477, 1.44, 1.92, -0.4, 1.0, 1.0, 1.30),
    "k-omega":   (-1.0, 0.5, -1.0, 7.6e-6, 1.0e-12, 0.5477, 0.555, 0.833, -0.6, 1.0, 2.0, 2.0),
    "gen":       (2.0, 1.0, -0.67, 1.0e-8, 1.0e-8, 0.5544, 1.0, 1.22, 0.1, 1.0, 0.8, 1.07),
    "my25":      (0.0, 1.0, 1.0, 5.0e-6, 5.0e-6, 0.5544, 0.9, 0.52, 2.5, 1.0, 1.96, 1.96),      # MY25_MIXING: the k-kl line of roms_*.in
}


def _grid_global(cfg, b, st):
    """Fill the 2-D grid arrays over the whole allocated range of tile bounds b
    (analytic functions are evaluated directly at ghost indices, which equals
    what the reference obtains after its periodic exchanges)."""
    Lm, Mm = cfg["Lm"], cfg["Mm"]
    app = cfg["app"]
    ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None]
    jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :]
    ni, nj = st.ni, st.nj
    ones = np.ones((ni, nj))
    if app == "BENCHMARK":
        Xsize, Esize = 360.0, 20.0
        dx, dy = Xsize / float(Lm), Esize / float(Mm)
        latr = (-70.0 + dy * (jj - 0.5)) * ones
        val1 = float(Lm) / (2.0 * math.pi * ERADIUS)
        val2 = float(Mm) * 360.0 / (2.0 * math.pi * ERADIUS * Esize)
        pm = val1 * (1.0 / np.cos(latr * DEG2RAD))
        pn = val2 * ones
        fval = 2.0 * (2.0 * math.pi * 366.25 / 365.25) / 86400.0
        f = fval * np.sin(latr * DEG2RAD)
        h = 500.0 + 1750.0 * (1.0 + np.tanh((68.0 + latr) / dy))
        st.lonr = dx * (ii - 0.5) * ones
        st.latr = latr
    else:
        if app == "UPWELLING":
            Xsize, Esize, depth, f0 = 1000.0 * Lm, 1000.0 * Mm, 150.0, -8.26e-5
        else:  # SEAMOUNT
            Xsize, Esize, depth, f0 = 320.0e3, 320.0e3, 5000.0, 1.0e-4
        dx, dy = Xsize / float(Lm), Esize / float(Mm)
        xr = dx * (ii - 0.5) * ones
        yr = dy * (jj - 0.5) * ones
        pm = ones / dx
        pn = ones / dy
        f = f0 * ones
        if app == "UPWELLING":
            jv = np.where(jj <= Mm // 2, jj, Mm + 1 - jj)
            h = np.minimum(depth, 84.5 + 66.526 * np.tanh((jv - 10.0) / 7.0)) * ones
        else:
            # periodic images of the seamount do not matter: it sits mid-domain
            xw = np.mod(xr, Xsize)
            v1 = (xw - 0.5 * Xsize) / 40000.0
            v2 = (yr - 0.5 * Esize) / 40000.0
            h = depth - 4500.0 * np.exp(-(v1 * v1 + v2 * v2))
        st.lonr, st.latr = xr, yr
    A = st.arr
    A["lonr"][:] = st.lonr
    A["latr"][:] = st.latr
    A["pm"][:] = pm
    A["pn"][:] = pn
    A["f"][:] = f
    if cfg.get("beach"):
        # WET_DRY test bathymetry (no application header of the reference has one that fits a 3-D channel): the
        # northern side shoals to a beach whose shoreline (h = 0) meanders around row Mm - 3.5 -- 0.25 m per row near
        # the shoreline, bed above the resting level (h < 0) beyond it; a function of the global indices only
        d = (Mm - 3.5 - jj) + 1.5 * np.sin(2.0 * math.pi * (ii - 0.5) / Lm)
        hb = np.where(d > 0.0, 0.25 * d + 0.15 * d * d, 0.25 * d)
        h = np.minimum(h, hb * ones)
    A["h"][:] = h
    # ---- metrics.F ----
    A["om_r"][:] = 1.0 / pm
    A["on_r"][:] = 1.0 / pn
    A["omn"][:] = 1.0 / (pm * pn)
    A["fomn"][:] = f * A["omn"]
    A["pnom_r"][:] = pn / pm
    A["pmon_r"][:] = pm / pn
    s = slice(1, None)
    m = slice(0, -1)
    A["pmon_u"][s, :] = (pm[m, :] + pm[s, :]) / (pn[m, :] + pn[s, :])
    A["pnom_u"][s, :] = (pn[m, :] + pn[s, :]) / (pm[m, :] + pm[s, :])
    A["om_u"][s, :] = 2.0 / (pm[m, :] + pm[s, :])
    A["on_u"][s, :] = 2.0 / (pn[m, :] + pn[s, :])
    A["pmon_v"][:, s] = (pm[:, m] + pm[:, s]) / (pn[:, m] + pn[:, s])
    A["pnom_v"][:, s] = (pn[:, m] + pn[:, s]) / (pm[:, m] + pm[:, s])
    A["om_v"][:, s] = 2.0 / (pm[:, m] + pm[:, s])
    A["on_v"][:, s] = 2.0 / (pn[:, m] + pn[:, s])
    pm4 = pm[m, m] + pm[m, s] + pm[s, m] + pm[s, s]
    pn4 = pn[m, m] + pn[m, s] + pn[s, m] + pn[s, s]
    A["pnom_p"][s, s] = pn4 / pm4
    A["pmon_p"][s, s] = pm4 / pn4
    A["om_p"][s, s] = 4.0 / pm4
    A["on_p"][s, s] = 4.0 / pn4
    # metrics.F exchanges the u- and psi-type combinations: their western-most ghost column is the periodic image
    if b.EWperiodic and b.ntileI == 1:
        for name in ("pmon_u", "pnom_u", "om_u", "on_u", "pmon_p", "pnom_p", "om_p", "on_p"):
            A[name][0, :] = A[name][Lm, :]
    # ana_grid.h:757-770 (CURVGRID && UV_ADV): interior rows only, wall rows stay 0
    if app == "BENCHMARK":
        c = slice(1, -1)
        dndx = np.zeros((ni, nj))
        dmde = np.zeros((ni, nj))
        dndx[c, :] = 0.5 * (1.0 / pn[2:, :] - 1.0 / pn[:-2, :])
        dmde[:, c] = 0.5 * (1.0 / pm[:, 2:] - 1.0 / pm[:, :-2])
        jlo, jhi = st.J(max(b.LBj, 1)), st.J(min(b.UBj, Mm))
        A["dndx"][:, jlo:jhi + 1] = dndx[:, jlo:jhi + 1]
        A["dmde"][:, jlo:jhi + 1] = dmde[:, jlo:jhi + 1]
    # horizontal mixing coefficients (ini_hmixcoef.F: uniform, not grid-scaled here)
    A["visc2_r"][:] = cfg["visc2"]
    A["visc2_p"][:] = cfg["visc2"]
    A["diff2"][:] = cfg["tnu2"]
    # biharmonic coefficients: the reference stores the square root (inp_par.F:986, read_phypar.F:6905)
    A["visc4_r"][:] = math.sqrt(abs(cfg.get("visc4", 0.0)))
    A["visc4_p"][:] = math.sqrt(abs(cfg.get("visc4", 0.0)))
    A["diff4"][:] = math.sqrt(abs(cfg.get("tnu4", 0.0)))


def set_masks(st, rmask):
    """Land/sea masks of one tile from the rho-point mask (1 = water): umask, vmask as ana_mask.h:226-236,
    pmask with the slipperiness rule of metrics.F:535-596 (all four water or one land -> 1, a straight coast,
    two land cells on one side -> 2 = no-slip, else 0).  Arrays cover the allocated tile; `rmask` must be a
    function of the global indices so that every tiling sees the same coast."""
    A = st.arr
    A["rmask"][:] = rmask
    s, m = slice(1, None), slice(0, -1)
    A["umask"][:] = 0.0
    A["vmask"][:] = 0.0
    A["pmask"][:] = 0.0
    A["umask"][s, :] = rmask[m, :] * rmask[s, :]
    A["vmask"][:, s] = rmask[:, m] * rmask[:, s]
    w = rmask > 0.5
    a, b_, c, d = w[m, s], w[s, s], w[m, m], w[s, m]          # (i-1,j) (i,j) (i-1,j-1) (i,j-1)
    nland = (~a).astype(int) + (~b_).astype(int) + (~c).astype(int) + (~d).astype(int)
    straight = (a & ~b_ & c & ~d) | (~a & b_ & ~c & d) | (a & b_ & ~c & ~d) | (~a & ~b_ & c & d)
    A["pmask"][s, s] = np.where(nland <= 1, 1.0, np.where((nland == 2) & straight, 2.0, 0.0))
    st.p.masking = 1


def island_mask(cfg, b):
    """A test coast that is the same for every tiling: an island (an ellipse plus a one-cell spur) east of the
    middle of the domain and a headland attached to the southern wall; at least three water cells everywhere else."""
    Lm, Mm = cfg["Lm"], cfg["Mm"]
    ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None]
    jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :]
    iw = np.mod(ii - 1.0, Lm) + 1.0                            # periodic image of the ghost columns
    ic, jc = 0.62 * Lm, 0.55 * Mm
    ri, rj = max(3.0, 0.09 * Lm), max(3.0, 0.12 * Mm)
    land = ((iw - ic) / ri) ** 2 + ((jj - jc) / rj) ** 2 <= 1.0
    land |= (np.abs(iw - np.floor(ic)) < 0.5) & (jj > jc) & (jj <= jc + rj + 2.0)            # spur, one cell wide
    land |= (np.abs(iw - 0.25 * Lm) <= max(2.0, 0.04 * Lm)) & (jj <= max(3.0, 0.15 * Mm))   # headland on the wall
    return np.where(land, 0.0, 1.0) * np.ones((b.UBi - b.LBi + 1, b.UBj - b.LBj + 1))


def z_levels(st, zeta2d):
    """set_depth.F:82-300, Vtransform == 2 (numpy restatement used for IC only)."""
    b, p = st.b, st.p
    N = b.N
    h = st["h"]
    hinv = 1.0 / (p.hc + h)
    z_w = np.zeros((st.ni, st.nj, N + 1), order="F")
    z_r = np.zeros((st.ni, st.nj, N), order="F")
    z_w[:, :, 0] = -h
    for k in range(1, N + 1):
        cff_w = p.hc * p.sc_w[k]
        cff_r = p.hc * p.sc_r[k]
        cff2_r = (cff_r + p.Cs_r[k] * h) * hinv
        cff2_w = (cff_w + p.Cs_w[k] * h) * hinv
        z_w[:, :, k] = zeta2d + (zeta2d + h) * cff2_w
        z_r[:, :, k - 1] = zeta2d + (zeta2d + h) * cff2_r
    return z_r, z_w


def make_tile(config, ntileI=1, ntileJ=1, tile=0, NT=None, overrides=None,
              perturb=0.0, seed=0, mask=None):
    """Build bounds, parameters and an initialised TileState for one tile of a
    named configuration.  `perturb` adds a smooth 2-D (i- and j-dependent)
    perturbation to the initial temperature and free surface so that
    x-direction stencil errors cannot hide behind BENCHMARK's zonal symmetry
    (hard parts)."""
    cfg = dict(CONFIGS[config])
    if overrides:
        cfg.update(overrides)
    NAT = cfg["NAT"]
    NT = NT or NAT
    adv = {cfg["Hadv"]} | set(cfg.get("Hadv_list", []))        # the horizontal schemes decide (inp_par.F:264-267)
    nghost = 3 if (adv & {"MPDATA", "HSIMT"}) or cfg.get("uv_vis4") else 2          # inp_par.F:264-278
    b = make_bounds(cfg["Lm"], cfg["Mm"], cfg["N"], NT, NAT, ntileI, ntileJ, tile,
                    EWperiodic=bool(cfg.get("EWperiodic", True)), NSperiodic=False, NghostPoints=nghost)
    p = make_params(cfg, NT)
    st = TileState(b, p)
    st.cfg = cfg
    _grid_global(cfg, b, st)
    app = cfg["app"]
    A = st.arr
    Lm, Mm, N = cfg["Lm"], cfg["Mm"], cfg["N"]
    ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None]
    jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :]
    # smooth doubly-varying bump, periodic in i
    bump = np.sin(2.0 * math.pi * (ii - 0.5) / Lm * 3.0) * np.sin(math.pi * (jj - 0.5) / Mm * 2.0)
    zeta0 = perturb * cfg.get("zeta_amp", 0.05) * bump
    if p.wet_dry:                            # an initial state consistent with the bed: total depth >= Dcrit
        zeta0 = np.maximum(zeta0, p.Dcrit - A["h"])
    # closed walls: zero-gradient ghost rows for zeta (zetabc.F closed branches)
    for k in range(3):
        A["zeta"][:, :, k] = zeta0
    A["Zt_avg1"][:] = zeta0
    z_r, z_w = z_levels(st, zeta0)
    if app == "BENCHMARK":
        val1 = (44.69 / 39.382) ** 2
        val2 = val1 * (RHO0 * 800.0 / G) * (5.0e-05 / ((42.689 / 44.69) ** 2))
        T = val2 * np.exp(z_r / 800.0) * (0.6 - 0.4 * np.tanh(z_r / 800.0))
        S = 35.0
    elif app == "UPWELLING":
        T = p.T0 + 8.0 * np.exp(z_r / 50.0)
        S = p.S0
    else:
        T = p.T0 + 7.5 * np.exp(z_r / 1000.0)
        S = None
    T = T + perturb * 0.5 * bump[:, :, None] * np.exp(z_r / 200.0)
    for lev in range(3):
        A["t"][:, :, :, lev, 0] = T
        if NAT > 1:
            A["t"][:, :, :, lev, 1] = S
        for it in range(NAT, NT):       # passive tracers: positive definite blobs
            pbump = 1.0 + np.exp(-(((ii - 0.5) / Lm - (it - NAT + 1) / (NT - NAT + 1.0)) / 0.08) ** 2
                                 - (((jj - 0.5) / Mm - 0.5) / 0.25) ** 2)
            A["t"][:, :, :, lev, it] = pbump[:, :, None] * np.exp(z_r / 500.0) + 0.5
    # ---- fixed forcing / mixing inputs (stand in for bulk_flux, set_vbc, lmd_vmix) ----
    if app == "BENCHMARK":
        lat = st.latr
        uwind = 15.0 * np.exp(-(0.2 * (60.0 + lat)) ** 2)     # ana_winds.h:118-126
        tau = 1.2 * 1.3e-3 * uwind * uwind / RHO0             # bulk stress magnitude, m2/s2
        A["sustr"][:] = tau
        A["svstr"][:] = 0.0
        A["srflx"][:] = 150.0 / (RHO0 * 3985.0) * (0.5 + 0.5 * np.cos(2 * math.pi * (ii - 0.5) / Lm))
        A["stflx"][:, :, 0] = A["srflx"] - 100.0 / (RHO0 * 3985.0)
        Akv = 1.0e-4 + 1.0e-2 * np.exp(z_w / 60.0)
        Akt = 1.0e-5 + 5.0e-3 * np.exp(z_w / 60.0)
        A["Akv"][:] = Akv
        for it in range(NAT):
            A["Akt"][:, :, :, it] = Akt * (1.0 if it == 0 else 0.8)
            gh = 2.0e-2 * np.exp(z_w / 30.0) * (z_w / z_w[:, :, :1]) * (1.0 - z_w / z_w[:, :, :1])
            A["ghats"][:, :, :, it] = gh * (1.0 if it == 0 else 0.0)
    elif app == "UPWELLING":
        # ana_smflux.h (UPWELLING): constant along-shore stress; ana_vmix.h:200,327
        A["sustr"][:] = -0.1 / RHO0                       # ana_smflux.h:316-330 after its two-day ramp
        A["srflx"][:] = 0.0
        A["Akv"][:] = 2.0e-3 + 8.0e-3 * np.exp(z_w / 150.0)
        for it in range(NAT):
            A["Akt"][:, :, :, it] = 1.0e-6
    else:
        A["Akv"][:] = p.Akv_bak
        for it in range(NAT):
            A["Akt"][:, :, :, it] = p.Akt_bak[it]
    # ---- inputs of the per-step physics (set_vbc, bulk_flux): roms_*.in and ana_*.h of the app ----
    A["rdrag2"][:] = 3.0e-3                       # RDRG2 (roms_benchmark*.in)
    A["rdrag"][:] = 3.0e-4                        # RDRG
    A["ZoBot"][:] = 0.02                          # Zob (m)
    A["stflux"][:] = A["stflx"]                   # raw surface tracer fluxes (ana_stflux.h: 0; heat: fixed profile)
    A["btflux"][:] = A["btflx"]
    if app == "BENCHMARK":
        cffw = 0.2 * (60.0 + st.latr)
        A["Uwind"][:] = 15.0 * np.exp(-cffw * cffw)      # ana_winds.h:118-126, Wmag = 15 m/s
        A["Vwind"][:] = 0.0
        A["Tair"][:] = 4.0                           # ana_tair.h
        A["Pair"][:] = 1025.0                        # ana_pair.h
        A["Hair"][:] = 0.8                           # ana_humid.h
        A["rain"][:] = 0.0                           # ana_rain.h
        A["cloud"][:] = 0.6                          # ana_cloud.h
    else:                                            # not BULK_FLUXES applications: plausible test inputs
        A["Uwind"][:] = 5.0 + 2.0 * bump
        A["Vwind"][:] = -3.0 + 1.5 * bump
        A["Tair"][:] = 12.0
        A["Pair"][:] = 1013.0
        A["Hair"][:] = 0.7
        A["rain"][:] = 1.0e-5
        A["cloud"][:] = 0.4
    if p.gls_mixing:
        # mod_mixing.F initialize_mixing: tke = gls_Kmin, gls = gls_Pmin, Akk / Akp / Akv / Akt = background; a smooth
        # `perturb` on top so that no stencil coefficient of the closure multiplies a constant field in the tests
        wob = 1.0 + perturb * 0.5 * (1.0 + bump[:, :, None] * np.cos(math.pi * np.arange(N + 1) / N)[None, None, :])
        for lev in range(3):
            A["tke"][:, :, :, lev] = p.gls_Kmin * wob * (1.0 + 0.1 * lev * perturb)
            A["gls"][:, :, :, lev] = p.gls_Pmin * wob * (1.0 + 0.07 * lev * perturb)
        A["Akk"][:] = p.Akk_bak * wob
        A["Akp"][:] = p.Akp_bak * wob
        A["Lscale"][:] = perturb * 0.3 * wob
    st.z_r0, st.z_w0 = z_r, z_w
    # land/sea masks: all water unless asked for (MASKING applications); mask = "island" -> island_mask()
    for name in ("rmask", "umask", "vmask", "pmask"):
        A[name][:] = 1.0
    p.masking = 0
    # pressure-gradient algorithm (prsgrd.F:16-26): the three application headers define DJ_GRADPS
    p.pgf = abi.PGF[cfg.get("pgf", "DJ_GRADPS")]
    if p.wet_dry and mask is None:           # WET_DRY needs MASKING (wetdry.F:325 reads rmask unconditionally): all water
        set_masks(st, np.ones((st.ni, st.nj)))
    for name in ("pmask_wet", "rmask_wet", "umask_wet", "vmask_wet", "pmask_full", "rmask_full", "umask_full",
                 "vmask_full"):
        A[name][:] = 1.0                      # all wet until initial.F's wetdry call (main3d.initial) sets them
    A["rmask_wet_avg"][:] = 0.0
    if mask == "island":
        set_masks(st, island_mask(cfg, b))
        # masked initial state, as the reference's ini_fields / ana_initial leave it
        A["zeta"][:] *= A["rmask"][:, :, None]
        A["Zt_avg1"][:] *= A["rmask"]
    elif mask is not None:
        raise ValueError(f"unknown mask {mask!r}")
    return st
