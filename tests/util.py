"""Shared helpers for the parity tests: seeded, smooth, fully 2-D-varying
states that exercise every stencil direction (BENCHMARK's
zonal symmetry can hide x-direction bugs, hence the perturbations)."""
import math

import numpy as np

from roms_trunk_mgh_amd import abi, ana


def step_idx(iic=3, ntfirst=1, nstp=1, nnew=2, nrhs=1, kstp=1, krhs=1, knew=2, iif=1, pred=0):
    return abi.StepIdx(iic=iic, ntfirst=ntfirst, nstp=nstp, nnew=nnew, nrhs=nrhs, kstp=kstp,
                       krhs=krhs, knew=knew, iif=iif, predictor_2d_step=pred)


# WET = True (set by tests/ref_worker.py "... wet" and by the WET_DRY tests): prepared_state builds a WET_DRY state --
# wet_dry = 1 and synthetic wet/dry masks of the general (fast-step) kind, which hold every value the reference's
# wetdry_mask_tile can produce (0, 1, 2 and -1 / 1 on one-sided faces)
WET = False


def synthetic_wet_masks(st, seed=5):
    """Wet/dry masks from a seeded rho-point flag (dry blobs, single dry cells, dry stretches along the edges) by the
    rules of wetdry.F:563-716 restated on whole arrays; independent of the oracle's loops."""
    b = st.b
    rng = np.random.default_rng(seed)
    ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None]
    jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :]
    iw = np.mod(ii - 1.0, b.Lm) + 1.0 if b.EWperiodic else ii
    wd = np.ones((st.ni, st.nj))
    for ic, jc, r in ((0.3 * b.Lm, 0.7 * b.Mm, 3.2), (0.8 * b.Lm, 0.25 * b.Mm, 2.4), (0.55 * b.Lm, b.Mm + 0.5, 4.0),
                      (0.15 * b.Lm, 0.5, 3.0), (1.0, 0.5 * b.Mm, 2.5), (b.Lm, 0.8 * b.Mm, 2.2)):
        wd[(iw - ic) ** 2 + (jj - jc) ** 2 <= r * r] = 0.0
    glob = rng.random((b.Lm + 8, b.Mm + 8)) < 0.03            # lone dry cells: a function of the global indices
    wd[glob[(iw.astype(int) + 3) % (b.Lm + 8), (jj.astype(int) + 3) % (b.Mm + 8)]] = 0.0
    wd *= st["rmask"]
    m, s = slice(0, -1), slice(1, None)
    um = np.zeros_like(wd)
    vm = np.zeros_like(wd)
    um[s, :] = wd[m, :] + wd[s, :]
    um[s, :] = np.where(um[s, :] == 1.0, wd[m, :] - wd[s, :], um[s, :])
    vm[:, s] = wd[:, m] + wd[:, s]
    vm[:, s] = np.where(vm[:, s] == 1.0, wd[:, m] - wd[:, s], vm[:, s])
    a, bq, c, d = wd[m, s] > 0.5, wd[s, s] > 0.5, wd[m, m] > 0.5, wd[s, m] > 0.5
    nwet = a.astype(int) + bq.astype(int) + c.astype(int) + d.astype(int)
    side = (a & c & ~bq & ~d) | (bq & d & ~a & ~c) | (a & bq & ~c & ~d) | (c & d & ~a & ~bq)
    pm = np.zeros_like(wd)
    pm[s, s] = np.where(nwet >= 3, 1.0, np.where((nwet == 2) & side, 2.0, 0.0))
    st["rmask_wet"][:] = wd
    st["umask_wet"][:] = um
    st["vmask_wet"][:] = vm
    st["pmask_wet"][:] = pm
    st["rmask_full"][:] = wd * st["rmask"]
    st["umask_full"][:] = um * st["umask"]
    st["vmask_full"][:] = vm * st["vmask"]
    st["pmask_full"][:] = np.maximum(pm * st["pmask"], 2.0)
    return wd


def prepared_state(config, seed=1, NT=None, overrides=None, oracle_backend=None, mask=None, wet=None):
    """A tile state with non-trivial velocities, fluxes, RHS terms and tracers at
    all time levels.  Deterministic (seeded).  mask = "island": a MASKING grid (ana.island_mask); the
    prognostic fields are then zero on land, as the reference keeps them."""
    import oracle
    wet = WET if wet is None else wet
    if wet:
        overrides = dict(overrides or {}, wet_dry=1)
    st = ana.make_tile(config, perturb=1.0, NT=NT, overrides=overrides, mask=mask)
    if wet:
        synthetic_wet_masks(st)
    b = st.b
    rng = np.random.default_rng(seed)
    Lm, Mm, N = b.Lm, b.Mm, b.N
    ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None, None]
    jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :, None]
    kk = (np.arange(1, N + 1, dtype=np.float64) / N)[None, None, :]
    px = 2.0 * math.pi / Lm
    py = math.pi / Mm

    def smooth(a, bq, ph):
        return np.sin(px * a * (ii - 0.5) + ph) * np.cos(py * bq * (jj - 0.5)) * (0.3 + kk)

    for lev in range(2):
        st["u"][:, :, :, lev] = 0.15 * smooth(2, 1, 0.3 + lev) + 0.02 * smooth(5, 3, 1.0)
        st["v"][:, :, :, lev] = 0.05 * np.sin(px * 3 * (ii - 0.5)) * np.sin(py * (jj - 1.0)) * (0.3 + kk)
    # closed walls: v = 0 on wall rows, outside rows unused
    for lev in range(3):
        amp = 1.0 + 0.1 * lev
        st["ubar"][:, :, lev] = amp * 0.05 * smooth(2, 1, 0.2)[:, :, 0]
        st["vbar"][:, :, lev] = amp * 0.02 * (np.sin(px * 3 * (ii - 0.5)) * np.sin(py * (jj - 1.0)))[:, :, 0]
        st["zeta"][:, :, lev] = amp * 0.1 * (np.cos(px * 2 * (ii - 0.5)) * np.cos(py * (jj - 0.5)))[:, :, 0]
    st["Zt_avg1"][:] = st["zeta"][:, :, 0]
    s = step_idx()
    o = oracle.Oracle(st)
    o.call("set_depth", s)
    o.call("set_massflux", s)
    o.call("omega", s)
    # tracer time levels differ slightly from one another
    for it in range(b.NT):
        base = st["t"][:, :, :, 0, it].copy()
        st["t"][:, :, :, 1, it] = base * (1.0 + 1e-3 * smooth(1, 2, 0.7))
        st["t"][:, :, :, 2, it] = base * (1.0 + 2e-3 * smooth(3, 1, 0.1))
        # t(nnew) enters step3d_t in units of Hz*t (pre_step3d leaves it so)
    for lev in range(2):
        st["ru"][:, :, 1:, lev] = 1e-1 * smooth(2, 2, 0.4 + lev)
        st["rv"][:, :, 1:, lev] = 1e-1 * smooth(1, 3, 0.9 + lev)
        st["ru"][:, :, 0, lev] = 1e-1 * smooth(2, 2, 0.4 + lev)[:, :, 0]
        st["rv"][:, :, 0, lev] = 1e-1 * smooth(1, 3, 0.9 + lev)[:, :, 0]
    for name in ("DU_avg1", "DU_avg2"):
        st[name][:] = 50.0 * smooth(2, 1, 0.2)[:, :, 0]
    for name in ("DV_avg1", "DV_avg2"):
        st[name][:] = 20.0 * (np.sin(px * 3 * (ii - 0.5)) * np.sin(py * (jj - 1.0)))[:, :, 0]
    st["rho"][:] = 2.0 * smooth(1, 1, 0.0) - 3.0 * kk + 28.0 * (1 - kk)
    st["rhoA"][:] = 0.02 + 0.001 * smooth(1, 1, 0.5)[:, :, 0]
    st["rhoS"][:] = 0.01 + 0.001 * smooth(2, 1, 0.5)[:, :, 0]
    st["bustr"][:] = 1e-5 * smooth(2, 1, 0.2)[:, :, 0]
    st["bvstr"][:] = 1e-5 * smooth(1, 2, 0.6)[:, :, 0]
    st["btflx"][:] = 0.0
    _ = rng
    if mask is not None:
        rm, um, vm = st["rmask"], st["umask"], st["vmask"]
        st["zeta"] *= rm[:, :, None]
        st["Zt_avg1"] *= rm
        st["ubar"] *= um[:, :, None]
        st["vbar"] *= vm[:, :, None]
        st["u"] *= um[:, :, None, None]
        st["v"] *= vm[:, :, None, None]
        st["t"] *= rm[:, :, None, None, None]
        for name in ("DU_avg1", "DU_avg2"):
            st[name] *= um
        for name in ("DV_avg1", "DV_avg2"):
            st[name] *= vm
        o.call("set_depth", s)
        o.call("set_massflux", s)
        o.call("omega", s)
    return st


def kpp_state(config="BENCHMARK_TINY", mask=None):
    """prepared_state made into a meaningful input of KPP (lmd_vmix): density, buoyancy frequency, expansion
    coefficients from the (pinned) oracle rho_eos instead of the zeros prepared_state leaves; surface forcing that
    gives shallow boundary layers (ending inside the top layer: ksbl = N) in the western half of the domain and deep
    ones in the eastern half; surface / bottom diffusivities of salinity different from the temperature's (the routine
    leaves levels 0 and N alone and reads level N when ksbl = N)."""
    import oracle
    st = prepared_state(config, mask=mask)
    oracle.Oracle(st).call("rho_eos", step_idx())
    b = st.b
    st["stflux"][:, :, 0] += 1.0e-6
    if b.NT > 1:
        st["stflux"][:, :, 1] = 2.0e-8
        st["btflx"][:, :, 1] = 1.0e-9
        st["stflx"][:, :, 1] = 1.0e-7
    west = (np.arange(b.LBi, b.UBi + 1) <= b.Lm // 2)[:, None]
    st["stflx"][:, :, 0] = np.where(west, 2.0e-4, -3.0e-5 + st["stflx"][:, :, 0])   # strong heating / cooling ...
    st["srflx"][:] = np.where(west, 1.0e-4, st["srflx"])
    for name in ("sustr", "svstr"):                                          # ... under a weak / a strong wind
        st[name][:] = np.where(west, 0.03 * st[name], 6.0 * st[name])
    st["Akt"][:, :, 0, 1] *= 1.9
    st["Akt"][:, :, -1, 1] = 1.7 * st["Akt"][:, :, -1, 0] + 3.0e-5
    if mask is not None:
        for name in ("stflx", "btflx"):
            st[name] *= st["rmask"][:, :, None]
        st["srflx"] *= st["rmask"]
        st["sustr"] *= st["umask"]
        st["svstr"] *= st["vmask"]
    return st


GLS_BUILDS = {   # the CPP choices of the two GLS reference builds (oracle/ref_headers/upwelling_gls.h, benchmark_gls.h)
    "UPWELLING": dict(gls_stability="KANTHA_CLAYSON", gls_n2s2_horavg=1, gls_ri_splines=1),
    "BENCHMARK": dict(gls_stability="CANUTO_A", gls_n2s2_horavg=0, gls_ri_splines=0),
}
MY25_BUILDS = {  # ... and of the MY25_MIXING builds (upwelling_my25.h, benchmark_my25.h)
    "UPWELLING": dict(gls_stability="KANTHA_CLAYSON", gls_n2s2_horavg=1, gls_ri_splines=1),
    "BENCHMARK": dict(gls_stability="GALPERIN", gls_n2s2_horavg=0, gls_ri_splines=0),
}


def gls_state(config, gls="k-epsilon", mask=None, basin=False, extra=None):
    """prepared_state of a GLS_MIXING application with the CPP choices of the matching reference build, made into a
    meaningful input of gls_prestep / gls_corstep: bvf from the (pinned) oracle rho_eos with both signs of the
    stratification, energetic tke / gls at the three time levels, diffusivities above their backgrounds, W as omega
    left it (prepared_state)."""
    import oracle
    from roms_trunk_mgh_amd import ana as _ana
    ov = dict((MY25_BUILDS if gls == "my25" else GLS_BUILDS)[_ana.CONFIGS[config]["app"]], gls=gls)
    if basin:
        ov["EWperiodic"] = False
    if extra:
        ov.update(extra)
    st = prepared_state(config, mask=mask, overrides=ov)
    oracle.Oracle(st).call("rho_eos", step_idx())
    b, p = st.b, st.p
    ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None, None]
    jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :, None]
    kk = (np.arange(0, b.N + 1, dtype=np.float64) / b.N)[None, None, :]
    wob = 1.0 + 0.5 * np.sin(2.0 * math.pi * 2 * (ii - 0.5) / b.Lm) * np.cos(math.pi * (jj - 0.5) / b.Mm) * (0.4 + kk)
    # unstable patches: the sign of the stratification selects gls_c3m / gls_c3p and the length-scale limiter
    st["bvf"][:] = np.where(np.sin(2.0 * math.pi * 3 * (ii - 0.5) / b.Lm) > 0.8, -0.3 * np.abs(st["bvf"]) - 1.0e-7, st["bvf"] + 1.0e-6 * wob)
    for lev in range(3):
        st["tke"][:, :, :, lev] = 2.0e-4 * wob * (1.0 + 0.05 * lev) * (0.2 + kk * (1.0 - kk) * 4.0)
        ls = 0.5 + 3.0 * kk * (1.0 - kk) * 4.0 * wob                      # a length scale (m)
        st["gls"][:, :, :, lev] = np.maximum((p.gls_cmu0 ** p.gls_p) * st["tke"][:, :, :, lev] ** p.gls_m * ls ** p.gls_n
                                             * (1.0 + 0.03 * lev), p.gls_Pmin)
    st["Lscale"][:] = 0.5 + 2.0 * kk * (1.0 - kk) * 4.0 * wob
    st["Akv"][:] = p.Akv_bak + 2.0e-3 * wob * kk * (1.0 - kk) * 4.0
    for it in range(b.NAT):
        st["Akt"][:, :, :, it] = p.Akt_bak[it] + 1.5e-3 * wob * kk * (1.0 - kk) * 4.0 * (1.0 + 0.1 * it)
    st["Akk"][:] = p.Akk_bak + st["Akv"] / p.gls_sigk
    st["Akp"][:] = p.Akp_bak + st["Akv"] / p.gls_sigp
    st["ZoBot"][:] = 0.02 * (1.0 + 0.3 * wob[:, :, 0])
    if mask is not None:
        for name in ("tke", "gls"):
            st[name] *= st["rmask"][:, :, None, None]
        st["sustr"] *= st["umask"]
        st["svstr"] *= st["vmask"]
    return st


def hz_weighted_tnew(st, nnew=2):
    """pre_step3d leaves t(nnew) multiplied by Hz; reproduce that for isolated
    step3d_t tests."""
    for it in range(st.b.NT):
        st["t"][:, :, :, nnew - 1, it] *= st["Hz"]
    return st


def max_rel_diff(a, bref):
    a = np.asarray(a)
    bref = np.asarray(bref)
    scale = max(float(np.max(np.abs(bref))), 1e-300)
    return float(np.max(np.abs(a - bref))) / scale


def compare_states(st_a, st_ref, names=None):
    from roms_trunk_mgh_amd import abi as _abi
    out = {}
    for name, _, _ in _abi.FIELDS:
        if names is not None and name not in names:
            continue
        d = max_rel_diff(st_a[name], st_ref[name])
        if d > 0.0:
            out[name] = d
    return out


def mpdata_private_arrays(st):
    """Deterministic inputs for an isolated mpdata_adiff_tile call: the tile-private arrays
    oHz, Ta (IminS:ImaxS,JminS:JmaxS,N) and the module-layout t(:,:,:,3,1).  Ta is positive,
    smooth and fully 3-D, with a patch of exact zeros and a flat patch so that the
    "no anti-diffusion" branches (Ta <= 0, |dTa| <= eps2) are taken too."""
    b = st.b
    IminS, ImaxS, JminS, JmaxS = b.Istr - 3, b.Iend + 3, b.Jstr - 3, b.Jend + 3
    nis, njs, N = ImaxS - IminS + 1, JmaxS - JminS + 1, b.N
    ii = np.arange(IminS, ImaxS + 1, dtype=np.float64)[:, None, None]
    jj = np.arange(JminS, JmaxS + 1, dtype=np.float64)[None, :, None]
    kk = np.arange(1, N + 1, dtype=np.float64)[None, None, :]
    Ta = 2.0 + np.sin(0.37 * ii + 0.2) * np.cos(0.23 * jj) + 0.5 * np.cos(0.31 * kk + 0.1 * ii) + 0.0 * jj
    Ta[5:9, 4:8, 3:6] = 0.0
    Ta[20:26, 10:15, :] = 1.5
    Ta = np.asfortranarray(Ta)
    oHz = np.zeros((nis, njs, N), order="F")
    ia, ib = max(b.LBi, IminS), min(b.UBi, ImaxS)
    ja, jb = max(b.LBj, JminS), min(b.UBj, JmaxS)
    hz = st["Hz"][ia - b.LBi:ib - b.LBi + 1, ja - b.LBj:jb - b.LBj + 1, :]
    oHz[ia - IminS:ib - IminS + 1, ja - JminS:jb - JminS + 1, :] = 1.0 / np.where(hz > 0.0, hz, 1.0)
    t3 = st["t"][:, :, :, 2, 0]
    assert t3.flags.f_contiguous
    return oHz, Ta, t3


def oracle_mpdata_adiff(st, oHz, Ta, t3):
    """Run the C oracle's mpdata_adiff on private arrays; returns (Ta, Ua, Va, Wa)."""
    import ctypes as C
    import oracle
    from roms_trunk_mgh_amd import abi
    nis, njs, N = Ta.shape
    Ta = Ta.copy(order="F")
    Ua = np.zeros((nis, njs, N), order="F")
    Va = np.zeros((nis, njs, N), order="F")
    Wa = np.zeros((nis, njs, N + 1), order="F")
    lib = oracle.lib()
    lib.oracle_mpdata_adiff.argtypes = [C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.StepIdx),
                                        C.POINTER(abi.Fields)] + [C.c_void_p] * 6
    F = st.fields_struct()
    s = step_idx()
    rc = lib.oracle_mpdata_adiff(C.byref(st.b), C.byref(st.p), C.byref(s), C.byref(F), oHz.ctypes.data,
                                 t3.ctypes.data, Ta.ctypes.data, Ua.ctypes.data, Va.ctypes.data, Wa.ctypes.data)
    assert rc == 0
    return Ta, Ua, Va, Wa


def river_sources(st, kind="walls", same_tracer=False, seed=3):
    """A point-source table for the state (LuvSrc; roms_trunk_mgh_amd/sources.py) and the switch in the parameters
    (a private copy of them).  kind = "walls": faces of the closed walls -- an inflow through the southern wall, and in a
    basin an inflow through the western wall and an outflow (a sink) through the eastern one; "coast": faces of the land
    mask -- one u-face and one v-face with land behind them, flowing into the water; "wells": cell-centred sources
    (LwSrc, Dsrc = 2) -- three inflows and a sink; "wells_dup": two of them share a cell (the reference then counts
    both in the free surface and the tracers but only the later one in omega -- as written, omega.F:173-190; for the
    device-against-oracle tests); "all" = walls + wells (+ coast with a mask).  Qshape grows towards the surface;
    the first tracer comes with the river (LtracerSrc), the others do not -- unless same_tracer: then every tracer does,
    with the value `same_tracer` (a river of ambient water)."""
    from roms_trunk_mgh_amd import sources
    b = st.b
    N, NT = b.N, b.NT
    # moves about a cell's volume in 250 steps; from the grid's nominal numbers, the same for every tiling
    q0 = {"UPWELLING": 1.3e3 * 41 * 80, "SEAMOUNT": 1.3e7 * 49 * 48, "BENCHMARK": 1.0e9 * 64 * 32}[st.cfg["app"]] / (b.Lm * b.Mm)
    I, J, D, Q = [], [], [], []
    if kind in ("walls", "both", "all"):
        I.append(max(2, b.Lm // 3)); J.append(1); D.append(1.0); Q.append(q0)               # southern wall, northward
        if not b.EWperiodic:
            I.append(1); J.append(max(2, b.Mm // 2)); D.append(0.0); Q.append(0.7 * q0)      # western wall, eastward
            I.append(b.Lm + 1); J.append(max(2, b.Mm // 4)); D.append(0.0); Q.append(0.4 * q0)   # eastern wall: a sink
    if kind in ("wells", "wells_dup", "all"):
        ci, cj = max(3, (2 * b.Lm) // 5), max(3, (2 * b.Mm) // 3)
        second = (0, 0) if kind == "wells_dup" else (1, 2)
        for di, dj, q in ((0, 0, 0.5 * q0), (second[0], second[1], 0.2 * q0), (3, -2, 0.3 * q0), (-4, 3, -0.25 * q0)):
            I.append(ci + di); J.append(cj + dj); D.append(2.0); Q.append(q)
    if kind in ("coast", "both") or (kind == "all" and st.p.masking):
        # the whole grid's mask (the same faces for every tiling)
        bg = ana.make_bounds(b.Lm, b.Mm, N, NT, b.NAT, 1, 1, 0, EWperiodic=bool(b.EWperiodic), NSperiodic=False,
                             NghostPoints=b.NghostPoints)
        rm = ana.island_mask(st.cfg, bg)
        done_u = done_v = False
        for j in range(3, b.Mm - 1):
            for i in range(3, b.Lm - 1):
                a, w, s = rm[i - bg.LBi, j - bg.LBj], rm[i - 1 - bg.LBi, j - bg.LBj], rm[i - bg.LBi, j - 1 - bg.LBj]
                if not done_u and a == 1.0 and w == 0.0:
                    I.append(i); J.append(j); D.append(0.0); Q.append(q0); done_u = True
                elif not done_v and a == 0.0 and s == 1.0 and j > b.Mm // 2:
                    I.append(i); J.append(j); D.append(1.0); Q.append(-0.6 * q0); done_v = True     # southward, into the water
        assert done_u and done_v
    n = len(I)
    assert n > 0, kind
    w = np.linspace(1.0, 3.0, N)
    Qshape = np.tile(w / w.sum(), (n, 1))
    Tsrc = np.zeros((n, N, NT))
    rng = np.random.default_rng(seed)
    for it in range(NT):
        Tsrc[:, :, it] = same_tracer if same_tracer else (4.0 + 2.0 * it + 0.5 * rng.random((n, N)))
    ltr = np.ones(NT, dtype=np.int32) if same_tracer else np.array([1] + [0] * (NT - 1), dtype=np.int32)
    st.sources = sources.Sources(I, J, D, Q, Qshape, Tsrc, ltr)
    st.p = type(st.p).from_buffer_copy(st.p)
    st.p.point_sources = (1 if any(d < 2.0 for d in D) else 0) | (2 if any(d == 2.0 for d in D) else 0)
    return st.sources
