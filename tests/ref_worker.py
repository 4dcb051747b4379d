"""Child process of tests/test_ref_pinning.py: loads oracle/_ref/<APP>/libref.so (the
reference's own Fortran, compiled from /root/reference by oracle/build_ref.sh) and
compares it with the C oracle on the same seeded inputs.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def atm_pressure(st):
    """A surface air-pressure field (mb) with gradients in both directions, for the ATM_PRESS cases."""
    b = st.b
    ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None]
    jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :]
    st["Pair"][:] = 1013.25 + 6.0 * np.sin(2.0 * np.pi * 2.0 * (ii - 0.5) / b.Lm) * np.cos(np.pi * (jj - 0.5) / b.Mm) + 0.02 * jj


def main(config, mask=None, pgf=0, basin=False, atm=False):
    import oracle
    import util
    from oracle import ref
    ov = {"tnu2": 300.0, "visc2": 800.0} if config != "SEAMOUNT" else {"tnu2": 300.0}
    if basin:                                # no periodic direction: western / eastern edges closed as well
        ov["EWperiodic"] = False
    if atm:                                  # ATM_PRESS builds (oracle/_ref/<APP>_ATM...)
        ov["atm_press"] = 1
    st0 = util.prepared_state(config, overrides=ov, mask=mask)
    if atm:
        atm_pressure(st0)
    st0.p.pgf = pgf                      # 1, 2: the reference built with prsgrd31.h (plain / WJ_GRADP)
    if util.WET:                         # set_depth.F:168-172: a bed at the resting level is lifted by 1e-14
        st0["h"][7 - st0.b.LBi, 9 - st0.b.LBj] = 0.0
        st0["h"][12 - st0.b.LBi, 3 - st0.b.LBj] = 0.0
    out = {"pgf": int(st0.p.pgf), "EWperiodic": int(st0.b.EWperiodic)}
    r = ref.Ref(st0.copy())
    bb = r.bounds()
    mine = st0.b.as_dict()
    out["bounds_mismatch"] = {k: (v, mine[k]) for k, v in bb.items() if mine[k] != v}
    nf, w1, w2 = r.set_weights(st0.p.ndtfast)
    n2 = 2 * st0.p.ndtfast
    out["nfast"] = [nf, st0.p.nfast]
    out["weights_maxdiff"] = float(max(max(abs(w1[i] - st0.p.weight1[i]), abs(w2[i] - st0.p.weight2[i])) for i in range(n2)))
    kernels = ["set_depth", "set_massflux", "set_zeta", "rho_eos", "prsgrd", "t3dmix2"]
    if config != "SEAMOUNT":
        kernels.append("uv3dmix2")
    s = util.step_idx()
    out["kernels"] = {}
    out["masking"] = int(st0.p.masking)
    for k in kernels:
        st_r, st_o = st0.copy(), st0.copy()
        # detune so that every kernel has something to do
        for st in (st_r, st_o):
            st["Zt_avg1"] *= 1.3
            st["u"] *= 1.1
        rr = ref.Ref(st_r)
        rr.call(k, s)
        oracle.Oracle(st_o).call(k, s)
        diffs = util.compare_states(st_o, st_r)
        changed = util.compare_states(st_r, st0)
        out["kernels"][k] = {"max_rel_diff": max(diffs.values()) if diffs else 0.0, "fields_diff": sorted(diffs),
                             "changed": sorted(changed)}
    print(json.dumps(out))


DIF4 = {"ts_dif4": 1, "uv_vis4": 1, "tnu4": 2.0e7, "visc4": 4.0e7}
# "... stab": the dif4 / iso comparisons against the builds with -DTS_MIX_STABILITY, t3dmix2 beside t3dmix4, with two
# distinct time levels nrhs = 3 and nstp = 1 (the model's own sequence has nrhs = nstp there, main3d.F:191, which would
# leave the 1/4 part untested)
STAB = False
MINSTRAT = False          # "... minstrat": the iso modes against the builds with -DTS_MIX_MIN_STRAT


def mix_step_idx():
    import util
    return util.step_idx(nstp=1, nnew=2, nrhs=3) if STAB else util.step_idx()


def stab_effect(st0, st_o, k, s):
    """How far the TS_MIX_STABILITY / TS_MIX_MIN_STRAT result lies from the plain operator's on the same state (the
    option has to act)."""
    import oracle
    import util
    st_p = st0.copy()
    st_p.p = type(st0.p).from_buffer_copy(st0.p)        # copy() shares the parameter block
    st_p.p.ts_mix_stability = 0
    st_p.p.ts_mix_min_strat = 0
    oracle.Oracle(st_p).call(k, s)
    return util.max_rel_diff(st_o["t"], st_p["t"])


def main_dif4(config, basin=None, mask=None):
    """The biharmonic operators t3dmix4 (UPWELLING: along s-surfaces, SEAMOUNT: along geopotentials) and uv3dmix4
    (along s-surfaces): reference Fortran (built with TS_DIF4 and UV_VIS4 added) vs C oracle.  basin = "closed" /
    "open": no periodic direction -- the rule for the first operator's result on the four edges and the corners
    (zero / gamma2-slip where the variable's condition is closed, a copy / zero otherwise)."""
    import oracle
    import util
    from oracle import ref
    from roms_trunk_mgh_amd import abi
    ov = dict(DIF4, ts_mix_stability=int(STAB))
    if STAB:
        ov["tnu2"] = 300.0                       # t3dmix2 is compared as well: a harmonic coefficient that acts
    if basin:
        ov["EWperiodic"] = False
    st0 = util.prepared_state(config, overrides=ov, mask=mask)
    assert st0.b.NghostPoints == 3 and st0.p.ts_dif4 == 1 and st0.p.uv_vis4 == 1 and st0.p.ts_mix_stability == int(STAB)
    if basin == "open":
        for sd in ("west", "east", "south", "north"):
            for var in ("u", "v", "t"):
                st0.p.lbc[abi.LBS[sd]][abi.LBV[var]] = abi.LBC["Gra"]
    out = {"EWperiodic": int(st0.b.EWperiodic), "masking": int(st0.p.masking), "kernels": {}}
    s = mix_step_idx()
    for k in (("t3dmix2", "t3dmix4") if STAB else ("t3dmix4", "uv3dmix4")):
        st_r, st_o = st0.copy(), st0.copy()
        ref.Ref(st_r).call(k, s)
        oracle.Oracle(st_o).call(k, s)
        diffs = util.compare_states(st_o, st_r)
        changed = util.compare_states(st_r, st0)
        out["kernels"][k] = {"max_rel_diff": max(diffs.values()) if diffs else 0.0, "fields_diff": sorted(diffs),
                             "changed": sorted(changed),
                             "change": max(util.max_rel_diff(st_r[n], st0[n]) for n in ("t", "u", "v"))}
        if STAB:
            out["kernels"][k]["stab_effect"] = stab_effect(st0, st_o, k, s)
    print(json.dumps(out))


def iso_state(config, basin=None, mask=None, extra=None):
    """prepared_state with MIX_ISO_TS (and the biharmonic options): potential density from the (pinned) rho_eos, made
    weakly stratified in one band of columns and strongly in another so that both branches of
    MAX(pden(k) - pden(k+1), eps) are taken."""
    import oracle
    import util
    from roms_trunk_mgh_amd import abi
    ov = dict(DIF4, mix_iso_ts=1, tnu2=300.0, ts_mix_stability=int(STAB), ts_mix_min_strat=int(MINSTRAT))
    if extra:
        ov.update(extra)
    if basin:
        ov["EWperiodic"] = False
    st0 = util.prepared_state(config, overrides=ov, mask=mask)
    oracle.Oracle(st0).call("rho_eos", util.step_idx())
    b = st0.b
    pd = st0["pden"]
    mid = pd[:, :, b.N // 2][:, :, None]
    band = slice(b.Lm // 3 - b.LBi, b.Lm // 2 - b.LBi)
    pd[band] = mid[band] + 0.02 * (pd[band] - mid[band])
    strong = slice(b.Lm // 2 - b.LBi, 3 * b.Lm // 4 - b.LBi)           # ... and strongly (differences above eps = 0.5) in another
    pd[strong] = mid[strong] + 60.0 * (pd[strong] - mid[strong])
    if basin == "open":
        for sd in ("west", "east", "south", "north"):
            st0.p.lbc[abi.LBS[sd]][abi.LBV["t"]] = abi.LBC["Gra"]
    return st0


def main_iso(config, basin=None, mask=None):
    """t3dmix2_iso and t3dmix4_iso (MIX_ISO_TS): reference Fortran (the application with MIX_ISO_TS, TS_DIF2 and
    TS_DIF4) vs C oracle."""
    import oracle
    import util
    from oracle import ref
    st0 = iso_state(config, basin, mask)
    assert st0.p.mix_iso_ts == 1 and st0.p.mix_geo_ts == 0 and st0.p.mix_s_ts == 0
    d = st0["pden"][:, :, :-1] - st0["pden"][:, :, 1:]
    out = {"EWperiodic": int(st0.b.EWperiodic), "masking": int(st0.p.masking), "kernels": {},
           "frac_below_eps": float((d < 0.5).mean())}
    s = mix_step_idx()
    for k in ("t3dmix2", "t3dmix4"):
        st_r, st_o = st0.copy(), st0.copy()
        ref.Ref(st_r).call(k, s)
        oracle.Oracle(st_o).call(k, s)
        diffs = util.compare_states(st_o, st_r)
        out["kernels"][k] = {"max_rel_diff": max(diffs.values()) if diffs else 0.0, "fields_diff": sorted(diffs),
                             "changed": sorted(util.compare_states(st_r, st0)),
                             "change": util.max_rel_diff(st_r["t"], st0["t"])}
        if STAB or MINSTRAT:
            out["kernels"][k]["stab_effect"] = stab_effect(st0, st_o, k, s)
    print(json.dumps(out))


def shallow_edges(st):
    """WET_DRY boundary-condition cases: a bed so shallow on stretches of the boundary rows / columns (and the two rows
    inside) that the boundary free surface falls below Dcrit - h there (zetabc.F:733-827) while h + zeta stays
    positive for the Chapman / Flather / Shchepetkin square roots."""
    b = st.b
    h = st["h"]
    ii = np.arange(b.LBi, b.UBi + 1)[:, None] * np.ones((1, st.nj), dtype=int)
    jj = np.arange(b.LBj, b.UBj + 1)[None, :] * np.ones((st.ni, 1), dtype=int)
    edge = (jj <= b.Jstr + 1) | (jj >= b.Jend - 1)
    if not b.EWperiodic:
        edge |= (ii <= b.Istr + 1) | (ii >= b.Iend - 1)
    stretch = ((ii // 5) % 2 == 0) & ((jj // 4) % 3 != 1)
    h[edge & stretch] = 0.16               # zeta of prepared_state lies within +-0.13: 0 < h + zeta, zeta <= Dcrit - h on a part


def main_bc(config, mask=None, rad2d=False, pc=False):
    """The six lateral boundary-condition routines on the S/N edges, every condition the library offers, for the
    three states of the barotropic stepping (first, predictor, corrector): reference Fortran vs C oracle."""
    import oracle
    import util
    from oracle import ref
    from roms_trunk_mgh_amd import abi
    st0 = util.prepared_state(config, mask=mask)
    if util.WET:
        shallow_edges(st0)
    st0.p.radiation_2d = int(rad2d)
    if pc:
        st0.p.atm_press = st0.p.press_compensate = 1
        atm_pressure(st0)
    rng = np.random.default_rng(11)
    for name in ("zeta_bry", "ubar_bry", "vbar_bry", "u_bry", "v_bry"):
        st0[name][:] = 1.0e-2 * rng.standard_normal(st0[name].shape)
    st0["t_bry"][:] = st0["t"][:, :, :, 0, :] * (1.0 + 1.0e-3 * rng.standard_normal(st0["t_bry"].shape))
    # boundary rows that do not already satisfy any of the conditions
    b = st0.b
    for name in ("zeta", "ubar", "vbar", "u", "v", "t"):
        a = st0[name]
        for j in (b.Jstr - 1, b.Jstr, b.Jend + 1):
            row = a[:, j - b.LBj]
            row += 1.0e-3 * (1.0 + np.abs(row)) * rng.standard_normal(row.shape)
    out = {"masking": int(st0.p.masking), "cases": {}}
    table = {"zetabc": ("zeta", ["Clo", "Gra", "Cla", "Cha", "Che", "Rad", "RadNud"]), "u2dbc": ("ubar", ["Clo", "Gra", "Cla", "Fla", "Shc", "Red", "RedAcq", "Rad", "RadNud"]),
             "v2dbc": ("vbar", ["Clo", "Gra", "Cla", "Fla", "Shc", "Red", "RedAcq", "Rad", "RadNud"]), "u3dbc": ("u", ["Clo", "Gra", "Cla", "Rad", "RadNud"]),
             "v3dbc": ("v", ["Clo", "Gra", "Cla", "Rad", "RadNud"]), "t3dbc": ("t", ["Clo", "Gra", "Cla", "Rad", "RadNud"])}
    steps = [util.step_idx(iic=5, iif=1, pred=1, kstp=1, krhs=1, knew=3), util.step_idx(iic=5, iif=3, pred=1, kstp=2, krhs=1, knew=3),
             util.step_idx(iic=5, iif=3, pred=0, kstp=1, krhs=3, knew=2)]
    for kind, (var, codes) in table.items():
        for code in codes:
            for q, s in enumerate(steps if kind in ("zetabc", "u2dbc", "v2dbc") else steps[:1]):
                st_r, st_o = st0.copy(), st0.copy()
                for st in (st_r, st_o):
                    st.p = type(st0.p).from_buffer_copy(st0.p)
                    for sd in ("south", "north"):
                        st.p.lbc[abi.LBS[sd]][abi.LBV[var]] = abi.LBC["Red" if code == "RedAcq" else code]
                        if code == "RedAcq":     # reduced physics with free-surface boundary data (zeta clamped)
                            st.p.lbc[abi.LBS[sd]][abi.LBV["zeta"]] = abi.LBC["Cla"]
                        st.p.obc_out[abi.LBS[sd]][abi.LBV[var]] = 2.0e-4      # RadNud: passive / active nudging (1/s)
                        st.p.obc_in[abi.LBS[sd]][abi.LBV[var]] = 1.5e-3
                nout = s.knew if kind in ("zetabc", "u2dbc", "v2dbc") else s.nnew
                itrc = st0.b.NT
                ref.Ref(st_r).bc(kind, s, nout, itrc)
                oracle.Oracle(st_o).bc(kind, s, nout, itrc)
                diffs = util.compare_states(st_o, st_r)
                changed = util.compare_states(st_r, st0)
                out["cases"][f"{kind}:{code}:{q}"] = {"max_rel_diff": max(diffs.values()) if diffs else 0.0,
                                                     "changed": sorted(changed)}
    print(json.dumps(out))


def basin_state(config, mask=None):
    """prepared_state of `config` as a basin: no periodic direction, physical edges on all four sides (LBC of
    the western / eastern edge = closed unless a case sets lbc), boundary data and boundary lines / corners that
    satisfy none of the conditions already."""
    import util
    st0 = util.prepared_state(config, overrides={"EWperiodic": False}, mask=mask)
    if util.WET:
        shallow_edges(st0)
    rng = np.random.default_rng(17)
    for name in ("zeta_bry", "ubar_bry", "vbar_bry", "u_bry", "v_bry"):
        st0[name][:] = 1.0e-2 * rng.standard_normal(st0[name].shape)
    st0["t_bry"][:] = st0["t"][:, :, :, 0, :] * (1.0 + 1.0e-3 * rng.standard_normal(st0["t_bry"].shape))
    b = st0.b
    for name in ("zeta", "ubar", "vbar", "u", "v", "t"):
        a = st0[name]
        for j in (b.Jstr - 1, b.Jstr, b.Jend + 1):
            row = a[:, j - b.LBj]
            row += 1.0e-3 * (1.0 + np.abs(row)) * rng.standard_normal(row.shape)
        for i in (b.Istr - 1, b.Istr, b.Iend + 1):
            col = a[i - b.LBi]
            col += 1.0e-3 * (1.0 + np.abs(col)) * rng.standard_normal(col.shape)
    return st0


BC_TABLE = {"zetabc": ("zeta", ["Clo", "Gra", "Cla", "Cha", "Che", "Rad", "RadNud"]), "u2dbc": ("ubar", ["Clo", "Gra", "Cla", "Fla", "Shc", "Red", "RedAcq", "Rad", "RadNud"]),
            "v2dbc": ("vbar", ["Clo", "Gra", "Cla", "Fla", "Shc", "Red", "RedAcq", "Rad", "RadNud"]), "u3dbc": ("u", ["Clo", "Gra", "Cla", "Rad", "RadNud"]),
            "v3dbc": ("v", ["Clo", "Gra", "Cla", "Rad", "RadNud"]), "t3dbc": ("t", ["Clo", "Gra", "Cla", "Rad", "RadNud"])}


def basin_cases(st0):
    """(key, kind, variable, state, step indices, nout, itrc): every condition on all four edges at once (the
    western / eastern edges then run the transposed code of each routine, and the four corners are set)."""
    import util
    from roms_trunk_mgh_amd import abi
    steps = [util.step_idx(iic=5, iif=1, pred=1, kstp=1, krhs=1, knew=3), util.step_idx(iic=5, iif=3, pred=1, kstp=2, krhs=1, knew=3),
             util.step_idx(iic=5, iif=3, pred=0, kstp=1, krhs=3, knew=2)]
    for kind, (var, codes) in BC_TABLE.items():
        for code in codes:
            for q, s in enumerate(steps if kind in ("zetabc", "u2dbc", "v2dbc") else steps[:1]):
                st = st0.copy()
                st.p = type(st0.p).from_buffer_copy(st0.p)
                for sd in ("west", "east", "south", "north"):
                    st.p.lbc[abi.LBS[sd]][abi.LBV[var]] = abi.LBC["Red" if code == "RedAcq" else code]
                    if code == "RedAcq":
                        st.p.lbc[abi.LBS[sd]][abi.LBV["zeta"]] = abi.LBC["Cla"]
                    st.p.obc_out[abi.LBS[sd]][abi.LBV[var]] = 2.0e-4          # RadNud: passive / active nudging (1/s)
                    st.p.obc_in[abi.LBS[sd]][abi.LBV[var]] = 1.5e-3
                nout = s.knew if kind in ("zetabc", "u2dbc", "v2dbc") else s.nnew
                yield f"{kind}:{code}:{q}", kind, var, st, s, nout, st0.b.NT


def main_bc4(config, mask=None, rad2d=False, pc=False):
    """The six boundary-condition routines on a basin (four physical edges + corners): reference vs oracle."""
    import oracle
    import util
    from oracle import ref
    st0 = basin_state(config, mask)
    st0.p.radiation_2d = int(rad2d)
    if pc:
        st0.p.atm_press = st0.p.press_compensate = 1
        atm_pressure(st0)
    out = {"masking": int(st0.p.masking), "EWperiodic": int(st0.b.EWperiodic), "cases": {}, "radiation_2d": int(rad2d)}
    bb = ref.Ref(st0.copy()).bounds()
    mine = st0.b.as_dict()
    out["bounds_mismatch"] = {k: (v, mine[k]) for k, v in bb.items() if mine[k] != v}
    for key, kind, var, st, s, nout, itrc in basin_cases(st0):
        st_r, st_o = st.copy(), st
        ref.Ref(st_r).bc(kind, s, nout, itrc)
        oracle.Oracle(st_o).bc(kind, s, nout, itrc)
        diffs = util.compare_states(st_o, st_r)
        b = st0.b
        a_r, a_0 = st_r[var], st0[var]
        # all four edges and all four corners must have been written
        iw = b.Istr if var in ("ubar", "u") else b.Istr - 1
        js = b.Jstr if var in ("vbar", "v") else b.Jstr - 1
        lines = {"west": (a_r[iw - b.LBi, 3 - b.LBj:6 - b.LBj], a_0[iw - b.LBi, 3 - b.LBj:6 - b.LBj]),
                 "east": (a_r[b.Iend + 1 - b.LBi, 3 - b.LBj:6 - b.LBj], a_0[b.Iend + 1 - b.LBi, 3 - b.LBj:6 - b.LBj]),
                 "south": (a_r[3 - b.LBi:6 - b.LBi, js - b.LBj], a_0[3 - b.LBi:6 - b.LBi, js - b.LBj]),
                 "north": (a_r[3 - b.LBi:6 - b.LBi, b.Jend + 1 - b.LBj], a_0[3 - b.LBi:6 - b.LBi, b.Jend + 1 - b.LBj]),
                 "corner_sw": (a_r[iw - b.LBi, js - b.LBj], a_0[iw - b.LBi, js - b.LBj]),
                 "corner_ne": (a_r[b.Iend + 1 - b.LBi, b.Jend + 1 - b.LBj], a_0[b.Iend + 1 - b.LBi, b.Jend + 1 - b.LBj])}
        out["cases"][key] = {"max_rel_diff": max(diffs.values()) if diffs else 0.0,
                             "unchanged": [k for k, (x, y) in lines.items() if np.array_equal(x, y)]}
    print(json.dumps(out))


INI_TABLES = {"closed": None,
              "cha_fla_rad": {"zeta": "Cha", "ubar": "Fla", "vbar": "Fla", "u": "Rad", "v": "Rad", "t": "Rad"},
              "gradient": {v: "Gra" for v in ("zeta", "ubar", "vbar", "u", "v", "t")},
              "clamped": {v: "Cla" for v in ("zeta", "ubar", "vbar", "u", "v", "t")},
              "radiation": {v: "Rad" for v in ("zeta", "ubar", "vbar", "u", "v", "t")}}


def ini_cases(config, mask):
    """(key, state, step indices) of the first-step initialisation cases: every boundary-condition table, with the
    time indices of the reference's first step (kstp = knew = 1, nstp = 1, nnew = 2; initial.F:133-143,
    main3d.F:189-191) and with knew = 2 (the routine's other branch of "knew /= kstp")."""
    import util
    from roms_trunk_mgh_amd import abi
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_bc import input_state
    st0 = input_state(config, mask)
    if util.WET:
        shallow_edges(st0)
    rng = np.random.default_rng(23)
    st0["u"][:, :, :, 0] += 0.01 * rng.standard_normal(st0["u"][:, :, :, 0].shape)      # so that ubar, vbar change
    st0["v"][:, :, :, 0] += 0.01 * rng.standard_normal(st0["v"][:, :, :, 0].shape)
    for name, table in INI_TABLES.items():
        for knew in (1, 2):
            st = st0.copy()
            st.p = type(st0.p).from_buffer_copy(st0.p)
            if table:
                for sd in ("south", "north"):
                    for var, code in table.items():
                        st.p.lbc[abi.LBS[sd]][abi.LBV[var]] = abi.LBC[code]
            s = util.step_idx(iic=1, iif=1, pred=0, kstp=1, krhs=1, knew=knew)
            s.nstp, s.nnew, s.nrhs = 1, 2, 1
            yield f"{name}:{knew}", st, s


def main_ini(config, mask=None):
    """ini_zeta + ini_fields (ini_fields.F; main3d.F:269-283): reference Fortran vs C oracle."""
    import oracle
    import util
    from oracle import ref
    out = {"cases": {}}
    for key, st, s in ini_cases(config, mask):
        out["masking"] = int(st.p.masking)
        st0, st_r, st_o = st.copy(), st.copy(), st
        for kind in ("ini_zeta", "ini_fields"):
            ref.Ref(st_r).bc(kind, s, 0, 0)
            oracle.Oracle(st_o).call(kind, s)
        diffs = util.compare_states(st_o, st_r)
        changed = util.compare_states(st_r, st0)
        out["cases"][key] = {"max_rel_diff": max(diffs.values()) if diffs else 0.0, "changed": sorted(changed)}
    print(json.dumps(out))


def main_mpdata(config, mask=None, basin=None):
    """mpdata_adiff_tile: reference Fortran vs C oracle on the same private arrays, with the
    3-ghost-point bounds an MPDATA run uses (also pins get_bounds for NghostPoints = 3).  mask = "island": the
    MASKING build (face masks in the cross terms, land out of the limiter's extrema, masked transports).
    basin = "closed" / "open": no periodic direction -- the boundary values and corners of Ta, and the wall rule of
    Ua / Va on the four edges: zero where the 3-D momentum's condition is closed, the neighbour's value otherwise."""
    import util
    from oracle import ref
    from roms_trunk_mgh_amd import abi
    ov = {"Hadv": "MPDATA", "Vadv": "MPDATA"}
    if basin:
        ov["EWperiodic"] = False
    st = util.prepared_state(config, overrides=ov, mask=mask)
    if basin == "open":
        for sd in ("west", "east", "south", "north"):
            for var in ("u", "v"):
                st.p.lbc[abi.LBS[sd]][abi.LBV[var]] = abi.LBC["Gra"]
    b = st.b
    out = {"NghostPoints": int(b.NghostPoints), "masking": int(st.p.masking)}
    r = ref.Ref(st)
    bb = r.bounds()
    mine = b.as_dict()
    out["bounds_mismatch"] = {k: (v, mine[k]) for k, v in bb.items() if mine[k] != v}
    oHz, Ta0, t3 = util.mpdata_private_arrays(st)
    nis, njs, N = Ta0.shape
    Ta = Ta0.copy(order="F")
    Ua = np.zeros((nis, njs, N), order="F")
    Va = np.zeros((nis, njs, N), order="F")
    Wa = np.zeros((nis, njs, N + 1), order="F")
    r.mpdata_adiff(oHz, t3, Ta, Ua, Va, Wa)
    res = {"ref": dict(Ta=Ta, Ua=Ua, Va=Va, Wa=Wa)}
    res["oracle"] = dict(zip(("Ta", "Ua", "Va", "Wa"), util.oracle_mpdata_adiff(st, oHz, Ta0, t3)))
    out["diff"] = {k: float(np.abs(res["ref"][k] - res["oracle"][k]).max()) for k in ("Ta", "Ua", "Va", "Wa")}
    out["nonzero"] = {k: int(np.count_nonzero(res["ref"][k])) for k in ("Ua", "Va", "Wa")}
    out["amax"] = {k: float(np.abs(res["ref"][k]).max()) for k in ("Ua", "Va", "Wa")}
    print(json.dumps(out))


def logdrag_state(config="UPWELLING"):
    """prepared_state with UV_LOGDRAG: a roughness length varying over two decades, so large in one corner that the drag
    coefficient runs into Cdb_max, so small in another that it runs into a raised Cdb_min."""
    import util
    st = util.prepared_state(config, overrides={"uv_drag": 3})
    b = st.b
    ii = np.arange(b.LBi, b.UBi + 1, dtype=np.float64)[:, None] / b.Lm
    jj = np.arange(b.LBj, b.UBj + 1, dtype=np.float64)[None, :] / b.Mm
    st["ZoBot"][:] = 1.0e-4 * 10.0 ** (2.0 * ii + 1.5 * jj)
    st["ZoBot"][-6:, -6:] = 0.6 * (st["z_r"][-6:, -6:, 0] - st["z_w"][-6:, -6:, 0])
    st.p.Cdb_min = 1.5e-3
    return st


def main_emp(config, mask=None):
    """bulk_flux with EMINUSP (bulk_flux.F:883-899; BENCHMARK built with -DEMINUSP): evap and the surface salt flux."""
    import oracle
    import util
    from oracle import ref
    st0 = util.prepared_state(config, overrides={"eminusp": 1}, mask=mask)
    st0["rain"] += 2.0e-5
    s = util.step_idx()
    st_r, st_o = st0.copy(), st0.copy()
    ref.Ref(st_r).physics("bulk_flux", s)
    oracle.Oracle(st_o).call("bulk_flux", s)
    names = ["evap", "stflux", "lhflx", "sustr"]
    out = {"masking": int(st0.p.masking),
           "bulk_flux": {"diffs": {n: util.max_rel_diff(st_o[n], st_r[n]) for n in names},
                         "changed": [n for n in names if not np.array_equal(st_r[n], st0[n])],
                         "amax_evap": float(np.abs(st_r["evap"]).max()),
                         "amax_saltflux": float(np.abs(st_r["stflux"][:, :, 1]).max())}}
    out["bulk_flux"]["max_rel_diff"] = max(out["bulk_flux"]["diffs"].values())
    print(json.dumps(out))


def main_limbs(config):
    """set_vbc with LIMIT_BSTRESS (set_vbc.F:533-567; the application built with -DLIMIT_BSTRESS): a strong drag so
    that the limit is reached at part of the points."""
    import oracle
    import util
    from oracle import ref
    st0 = util.prepared_state(config, overrides={"limit_bstress": 1})
    st0["rdrag"] *= 25.0 * (1.0 + 3.0 * np.linspace(0.0, 1.0, st0["rdrag"].shape[0]))[:, None]
    st0["rdrag2"] *= 4000.0
    s = util.step_idx()
    st_r, st_o, st_n = st0.copy(), st0.copy(), st0.copy()
    st_n.p = type(st0.p).from_buffer_copy(st0.p)
    st_n.p.limit_bstress = 0
    ref.Ref(st_r).physics("set_vbc", s)
    oracle.Oracle(st_o).call("set_vbc", s)
    oracle.Oracle(st_n).call("set_vbc", s)
    names = ["bustr", "bvstr"]
    out = {"set_vbc": {"diffs": {n: util.max_rel_diff(st_o[n], st_r[n]) for n in names},
                       "changed": [n for n in names if not np.array_equal(st_r[n], st0[n])]},
           "frac_limited": float((st_o.interior("bustr") != st_n.interior("bustr")).mean())}
    out["set_vbc"]["max_rel_diff"] = max(out["set_vbc"]["diffs"].values())
    print(json.dumps(out))


def main_logdrag(config):
    """set_vbc with UV_LOGDRAG (set_vbc.F:542-580): reference Fortran (UPWELLING with UV_LOGDRAG in the place of
    UV_LDRAG) vs C oracle; LOG comes from two math libraries, the relative difference is reported."""
    import oracle
    import util
    from oracle import ref
    st0 = logdrag_state(config)
    s = util.step_idx()
    st_r, st_o = st0.copy(), st0.copy()
    ref.Ref(st_r).physics("set_vbc", s)
    oracle.Oracle(st_o).call("set_vbc", s)
    names = ["bustr", "bvstr"]
    dz = st0["z_r"][:, :, 0] - st0["z_w"][:, :, 0]
    cd = (0.41 / np.log(dz / st0["ZoBot"])) ** 2
    out = {"set_vbc": {"diffs": {n: util.max_rel_diff(st_o[n], st_r[n]) for n in names},
                       "changed": [n for n in names if not np.array_equal(st_r[n], st0[n])],
                       "amax": {n: float(np.abs(st_r[n]).max()) for n in names}},
           "frac_at_max": float((cd > st0.p.Cdb_max).mean()), "frac_at_min": float((cd < st0.p.Cdb_min).mean())}
    out["set_vbc"]["max_rel_diff"] = max(out["set_vbc"]["diffs"].values())
    print(json.dumps(out))


def main_physics(config, mask=None, basin=False):
    """set_vbc (all applications) and bulk_flux (BENCHMARK: the BULK_FLUXES application): reference
    Fortran vs C oracle.  set_vbc has only +,*,sqrt: bit for bit; bulk_flux calls log/exp/pow/atan
    from two different math libraries: relative difference reported.  mask = "island": the MASKING builds."""
    import oracle
    import util
    from oracle import ref
    st0 = util.prepared_state(config, mask=mask, overrides={"EWperiodic": False} if basin else None)
    s = util.step_idx()
    out = {"masking": int(st0.p.masking), "EWperiodic": int(st0.b.EWperiodic)}
    kernels = ["set_vbc"] + (["bulk_flux", "lmd_vmix"] if config.startswith("BENCHMARK") else [])
    for k in kernels:
        st_r, st_o = st0.copy(), st0.copy()
        for st in (st_r, st_o):                     # so that every output changes
            st["stflux"][:, :, 0] += 1.0e-6
            st["Vwind"] += 0.3 * st["Uwind"] - 2.0          # both stress components, both signs
            st["rain"] += 2.0e-5                            # rain heat flux and rain stress terms
            if st.b.NT > 1:
                st["stflux"][:, :, 1] = 2.0e-8
                st["btflx"][:, :, 1] = 1.0e-9
        ref.Ref(st_r).physics(k, s)
        oracle.Oracle(st_o).call(k, s)
        names = {"set_vbc": ["stflx", "btflx", "bustr", "bvstr"],
                 "bulk_flux": ["sustr", "svstr", "lrflx", "lhflx", "shflx", "stflux"],
                 "lmd_vmix": ["Akv", "Akt", "ghats", "hsbl"]}[k]
        diffs = {n: util.max_rel_diff(st_o[n], st_r[n]) for n in names}
        changed = [n for n in names if not np.array_equal(st_r[n], st0[n])]
        out[k] = {"max_rel_diff": max(diffs.values()), "diffs": diffs, "changed": changed,
                  "amax": {n: float(np.abs(st_r[n]).max()) for n in names}}
        if mask:                                    # land points of the masked outputs are zero in the reference
            land = st0["rmask"] == 0.0
            out[k]["land_zero"] = bool(all(not st_r[n][land].any() for n in names
                                           if n in ("lrflx", "lhflx", "shflx", "hsbl")))
    if config.startswith("BENCHMARK") and not basin:
        # KPP on a stratified state (prepared_state leaves pden, bvf, alpha, beta zero): util.kpp_state
        st0 = util.kpp_state(config, mask=mask)
        st_r, st_o = st0.copy(), st0.copy()
        ref.Ref(st_r).physics("lmd_vmix", s)
        oracle.Oracle(st_o).call("lmd_vmix", s)
        names = ["Akv", "Akt", "ghats", "hsbl"]
        diffs = {n: util.max_rel_diff(st_o[n], st_r[n]) for n in names}
        hs, zw = st_r.interior("hsbl"), st_r.interior("z_w")
        wet = st_r.interior("rmask") > 0.0
        out["lmd_vmix_stratified"] = {
            "max_rel_diff": max(diffs.values()), "diffs": diffs,
            "changed": [n for n in names if not np.array_equal(st_r[n], st0[n])],
            "amax": {n: float(np.abs(st_r[n]).max()) for n in names},
            "frac_in_top_layer": float((hs[wet] > zw[:, :, -2][wet]).mean()),
            "frac_below_level_Nm3": float((hs[wet] < zw[:, :, -4][wet]).mean()),
            "land_zero": bool(not st_r["hsbl"][st0["rmask"] == 0.0].any()) if mask else None}
    print(json.dumps(out))


def main_gls(config, mask=None, basin=False):
    """gls_prestep_tile and gls_corstep_tile (with tkebc_tile): reference Fortran (the GLS builds) vs C oracle, for the
    four parameter sets of roms_*.in and both start-up branches.  The closure raises to real powers (pow of either
    build's math library): relative difference reported."""
    import oracle
    import util
    from oracle import ref
    out = {"cases": {}}
    names = ["tke", "gls", "Akv", "Akt", "Akk", "Akp", "Lscale"]
    for gset in ("k-epsilon", "k-kl", "k-omega", "gen"):
        st0 = util.gls_state(config, gls=gset, mask=mask, basin=basin)
        out["masking"], out["EWperiodic"] = int(st0.p.masking), int(st0.b.EWperiodic)
        for kernel in ("gls_prestep", "gls_corstep"):
            for iic in (1, 5):
                s = util.step_idx(iic=iic)
                st_r, st_o = st0.copy(), st0.copy()
                if kernel == "gls_corstep":                 # as gls_prestep leaves the nnew level: Hz-weighted
                    for st in (st_r, st_o):
                        hzw = np.zeros_like(st["Akv"])
                        hzw[:, :, 1:-1] = 0.5 * (st["Hz"][:, :, :-1] + st["Hz"][:, :, 1:])
                        hzw[:, :, 0] = hzw[:, :, 1]
                        hzw[:, :, -1] = hzw[:, :, -2]
                        for n in ("tke", "gls"):
                            st[n][:, :, :, s.nnew - 1] = hzw * st[n][:, :, :, s.nstp - 1]
                ref.Ref(st_r).gls(kernel, s)
                oracle.Oracle(st_o).call(kernel, s)
                diffs = {n: util.max_rel_diff(st_o[n], st_r[n]) for n in names}
                changed = [n for n in names if not np.array_equal(st_r[n], st0[n])]
                out["cases"][f"{gset}/{kernel}/iic{iic}"] = {
                    "max_rel_diff": max(diffs.values()), "diffs": diffs, "changed": changed,
                    "finite": bool(all(np.isfinite(st_r[n]).all() for n in names)),
                    "amax": {n: float(np.abs(st_r[n]).max()) for n in names}}
    print(json.dumps(out))


def main_diag(config):
    """wvelocity (bit for bit on wvel and on the exchanged DU_avg1/DV_avg1) and diag (the reference keeps
    only its printed report: compared at the printed 7 digits) -- reference Fortran vs C oracle."""
    import tempfile
    import oracle
    import util
    from oracle import ref
    st0 = util.prepared_state(config)
    s = util.step_idx()
    st_r, st_o = st0.copy(), st0.copy()
    R, O = ref.Ref(st_r), oracle.Oracle(st_o)
    tmp = tempfile.mkdtemp()
    R.diagnostics("wvelocity", s, tmp)
    O.call("wvelocity", s)
    out = {"wvel_diff": float(np.abs(st_r["wvel"] - st_o["wvel"]).max()), "wvel_amax": float(np.abs(st_r["wvel"]).max()),
           "DU_equal": bool(np.array_equal(st_r["DU_avg1"], st_o["DU_avg1"]) and np.array_equal(st_r["DV_avg1"], st_o["DV_avg1"]))}
    if True:
        d = R.diagnostics("diag", s, tmp)
        v = O.diag(s)
        mine = dict(avgke=v[1] / v[0], avgpe=v[2] / v[0], volume=v[0], Cu=v[6], Cv=v[7], Cw=v[8], maxspeed=v[3])
        out["diag_rel"] = {k: abs(mine[k] - d[k]) / abs(d[k]) for k in mine}
        out["diag_loc"] = [[int(v[9]), int(v[10]), int(v[11])], [d["Ci"], d["Cj"], d["Ck"]]]
    print(json.dumps(out))


def main_ana(config):
    """roms_trunk_mgh_amd/ana.py (the inputs of every test and of bench.py) against the reference's own analytic
    routines: ana_grid + metrics, set_scoord, ana_initial and the ana_* forcing of the application."""
    import util
    from oracle import ref
    from roms_trunk_mgh_amd import ana
    st = ana.make_tile(config, perturb=0.0)
    cfg = st.cfg
    cfg5 = [cfg["theta_s"], cfg["theta_b"], cfg["Tcline"], 4, 0.0]
    out = {}

    def cmp(names, a, b_, interior_only=False):
        d = {}
        for n in names:
            x, y = a[n], b_[n]
            if interior_only:
                x, y = a.interior(n), b_.interior(n)
            scale = max(float(np.abs(y).max()), 1e-300)
            d[n] = float(np.abs(x - y).max()) / scale
        return d

    # vertical coordinate
    r = ref.Ref(st.copy())
    sc = r.ana("scoord", cfg5)
    N = st.b.N
    out["scoord"] = {k: float(np.abs(np.array([getattr(st.p, k)[q] for q in range(N + 1)])[1 if k.endswith("_r") else 0:] -
                                     sc[k][1 if k.endswith("_r") else 0:]).max()) for k in ("sc_r", "Cs_r", "sc_w", "Cs_w")}
    out["hc"] = abs(sc["hc"] - st.p.hc)
    # grid + metrics
    st_r = st.copy()
    for n in GRID2D + ["dndx", "dmde"]:
        st_r[n][...] = -9.0e9
    ref.Ref(st_r).ana("grid", cfg5)
    # everything the reference defines: all columns up to Lm + NghostPoints, rows 0..Mm+1 (the arrays carry one
    # spare row / column of padding when Mm / Lm is even, mod_param.F initialize_param)
    b = st.b
    reg = (slice(0, b.Lm + b.NghostPoints - b.LBi + 1), slice(0 - b.LBj, b.Mm + 1 - b.LBj + 1))
    out["grid"] = {n: float(np.abs(st[n][reg] - st_r[n][reg]).max()) / max(float(np.abs(st_r[n][reg]).max()), 1e-300)
                   for n in GRID2D + (["dndx", "dmde"] if config.startswith("BENCHMARK") else [])}
    # initial conditions (on the z-levels set_depth gives for the state's zeta = 0)
    import oracle
    oracle.Oracle(st).call("set_depth", util.step_idx())
    st_r = st.copy()
    for n in ("zeta", "ubar", "vbar", "u", "v", "t"):
        st_r[n][...] = -9.0e9
    ref.Ref(st_r).ana("initial", cfg5)
    out["initial"] = {}
    for n in ("zeta", "ubar", "vbar", "u", "v", "t"):
        lev = 0
        x = st.interior(n)[..., lev] if n in ("zeta", "ubar", "vbar") else (st.interior(n)[..., lev] if n in ("u", "v") else st.interior(n)[..., lev, :])
        y = st_r.interior(n)[..., lev] if n in ("zeta", "ubar", "vbar") else (st_r.interior(n)[..., lev] if n in ("u", "v") else st_r.interior(n)[..., lev, :])
        out["initial"][n] = float(np.abs(x - y).max()) / max(float(np.abs(y).max()), 1e-300)
    # forcing
    names = ["Uwind", "Vwind", "Tair", "Pair", "Hair", "rain", "cloud"] if config.startswith("BENCHMARK") else ["sustr", "svstr"]
    st_r = st.copy()
    for n in names:
        st_r[n][...] = -9.0e9
    cfg5[4] = 3.0                                  # after the two-day ramp of the UPWELLING wind stress
    ref.Ref(st_r).ana("forcing", cfg5)
    out["forcing"] = cmp(names, st, st_r, interior_only=True)
    if config.startswith("BENCHMARK"):
        # ana_srflux (ALBEDO branch) at several times of day and of the year, and the host clock feeding it
        from roms_trunk_mgh_amd import main3d
        worst, clock = 0.0, 0.0
        for tdays in (0.0, 150.0 / 86400.0, 0.3, 0.5, 0.75, 10.4, 200.0 + 7350.0 / 86400.0):
            cfg5[4] = tdays
            s_r, s_o = st.copy(), st.copy()
            s_r["srflx"][...] = -9.0e9
            clk = ref.Ref(s_r).ana("srflux", cfg5)
            oracle.Oracle(s_o).ana_srflux(clk["yday"], clk["hour"])
            x, y = s_o.interior("srflx"), s_r.interior("srflx")
            worst = max(worst, float(np.abs(x - y).max()) / max(float(np.abs(y).max()), 1e-300))
            yd, hr = main3d.host_clock(tdays)
            clock = max(clock, abs(yd - clk["yday"]), abs(hr - clk["hour"]))
        out["srflux"] = worst
        out["host_clock"] = clock
    if config == "UPWELLING":            # ANA_VMIX
        for n in ("Akv", "Akt"):
            x, y = st.interior(n)[:, :, 1:-1], st_r.interior(n)[:, :, 1:-1]        # W-levels 1..N-1
            out["forcing"][n] = float(np.abs(x - y).max()) / max(float(np.abs(y).max()), 1e-300)
    if not config.startswith("BENCHMARK"):
        x, y = st.interior("stflux")[..., 0], st_r.interior("stflux")[..., 0]
        out["forcing"]["stflux_T"] = float(np.abs(x - y).max()) / max(float(np.abs(y).max()), 1e-300)
    print(json.dumps(out))


GRID2D = ["h", "f", "fomn", "pm", "pn", "om_r", "on_r", "om_u", "on_u", "om_v", "on_v", "om_p", "on_p", "omn",
          "pmon_r", "pnom_r", "pmon_p", "pnom_p", "pmon_u", "pnom_u", "pmon_v", "pnom_v"]


if __name__ == "__main__":
    if sys.argv[-1] == "wet":                  # ... wet: the same comparison on a WET_DRY state against the _WET builds
        import util as _util
        _util.WET = True
        sys.argv.pop()
    if sys.argv[-1] == "minstrat":             # ... minstrat: the iso modes against the _MINSTRAT builds
        MINSTRAT = True
        sys.argv.pop()
    if sys.argv[-1] == "stab":                 # ... stab: the dif4 / iso modes against the _STAB builds
        STAB = True
        sys.argv.pop()
    PC = len(sys.argv) > 3 and sys.argv[3] == "pc"            # ATM_PRESS + PRESS_COMPENSATE builds (bc, bc4 modes)
    RAD2D = len(sys.argv) > 3 and sys.argv[3] == "rad2d"      # the builds with -DRADIATION_2D (bc, bc4 modes)
    if len(sys.argv) > 2 and sys.argv[2] == "ana":
        main_ana(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] in ("gls", "gls_mask", "gls_basin"):
        main_gls(sys.argv[1], mask="island" if sys.argv[2] == "gls_mask" else None, basin=sys.argv[2] == "gls_basin")
    elif len(sys.argv) > 2 and sys.argv[2] == "diag":
        main_diag(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] in ("physics", "physics_mask", "physics_basin"):
        main_physics(sys.argv[1], mask="island" if sys.argv[2] == "physics_mask" else None,
                     basin=sys.argv[2] == "physics_basin")
    elif len(sys.argv) > 2 and sys.argv[2] == "basin":
        main(sys.argv[1], basin=True)
    elif len(sys.argv) > 2 and sys.argv[2] in ("emp", "emp_mask"):
        main_emp(sys.argv[1], mask="island" if sys.argv[2] == "emp_mask" else None)
    elif len(sys.argv) > 2 and sys.argv[2] == "limbs":
        main_limbs(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "logdrag":
        main_logdrag(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] in ("iso", "iso_closed", "iso_open", "iso_mask", "iso_mask_open"):
        m = sys.argv[2].split("_")
        main_iso(sys.argv[1], basin=m[-1] if m[-1] in ("closed", "open") else None, mask="island" if "mask" in m else None)
    elif len(sys.argv) > 2 and sys.argv[2] in ("dif4", "dif4_closed", "dif4_open", "dif4_mask", "dif4_mask_open"):
        m = sys.argv[2].split("_")
        main_dif4(sys.argv[1], basin=m[-1] if m[-1] in ("closed", "open") else None, mask="island" if "mask" in m else None)
    elif len(sys.argv) > 2 and sys.argv[2] in ("mpdata", "mpdata_mask"):
        main_mpdata(sys.argv[1], mask="island" if sys.argv[2] == "mpdata_mask" else None)
    elif len(sys.argv) > 2 and sys.argv[2] in ("mpdata_closed", "mpdata_open", "mpdata_mask_open"):
        main_mpdata(sys.argv[1], mask="island" if "mask" in sys.argv[2] else None, basin=sys.argv[2].split("_")[-1])
    elif len(sys.argv) > 2 and sys.argv[2] == "mask":
        main(sys.argv[1], mask="island")
    elif len(sys.argv) > 2 and sys.argv[2] in ("atm", "atm_pg31", "atm_pj"):
        main(sys.argv[1], pgf={"atm": 0, "atm_pg31": 1, "atm_pj": 3}[sys.argv[2]], atm=True)
    elif len(sys.argv) > 2 and sys.argv[2] in ("pg31", "wj", "pj"):
        main(sys.argv[1], pgf={"pg31": 1, "wj": 2, "pj": 3}[sys.argv[2]])
    elif len(sys.argv) > 2 and sys.argv[2] in ("ini", "ini_mask"):
        main_ini(sys.argv[1], mask="island" if sys.argv[2] == "ini_mask" else None)
    elif len(sys.argv) > 2 and sys.argv[2] in ("bc4", "bc4_mask"):
        main_bc4(sys.argv[1], mask="island" if sys.argv[2] == "bc4_mask" else None, rad2d=RAD2D, pc=PC)
    elif len(sys.argv) > 2 and sys.argv[2] in ("bc", "bc_mask"):
        main_bc(sys.argv[1], mask="island" if sys.argv[2] == "bc_mask" else None, rad2d=RAD2D, pc=PC)
    else:
        main(sys.argv[1])
