"""Front end of oracle/_ref/<APP>/libref.so -- the subset of the reference's own
Fortran that compiles here (oracle/build_ref.sh) behind oracle/ref_wrap.F90.
TEST INFRASTRUCTURE ONLY.  One process can hold one configuration (the
reference keeps its state in module variables), so tests run it in a child."""
import ctypes as C
import os

from roms_trunk_mgh_amd import abi

_DIR = os.path.dirname(os.path.abspath(__file__))
KERNEL_ID = {"set_depth": 1, "set_massflux": 2, "set_zeta": 3, "rho_eos": 4, "prsgrd": 5,
             "t3dmix2": 6, "uv3dmix2": 7, "t3dmix4": 8, "uv3dmix4": 9}


def lib_path(app):
    return os.path.join(_DIR, "_ref", app, "libref.so")


def available(app):
    return os.path.exists(lib_path(app))


class Ref:
    def __init__(self, state):
        app = state.cfg["app"] + ("_MASK" if state.p.masking else "")      # <APP>_MASK: built with -DMASKING
        if state.p.wet_dry:
            app += "_WET"                                                  # <APP>_MASK_WET...: built with -DWET_DRY as well
        if state.p.atm_press:
            app += "_ATM"                                                  # built with -DATM_PRESS as well
            if state.p.press_compensate:
                app += "_PC"                                               # ... and -DPRESS_COMPENSATE
        app += {0: "", 1: "_PG31", 2: "_WJ", 3: "_PJ"}[int(state.p.pgf)]   # prsgrd31.h builds (plain / WJ_GRADP), prsgrd40.h
        if state.p.uv_drag == 3:
            app += "_LOGDRAG"                                              # UV_LOGDRAG instead of the application's law
        if state.p.ts_mix_min_strat:
            app += "_MINSTRAT"                                             # built with -DTS_MIX_MIN_STRAT as well
        if state.p.ts_mix_stability:
            app += "_STAB"                                                 # built with -DTS_MIX_STABILITY as well
        if state.p.mix_iso_ts:
            app += "_ISO"                                                  # ... and MIX_ISO_TS as the tracer mixing choice
        elif state.p.ts_dif4 or state.p.uv_vis4:
            app += "_DIF4"                                                 # built with TS_DIF4 and UV_VIS4 added
        if state.p.eminusp:
            app += "_EMP"                                                  # built with -DEMINUSP
        if state.p.limit_bstress:
            app += "_LIMBS"                                                # built with -DLIMIT_BSTRESS
        if state.p.radiation_2d:
            app += "_RAD2D"                                                # built with -DRADIATION_2D
        if state.p.uv_vis2 == 2:
            app += "_GEOUV"                                                # MIX_GEO_UV instead of MIX_S_UV (uv3dmix2_geo.h)
        if state.p.gls_mixing == 2:
            app += "_MY25"                                                 # MY25_MIXING builds (ref_headers/*_my25.h)
        elif state.p.gls_mixing:
            app += "_GLS"                                                  # GLS_MIXING builds (ref_headers/*_gls.h)
        self.l = C.CDLL(lib_path(app))
        self.st = state
        self.l.ref_abi_sizeof.argtypes = [C.c_int]
        want = [C.sizeof(abi.Bounds), C.sizeof(abi.Params), C.sizeof(abi.StepIdx), C.sizeof(abi.Fields)]
        got = [self.l.ref_abi_sizeof(i) for i in range(4)]
        if want != got:
            raise RuntimeError(f"ref ABI mismatch ({app}) python={want} fortran={got}")
        self.l.ref_setup.argtypes = [C.POINTER(abi.Bounds), C.POINTER(abi.Params)]
        rc = self.l.ref_setup(C.byref(state.b), C.byref(state.p))
        if rc != 0:
            raise RuntimeError("ref_setup failed")
        self.l.ref_call.argtypes = [C.c_int, C.POINTER(abi.Bounds), C.POINTER(abi.Params),
                                    C.POINTER(abi.StepIdx), C.POINTER(abi.Fields)]
        self.F = state.fields_struct()

    def bounds(self):
        out = (C.c_int * 50)()
        self.l.ref_get_bounds(out)
        names = ("LBi UBi LBj UBj Istr Iend Jstr Jend "
                 "IstrB IendB IstrM IstrP IendP IstrR IendR IstrT IendT IstrU "
                 "JstrB JendB JstrM JstrP JendP JstrR JendR JstrT JendT JstrV "
                 "Istrm3 Istrm2 Istrm1 IstrUm2 IstrUm1 Iendp1 Iendp2 Iendp2i Iendp3 "
                 "Jstrm3 Jstrm2 Jstrm1 JstrVm2 JstrVm1 Jendp1 Jendp2 Jendp2i Jendp3 "
                 "west_edge east_edge south_edge north_edge").split()
        return dict(zip(names, list(out)))

    def set_weights(self, ndtfast):
        import numpy as np
        w1 = np.zeros(2 * ndtfast)
        w2 = np.zeros(2 * ndtfast)
        nf = C.c_int(0)
        self.l.ref_set_weights.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        self.l.ref_set_weights(ndtfast, C.byref(nf), w1.ctypes.data, w2.ctypes.data)
        return nf.value, w1, w2

    def call(self, kernel, s):
        rc = self.l.ref_call(KERNEL_ID[kernel], C.byref(self.st.b), C.byref(self.st.p), C.byref(s), C.byref(self.F))
        if rc != 0:
            raise RuntimeError(f"ref_call {kernel} rc={rc}")

    def bc(self, kind, s, nout, itrc=1):
        """zetabc / u2dbc / v2dbc / u3dbc / v3dbc / t3dbc _tile of the reference on the S/N edges, with LBC(...) from
        the state's lbc table and BOUNDARY(ng)%*_south/_north from its *_bry fields (ref_bc in ref_wrap.F90)."""
        kid = {"zetabc": 1, "u2dbc": 2, "v2dbc": 3, "u3dbc": 4, "v3dbc": 5, "t3dbc": 6,
               "ini_zeta": 7, "ini_fields": 8}[kind]          # 7, 8: ini_fields.F, the first-step initialisation
        self.l.ref_bc.argtypes = [C.c_int, C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.StepIdx),
                                  C.POINTER(abi.Fields), C.c_int, C.c_int]
        rc = self.l.ref_bc(kid, C.byref(self.st.b), C.byref(self.st.p), C.byref(s), C.byref(self.F), int(nout), int(itrc))
        if rc != 0:
            raise RuntimeError(f"ref_bc {kind} rc={rc}")

    def mpdata_adiff(self, oHz, t3, Ta, Ua, Va, Wa):
        """mpdata_adiff_tile on caller-held private arrays (Fortran-ordered float64):
        oHz, Ta, Ua, Va (nis,njs,N), Wa (nis,njs,N+1), t3 = t(:,:,:,3,itrc) (ni,nj,N)."""
        self.l.ref_mpdata_adiff.argtypes = [C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.Fields)] + \
                                           [C.c_void_p] * 6
        rc = self.l.ref_mpdata_adiff(C.byref(self.st.b), C.byref(self.st.p), C.byref(self.F),
                                     oHz.ctypes.data, t3.ctypes.data, Ta.ctypes.data, Ua.ctypes.data,
                                     Va.ctypes.data, Wa.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"ref_mpdata_adiff rc={rc}")

    def physics(self, kernel, s):
        """set_vbc / bulk_flux through the reference's own module procedures."""
        kid = {"set_vbc": 1, "bulk_flux": 2, "lmd_vmix": 3}[kernel]
        self.l.ref_physics.argtypes = [C.c_int, C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.StepIdx),
                                       C.POINTER(abi.Fields)]
        rc = self.l.ref_physics(kid, C.byref(self.st.b), C.byref(self.st.p), C.byref(s), C.byref(self.F))
        if rc != 0:
            raise RuntimeError(f"ref_physics {kernel} rc={rc}")

    def gls(self, kernel, s):
        """gls_prestep / gls_corstep through the reference's own module procedures (GLS builds only: ref_gls)."""
        kid = {"gls_prestep": 1, "gls_corstep": 2}[kernel]
        self.l.ref_gls.argtypes = [C.c_int, C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.StepIdx),
                                   C.POINTER(abi.Fields)]
        rc = self.l.ref_gls(kid, C.byref(self.st.b), C.byref(self.st.p), C.byref(s), C.byref(self.F))
        if rc != 0:
            raise RuntimeError(f"ref_gls {kernel} rc={rc}")

    def diagnostics(self, kernel, s, workdir="."):
        """wvelocity (writes wvel) or diag through the reference's own module procedures.  diag keeps no
        result (it prints and resets, diag.F:449-540); its report is captured from the file the wrapper
        points `stdout` at and returned parsed: dict(avgke, avgpe, avgkp, volume, Ci, Cj, Ck, Cu, Cv, Cw,
        maxspeed) at the printed precision (1pe14.6 / 1pe13.6)."""
        import re
        kid = {"wvelocity": 1, "diag": 2}[kernel]
        self.l.ref_diagnostics.argtypes = [C.c_int, C.POINTER(abi.Bounds), C.POINTER(abi.Params),
                                           C.POINTER(abi.StepIdx), C.POINTER(abi.Fields)]
        cwd = os.getcwd()
        os.chdir(workdir)
        try:
            rc = self.l.ref_diagnostics(kid, C.byref(self.st.b), C.byref(self.st.p), C.byref(s), C.byref(self.F))
            if rc != 0:
                raise RuntimeError(f"ref_diagnostics {kernel} rc={rc}")
            if kernel != "diag":
                return None
            text = open("ref_diag_stdout.txt").read()
            os.remove("ref_diag_stdout.txt")
        finally:
            os.chdir(cwd)
        num = r"[-+]?\d\.\d+E[-+]\d+"
        m1 = re.search(r"^\s*(\d+) 0001-01-01 00:00:00\.00\s*(%s)\s*(%s)\s*(%s)\s*(%s)" % (num, num, num, num), text, re.M)
        m2 = re.search(r"\((\d+),(\d+),(\d+)\)\s*(%s)\s*(%s)\s*(%s)\s*(%s)" % (num, num, num, num), text)
        if not (m1 and m2):
            raise RuntimeError("could not parse the reference's diag report:\n" + text)
        return dict(istep=int(m1.group(1)), avgke=float(m1.group(2)), avgpe=float(m1.group(3)), avgkp=float(m1.group(4)),
                    volume=float(m1.group(5)), Ci=int(m2.group(1)), Cj=int(m2.group(2)), Ck=int(m2.group(3)),
                    Cu=float(m2.group(4)), Cv=float(m2.group(5)), Cw=float(m2.group(6)), maxspeed=float(m2.group(7)),
                    text=text)

    def ana(self, kernel, cfg5):
        """The reference's own analytic set-up routines (ref_ana in ref_wrap.F90): "grid" (ana_grid + metrics),
        "scoord" (set_scoord; returns sc_r, Cs_r, sc_w, Cs_w, hc), "initial" (ana_initial), "forcing" (the
        ana_* forcing routines of the application).  cfg5 = theta_s, theta_b, Tcline, Vstretching, tdays."""
        import numpy as np
        kid = {"grid": 1, "scoord": 2, "initial": 3, "forcing": 4, "srflux": 5}[kernel]
        dp = C.POINTER(C.c_double)
        self.l.ref_ana.argtypes = [C.c_int, C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.Fields), dp, dp]
        cfg = np.array(cfg5, dtype=np.float64)
        N = self.st.b.N
        out = np.zeros(4 * (N + 1) + 1)
        rc = self.l.ref_ana(kid, C.byref(self.st.b), C.byref(self.st.p), C.byref(self.F), cfg.ctypes.data_as(dp),
                            out.ctypes.data_as(dp))
        if rc != 0:
            raise RuntimeError(f"ref_ana {kernel} rc={rc}")
        if kernel == "scoord":
            n1 = N + 1
            return dict(sc_r=out[0:n1], Cs_r=out[n1:2 * n1], sc_w=out[2 * n1:3 * n1], Cs_w=out[3 * n1:4 * n1], hc=out[4 * n1])
        if kernel == "srflux":
            return dict(yday=float(out[0]), hour=float(out[1]))
        return None
