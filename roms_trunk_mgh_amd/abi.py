"""ctypes mirror of include/roms_hip.h (the C ABI) and of include/roms_fields.def.

The field table is parsed from the .def file so that the C side and the Python
side cannot drift apart; the struct layouts are checked at load time against
``roms_abi_sizeof`` exported by both shared libraries.
"""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE_DIR = os.path.join(ROOT, "include")

ROMS_MAXN = 64
ROMS_MAXNT = 16
ROMS_MAXFAST = 256

KINDS = ["K_2D", "K_2D_T2", "K_2D_T3", "K_2D_NT", "K_3DR", "K_3DW",
         "K_3DR_T2", "K_3DW_T2", "K_3DW_NAT", "K_4DT", "K_3DR_NT", "K_3DW_T3"]

# enum roms_adv (T_ADV logical records, ROMS/Modules/mod_param.F:382-394)
ADV = {"C2": 0, "C4": 1, "A4": 2, "U3": 3, "SU3": 4, "SPLINES": 5,
       "MPDATA": 6, "HSIMT": 7}
GLS_STAB = {"GALPERIN": 0, "KANTHA_CLAYSON": 1, "CANUTO_A": 2, "CANUTO_B": 3}            # enum roms_gls_stab
PGF = {"DJ_GRADPS": 0, "STANDARD": 1, "WJ_GRADP": 2, "PJ_GRADP": 3}         # enum roms_pgf (prsgrd.F:16-26)
LBC_PERIODIC, LBC_CLOSED, LBC_GRADIENT, LBC_CLAMPED, LBC_CHAPMAN_IMPLICIT, LBC_FLATHER, LBC_RADIATION = range(7)
LBC = {"Per": 0, "Clo": 1, "Gra": 2, "Cla": 3, "Cha": 4, "Fla": 5, "Rad": 6, "RadNud": 7, "Che": 8, "Shc": 9, "Red": 10}      # the keywords of roms_*.in
LBV = {"zeta": 0, "ubar": 1, "vbar": 2, "u": 3, "v": 4, "t": 5}
LBS = {"west": 0, "east": 1, "south": 2, "north": 3}


def _parse_fields():
    out = []
    pat = re.compile(r"^ROMS_FIELD\(\s*(\w+)\s*,\s*(\w+)\s*,\s*(\w+)\s*\)")
    with open(os.path.join(INCLUDE_DIR, "roms_fields.def")) as fh:
        for line in fh:
            m = pat.match(line.strip())
            if m:
                out.append((m.group(1), m.group(2), m.group(3)))
    return out


FIELDS = _parse_fields()                      # [(name, kind, owner)]
FIELD_ID = {n: i for i, (n, _, _) in enumerate(FIELDS)}
FIELD_KIND = {n: k for n, k, _ in FIELDS}


def trailing_shape(kind, N, NT, NAT):
    """Trailing dimensions (after LBi:UBi,LBj:UBj) of a field kind."""
    return {
        "K_2D": (), "K_2D_T2": (2,), "K_2D_T3": (3,), "K_2D_NT": (NT,),
        "K_3DR": (N,), "K_3DW": (N + 1,), "K_3DR_T2": (N, 2),
        "K_3DW_T2": (N + 1, 2), "K_3DW_NAT": (N + 1, NAT), "K_4DT": (N, 3, NT), "K_3DR_NT": (N, NT),
        "K_3DW_T3": (N + 1, 3),
    }[kind]


class Bounds(C.Structure):
    """roms_bounds_t -- ROMS/Include/set_bounds.h:20-79, tile.h:21-45."""
    _names = (
        "Lm Mm N NT NAT ntileI ntileJ tile Itile Jtile "
        "NghostPoints EWperiodic NSperiodic "
        "west_edge east_edge south_edge north_edge "
        "LBi UBi LBj UBj Istr Iend Jstr Jend "
        "IstrB IendB IstrM IstrP IendP IstrR IendR IstrT IendT IstrU "
        "JstrB JendB JstrM JstrP JendP JstrR JendR JstrT JendT JstrV "
        "Istrm3 Istrm2 Istrm1 IstrUm2 IstrUm1 Iendp1 Iendp2 Iendp2i Iendp3 "
        "Jstrm3 Jstrm2 Jstrm1 JstrVm2 JstrVm1 Jendp1 Jendp2 Jendp2i Jendp3"
    ).split()
    _fields_ = [(n, C.c_int) for n in _names]

    def as_dict(self):
        return {n: getattr(self, n) for n in self._names}


class Params(C.Structure):
    """roms_params_t -- scalars of mod_scalars.F / mod_param.F used on the path."""
    _fields_ = [
        ("dt", C.c_double), ("dtfast", C.c_double),
        ("g", C.c_double), ("rho0", C.c_double),
        ("gamma2", C.c_double), ("lambda_", C.c_double),
        ("ndtfast", C.c_int), ("nfast", C.c_int),
        ("weight1", C.c_double * ROMS_MAXFAST),
        ("weight2", C.c_double * ROMS_MAXFAST),
        ("Vtransform", C.c_int), ("limit_bstress", C.c_int),
        ("hc", C.c_double),
        ("sc_r", C.c_double * (ROMS_MAXN + 1)), ("Cs_r", C.c_double * (ROMS_MAXN + 1)),
        ("sc_w", C.c_double * (ROMS_MAXN + 1)), ("Cs_w", C.c_double * (ROMS_MAXN + 1)),
        ("Hadv", C.c_int * ROMS_MAXNT), ("Vadv", C.c_int * ROMS_MAXNT),
        ("lbc_west", C.c_int), ("lbc_east", C.c_int),
        ("lbc_south", C.c_int), ("lbc_north", C.c_int),
        ("nonlin_eos", C.c_int), ("eminusp", C.c_int),
        ("R0", C.c_double), ("T0", C.c_double), ("S0", C.c_double),
        ("Tcoef", C.c_double), ("Scoef", C.c_double),
        ("uv_adv", C.c_int), ("uv_cor", C.c_int), ("uv_vis2", C.c_int),
        ("curvgrid", C.c_int), ("var_rho_2d", C.c_int),
        ("ts_dif2", C.c_int), ("mix_geo_ts", C.c_int), ("mix_s_ts", C.c_int),
        ("salinity", C.c_int), ("lmd_nonlocal", C.c_int), ("solar_source", C.c_int),
        ("splines_vdiff", C.c_int), ("splines_vvisc", C.c_int),
        ("Akt_bak", C.c_double * ROMS_MAXNT), ("Akv_bak", C.c_double),
        ("swfrac_mu1", C.c_double), ("swfrac_mu2", C.c_double), ("swfrac_r1", C.c_double),
        ("uv_drag", C.c_int), ("mpdata_fast", C.c_int),
        ("blk_ZQ", C.c_double), ("blk_ZT", C.c_double), ("blk_ZW", C.c_double),
        ("masking", C.c_int), ("pgf", C.c_int),
        ("lbc", (C.c_int * 6) * 4),
        ("obc_out", (C.c_double * 6) * 4), ("obc_in", (C.c_double * 6) * 4),
        ("ts_dif4", C.c_int), ("uv_vis4", C.c_int), ("mix_iso_ts", C.c_int), ("radiation_2d", C.c_int),
        ("Cdb_min", C.c_double), ("Cdb_max", C.c_double),
        ("gls_mixing", C.c_int), ("gls_stability", C.c_int), ("gls_n2s2_horavg", C.c_int), ("gls_ri_splines", C.c_int),
        ("gls_p", C.c_double), ("gls_m", C.c_double), ("gls_n", C.c_double), ("gls_cmu0", C.c_double),
        ("gls_c1", C.c_double), ("gls_c2", C.c_double), ("gls_c3m", C.c_double), ("gls_c3p", C.c_double),
        ("gls_sigk", C.c_double), ("gls_sigp", C.c_double), ("gls_Kmin", C.c_double), ("gls_Pmin", C.c_double),
        ("Akk_bak", C.c_double), ("Akp_bak", C.c_double), ("Zos", C.c_double),
        ("wet_dry", C.c_int), ("point_sources", C.c_int), ("Dcrit", C.c_double),
        ("atm_press", C.c_int), ("press_compensate", C.c_int), ("ts_mix_stability", C.c_int), ("ts_mix_min_strat", C.c_int),
    ]


class StepIdx(C.Structure):
    """roms_step_idx_t -- mod_stepping.F indices, main3d.F:189-191,597-662."""
    _fields_ = [(n, C.c_int) for n in
                "iic ntfirst nstp nnew nrhs kstp krhs knew iif predictor_2d_step".split()]


class Fields(C.Structure):
    """roms_fields_t -- one double* per module array."""
    _fields_ = [(n, C.POINTER(C.c_double)) for n, _, _ in FIELDS]


def check_abi(lib):
    """Compare struct sizes with the C side (0=bounds 1=params 2=step 3=fields)."""
    lib.roms_abi_sizeof.restype = C.c_int
    lib.roms_abi_sizeof.argtypes = [C.c_int]
    want = [C.sizeof(Bounds), C.sizeof(Params), C.sizeof(StepIdx), C.sizeof(Fields)]
    got = [lib.roms_abi_sizeof(i) for i in range(4)]
    if want != got:
        raise RuntimeError(f"ABI struct size mismatch python={want} C={got}")
    lib.roms_abi_sizeof.argtypes = [C.c_int]
    if lib.roms_abi_sizeof(4) != len(FIELDS):
        raise RuntimeError("ABI field-count mismatch")
