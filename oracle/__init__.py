"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

ctypes front end of oracle/_build/liboracle.so, the plain-C restatement of the
reference's nonlinear 3-D kernels (see oracle/oracle.h for the pinning status).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (roms_trunk_mgh_amd) never does.
"""
import ctypes as C
import os
import subprocess

from roms_trunk_mgh_amd import abi

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KERNELS = ["set_massflux", "omega", "set_zeta", "set_depth", "rho_eos", "pre_step3d",
           "prsgrd", "t3dmix2", "rhs3d_tile", "uv3dmix2", "rhs3d", "step2d",
           "step3d_uv", "step3d_t", "bulk_flux", "set_vbc", "lmd_vmix", "wvelocity", "ini_zeta", "ini_fields",
           "t3dmix4", "uv3dmix4", "gls_prestep", "gls_corstep", "wetdry"]


def build(force=False):
    so = os.path.join(_DIR, "_build", "liboracle.so")
    if force or not os.path.exists(so) or os.path.exists("/usr/bin/make"):
        if os.path.exists("/usr/bin/make") or force:
            subprocess.run(["make", "-s", "-C", _DIR], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_DIR, "_build", "liboracle.so")
        if not os.path.exists(so):
            so = build()
        # ROMS_ORACLE_VARIANT=O3: the -O3 -march=native build, used ONLY by bench.py's cpu_baseline leg
        # (timing); every parity check runs the -O2 -ffp-contract=off build above
        if os.environ.get("ROMS_ORACLE_VARIANT") == "O3":
            so = os.path.join(_DIR, "_build", "liboracle_O3.so")
        _LIB = C.CDLL(so)
        abi.check_abi(_LIB)
        for k in KERNELS:
            fn = getattr(_LIB, "oracle_" + k, None)
            if fn is None:
                continue
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(abi.Bounds), C.POINTER(abi.Params),
                           C.POINTER(abi.StepIdx), C.POINTER(abi.Fields)]
        _ip, _dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        _LIB.oracle_set_sources.restype = C.c_int
        _LIB.oracle_set_sources.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int]
        if hasattr(_LIB, "oracle_step2d_loop"):
            _LIB.oracle_step2d_loop.restype = C.c_int
            _LIB.oracle_step2d_loop.argtypes = [C.POINTER(abi.Bounds), C.POINTER(abi.Params),
                                                C.POINTER(abi.StepIdx), C.POINTER(abi.Fields),
                                                C.POINTER(C.c_int)]
    return _LIB


class Oracle:
    """Backend with the same method names as roms_trunk_mgh_amd.hip.RomsHip,
    operating in place on a TileState's host arrays."""

    name = "oracle"

    def __init__(self, state):
        self.st = state
        self.l = lib()
        self.F = state.fields_struct()
        if getattr(state, "sources", None) is not None:
            self.set_sources(state.sources)
        elif hasattr(self.l, "oracle_set_sources"):
            self.l.oracle_set_sources(0, None, None, None, None, None, None, None, 0, 0)     # the table is process-wide

    def set_sources(self, src):
        rc = self.l.oracle_set_sources(*src.c_args(), C.c_int(self.st.b.N), C.c_int(self.st.b.NT))
        if rc != 0:
            raise RuntimeError(f"oracle_set_sources returned {rc}")

    def call(self, kernel, s):
        fn = getattr(self.l, "oracle_" + kernel)
        rc = fn(C.byref(self.st.b), C.byref(self.st.p), C.byref(s), C.byref(self.F))
        if rc != 0:
            raise RuntimeError(f"oracle_{kernel} returned {rc}")

    def ana_srflux(self, yday, hour):
        self.l.oracle_ana_srflux.restype = C.c_int
        self.l.oracle_ana_srflux.argtypes = [C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.StepIdx),
                                             C.POINTER(abi.Fields), C.c_double, C.c_double]
        s0 = abi.StepIdx()
        rc = self.l.oracle_ana_srflux(C.byref(self.st.b), C.byref(self.st.p), C.byref(s0), C.byref(self.F),
                                      float(yday), float(hour))
        if rc != 0:
            raise RuntimeError(f"oracle_ana_srflux returned {rc}")

    def diag(self, s):
        import numpy as np
        out = np.zeros(12)
        dp = C.POINTER(C.c_double)
        self.l.oracle_diag.restype = C.c_int
        self.l.oracle_diag.argtypes = [C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.StepIdx),
                                       C.POINTER(abi.Fields), dp]
        rc = self.l.oracle_diag(C.byref(self.st.b), C.byref(self.st.p), C.byref(s), C.byref(self.F),
                                out.ctypes.data_as(dp))
        if rc != 0:
            raise RuntimeError(f"oracle_diag returned {rc}")
        return out

    def bc(self, kind, s, nout, itrc=1):
        """One lateral boundary-condition routine (zetabc, u2dbc, v2dbc, u3dbc, v3dbc, t3dbc) on its own."""
        fn = getattr(self.l, "oracle_bc")
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(abi.Bounds), C.POINTER(abi.Params), C.POINTER(abi.StepIdx), C.POINTER(abi.Fields),
                       C.c_int, C.c_int, C.c_int]
        kid = {"zetabc": 1, "u2dbc": 2, "v2dbc": 3, "u3dbc": 4, "v3dbc": 5, "t3dbc": 6}[kind]
        rc = fn(C.byref(self.st.b), C.byref(self.st.p), C.byref(s), C.byref(self.F), kid, int(nout), int(itrc))
        if rc != 0:
            raise RuntimeError(f"oracle_bc {kind} returned {rc}")

    def step2d_loop(self, s, indx1):
        ii = C.c_int(indx1)
        rc = self.l.oracle_step2d_loop(C.byref(self.st.b), C.byref(self.st.p), C.byref(s),
                                       C.byref(self.F), C.byref(ii))
        if rc != 0:
            raise RuntimeError(f"oracle_step2d_loop returned {rc}")
        return ii.value

    def to_host(self):
        return self.st

    def sync(self):
        pass
