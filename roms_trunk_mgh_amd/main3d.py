"""Host-side mirror of the reference's step sequencer `main3d`
(ROMS/Nonlinear/main3d.F:3-927) restricted to the hot path of SURVEY.md section 8a.

In a ROMS build the unchanged Fortran main3d drives libroms_hip.so through the
ISO_C_BINDING shims (fortran/roms_hip_mod.F90).  This Python driver exists
because the full Fortran executable cannot be linked in this environment
(netCDF-Fortran is absent); it issues exactly the same calls in the same order,
against either backend (`hip.RomsHip` = the product, or the CPU oracle in tests):

    main3d.F:189-191  nstp/nnew/nrhs rotation
    main3d.F:269-283  first step only: ini_zeta, set_depth, ini_fields
    main3d.F:307-314  set_massflux, rho_eos, diag      (diag: diagnostics=True, every ninfo steps)
    main3d.F:280      set_data -> ana_srflux           (physics=True, BENCHMARK: shortwave flux of the hour)
    main3d.F:388-394  bulk_flux, set_vbc               (physics=True; else fixed forcing inputs)
    main3d.F:467-475  lmd_vmix (physics=True; else fixed mixing inputs); omega; wvelocity (diagnostics=True)
    main3d.F:489      set_zeta
    main3d.F:563      rhs3d
    main3d.F:567      gls_prestep                      (GLS_MIXING applications)
    main3d.F:592-700  LOOP_2D (predictor/corrector step2d)
    main3d.F:736      set_depth
    main3d.F:762      step3d_uv
    main3d.F:789      omega
    main3d.F:793      gls_corstep                      (GLS_MIXING applications)
    main3d.F:814      step3d_t
    main3d.F:914      iic += 1
"""
from . import abi


def host_clock(tdays):
    """What CALL caldate (tdays(ng), yd_dp=yday, h_dp=hour) returns with TIME_REF = 0 (roms_benchmark*.in:424;
    ROMS/Utility/dateclock.F:73-236: date number of 0001-01-01 = 367, day fraction of the sum, seconds rounded
    to the nearest one) during the first year -- enough for the runs of this repository; pinned against the
    reference's caldate in tests/test_ref_pinning.py."""
    import math
    dn = 367.0 + tdays
    frac = abs(dn - math.trunc(dn))
    day = math.trunc(dn) - 367
    if not 0 <= day < 365:
        raise ValueError("host_clock covers the first 365 days only")
    return float(day + 1) + frac, round(frac * 86400.0) / 3600.0


class Main3D:
    def __init__(self, backend, ntstart=1, physics=False, diagnostics=False, ninfo=1):
        """physics=True also runs the per-step physics that is on the device:
        bulk_flux and lmd_vmix (BULK_FLUXES / LMD_MIXING applications, i.e. BENCHMARK) and set_vbc, in
        the reference's order;
        with physics=False their outputs stay the fixed fields ana.py filled in.
        diagnostics=True adds the two diagnostics the reference's step carries: wvelocity after the
        first omega (main3d.F:475) and, every `ninfo` steps, the tile-local part of diag after rho_eos
        (main3d.F:314; it therefore sees the wvel of the previous step, as in the reference); the
        12-vector of the last call is kept in `last_diag` (layout: roms_hip.h, roms_hip_diag)."""
        self.be = backend
        self.physics = physics
        self.diagnostics = diagnostics
        self.ninfo = ninfo
        self.last_diag = None
        self.iic = ntstart
        self.ntstart = ntstart
        self.ntfirst = ntstart
        self.indx1 = 1                      # mod_stepping.F initial value
        self.s = abi.StepIdx(iic=ntstart, ntfirst=ntstart, nstp=1, nnew=2, nrhs=1,
                             kstp=1, krhs=1, knew=1, iif=1, predictor_2d_step=0)

    def initial(self):
        """The hot-path part of ROMS/Nonlinear/initial.F:337-571:
        set_depth, set_massflux, omega, rho_eos on the initial state."""
        s = self.s
        s.nstp, s.nnew, s.nrhs = 1, 2, 1
        if self.be.st.p.wet_dry:              # initial.F:438-466: the wet/dry masks of the initial state (Tindex = 1)
            s.kstp = 1
            self.be.call("wetdry", s)
        for k in ("set_depth", "set_massflux", "omega", "rho_eos"):
            self.be.call(k, s)

    def _rotate(self):
        s = self.s
        s.iic = self.iic
        s.ntfirst = self.ntfirst
        s.nstp = 1 + (self.iic - self.ntstart) % 2
        s.nnew = 3 - s.nstp
        s.nrhs = s.nstp

    def step(self):
        be, s = self.be, self
        self._rotate()
        s = self.s
        if self.iic == self.ntstart:          # main3d.F:269-283: all time levels and the other initial fields
            be.call("ini_zeta", s)
            be.call("set_depth", s)
            be.call("ini_fields", s)
        be.call("set_massflux", s)
        be.call("rho_eos", s)
        if self.diagnostics and (self.iic - 1) % self.ninfo == 0:
            self.last_diag = be.diag(s)
        if self.physics:
            bench_app = getattr(be.st, "cfg", {}).get("app") == "BENCHMARK"
            if bench_app:                     # set_data: ANA_SRFLUX with ALBEDO is the time-dependent forcing
                dt = be.st.p.dt
                be.ana_srflux(*host_clock((self.iic - self.ntstart) * dt / 86400.0))
            if bench_app:                     # BULK_FLUXES
                be.call("bulk_flux", s)
            be.call("set_vbc", s)
            if bench_app:                     # LMD_MIXING
                be.call("lmd_vmix", s)
        be.call("omega", s)
        if self.diagnostics:
            be.call("wvelocity", s)
        be.call("set_zeta", s)
        be.call("rhs3d", s)
        gls = bool(be.st.p.gls_mixing)
        if gls:                               # main3d.F:564-567
            be.call("gls_prestep", s)
        self.indx1 = be.step2d_loop(s, self.indx1)
        be.call("set_depth", s)
        be.call("step3d_uv", s)
        be.call("omega", s)
        if gls:                               # main3d.F:790-793
            be.call("gls_corstep", s)
        be.call("step3d_t", s)
        self.iic += 1

    def run(self, nsteps):
        for _ in range(nsteps):
            self.step()


def reduce_diag(vectors):
    """Combine the tile-local 12-vectors of `diag` the way diag.F:398-420 does across ranks: SUM of
    volume, avgke, avgpe; MAX of maxspeed, maxrho; MAXLOC of the Courant number (the tile holding the
    largest max_C supplies Cu, Cv, Cw and the location)."""
    import numpy as np
    v = np.asarray(vectors, dtype=float).reshape(-1, 12)
    out = np.zeros(12)
    out[0:3] = v[:, 0:3].sum(axis=0)
    out[3:5] = v[:, 3:5].max(axis=0)
    out[5:12] = v[int(np.argmax(v[:, 5])), 5:12]
    return out
