"""ctypes binding of libroms_hip.so -- the Python stand-in for the Fortran
ISO_C_BINDING shims (fortran/roms_hip_mod.F90).  Method names mirror the
reference's module procedures (`step3d_t(ng,tile)` -> `call("step3d_t", s)`).

There is NO CPU fallback: if the shared library is missing, or no HIP device is
present, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libroms_hip.so")
_LIB = None


def use_library(path):
    """Developer A/B tools only (tools/bench_kernel.py --lib): load another build of the same ABI instead of the
    in-tree product library.  Must be called before the first load(); nothing in tests/, bench.py or
    __graft_entry__.py calls it, and no environment variable can redirect the product path."""
    global LIB_PATH
    if _LIB is not None:
        raise RuntimeError("use_library() after the library was loaded")
    LIB_PATH = os.path.abspath(path)

ENTRIES = ["set_massflux", "rho_eos", "omega", "set_zeta", "set_depth", "rhs3d",
           "pre_step3d", "prsgrd", "t3dmix2", "rhs3d_tile", "uv3dmix2", "step2d",
           "step3d_uv", "step3d_t", "bulk_flux", "set_vbc", "lmd_vmix", "wvelocity", "ini_zeta", "ini_fields",
           "t3dmix4", "uv3dmix4", "gls_prestep", "gls_corstep", "wetdry"]

# every symbol include/roms_hip.h declares
DECLARED_SYMBOLS = (
    ["roms_hip_init", "roms_hip_finalize", "roms_hip_get_unique_id", "roms_hip_set_bounds",
     "roms_hip_set_params", "roms_hip_register_field", "roms_hip_sync_to_device",
     "roms_hip_sync_to_host", "roms_hip_sync_all_to_device", "roms_hip_sync_all_to_host",
     "roms_hip_device_ptr", "roms_hip_device_synchronize", "roms_hip_last_error",
     "roms_hip_step2d_loop", "roms_hip_exchange", "roms_hip_timing_enable",
     "roms_hip_timing_last_ms", "roms_hip_calib_stream", "roms_hip_set_halo_relay", "roms_hip_diag",
     "roms_hip_snapshot_begin", "roms_hip_snapshot_end", "roms_hip_halo_plan",
     "roms_hip_ana_srflux", "roms_hip_check_guards", "roms_hip_row_metrics_state",
     "roms_hip_graph_exchanges", "roms_hip_graph_exchanges_state", "roms_hip_set_sources"] + ["roms_hip_" + e for e in ENTRIES])


_DP = C.POINTER(C.c_double)
class HaloMsg(C.Structure):
    """roms_halo_msg_t of include/roms_hip.h"""
    _fields_ = [("peer", C.c_int), ("tag", C.c_int), ("count", C.c_long), ("buf", _DP)]


RELAY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(HaloMsg), C.c_int, C.POINTER(HaloMsg))


def load():
    """dlopen libroms_hip.so (does not touch the GPU)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m roms_trunk_mgh_amd._build` "
            "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    abi.check_abi(lib)
    lib.roms_hip_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.roms_hip_set_bounds.argtypes = [C.POINTER(abi.Bounds)]
    lib.roms_hip_set_params.argtypes = [C.POINTER(abi.Params)]
    lib.roms_hip_register_field.argtypes = [C.c_int, C.c_void_p, C.c_long]
    lib.roms_hip_last_error.restype = C.c_char_p
    lib.roms_hip_device_ptr.restype = C.c_void_p
    lib.roms_hip_device_ptr.argtypes = [C.c_int]
    lib.roms_hip_timing_last_ms.restype = C.c_double
    lib.roms_hip_timing_last_ms.argtypes = [C.c_char_p]
    lib.roms_hip_get_unique_id.argtypes = [C.c_void_p]
    for e in ENTRIES:
        fn = getattr(lib, "roms_hip_" + e, None)
        if fn is not None:
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(abi.StepIdx)]
    if hasattr(lib, "roms_hip_step2d_loop"):
        lib.roms_hip_step2d_loop.argtypes = [C.POINTER(abi.StepIdx), C.POINTER(C.c_int)]
    lib.roms_hip_calib_stream.argtypes = [C.c_long]
    _ip = C.POINTER(C.c_int)
    lib.roms_hip_set_sources.argtypes = [C.c_int, _ip, _ip, _DP, _DP, _DP, _DP, _ip]
    lib.roms_hip_set_halo_relay.argtypes = [RELAY_FN, C.c_void_p]
    if hasattr(lib, "roms_hip_tile_neighbors"):
        lib.roms_hip_tile_neighbors.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)]
    _LIB = lib
    return lib


def tile_neighbors(rank, ntileI, ntileJ, Nghost, NghostPoints, EWperiodic, NSperiodic):
    """Host-only helper of the library (mp_exchange.F:73-286); needs no GPU."""
    out = (C.c_int * 12)()
    load().roms_hip_tile_neighbors(rank, ntileI, ntileJ, Nghost, NghostPoints,
                                   int(EWperiodic), int(NSperiodic), out)
    keys = ["Wtile", "Etile", "Stile", "Ntile", "GsendW", "GsendE", "GrecvW", "GrecvE",
            "GsendS", "GsendN", "GrecvS", "GrecvN"]
    return dict(zip(keys, list(out)))


def halo_plan(bounds, rank):
    """Host-only helper of the library: the (sends, recvs) of one halo update of tile `rank`, each a list
    of dicts peer, tag, i0, wi, j0, wj (include/roms_hip.h, roms_hip_halo_plan); needs no GPU."""
    out = (C.c_int * 98)()
    lib = load()
    lib.roms_hip_halo_plan.argtypes = [C.POINTER(abi.Bounds), C.c_int, C.POINTER(C.c_int)]
    lib.roms_hip_halo_plan(C.byref(bounds), rank, out)
    ns, nr = out[0], out[1]
    keys = ["peer", "tag", "i0", "wi", "j0", "wj"]
    msgs = [dict(zip(keys, out[2 + 6 * m: 8 + 6 * m])) for m in range(ns + nr)]
    return msgs[:ns], msgs[ns:]


class RomsHip:
    """One tile on one GPU.  Owns nothing on the host: the TileState arrays stay
    the caller's (as the Fortran module arrays do)."""

    name = "hip"
    _live = None

    def __init__(self, state, rank=0, device=0, nccl_unique_id=None, leave_unregistered=()):
        """leave_unregistered: names of fields an application without the option does not have (the land/sea masks
        without MASKING, visc4_p / visc4_r / diff4 without UV_VIS4 / TS_DIF4): the library keeps defaults for them."""
        self.st = state
        self.l = load()
        if RomsHip._live is not None:
            RomsHip._live.close()
        RomsHip._live = self
        b = state.b
        uid = None
        if nccl_unique_id is not None:
            self._uid = C.create_string_buffer(bytes(nccl_unique_id), 128)
            uid = C.cast(self._uid, C.c_void_p)
        self._chk(self.l.roms_hip_init(rank, b.ntileI, b.ntileJ, device, uid), "init")
        self._chk(self.l.roms_hip_set_bounds(C.byref(b)), "set_bounds")
        self._chk(self.l.roms_hip_set_params(C.byref(state.p)), "set_params")
        for name, _, _ in abi.FIELDS:
            if name in leave_unregistered:
                continue
            a = state.arr[name]
            self._chk(self.l.roms_hip_register_field(abi.FIELD_ID[name], a.ctypes.data, a.size),
                      "register_field " + name)
        self._chk(self.l.roms_hip_sync_all_to_device(), "sync_all_to_device")
        if getattr(state, "sources", None) is not None:
            self.set_sources(state.sources)

    def set_sources(self, src):
        """SOURCES(ng) with LuvSrc (roms_trunk_mgh_amd/sources.py) -> roms_hip_set_sources"""
        self._chk(self.l.roms_hip_set_sources(*src.c_args()), "set_sources")

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.l.roms_hip_last_error()
            raise RuntimeError(f"roms_hip {what} failed rc={rc}: {msg.decode() if msg else ''}")

    def call(self, kernel, s):
        self._chk(getattr(self.l, "roms_hip_" + kernel)(C.byref(s)), kernel)

    def snapshot_begin(self, names):
        """Start an asynchronous device-to-host snapshot of the named fields (roms_hip.h)."""
        ids = (C.c_int * len(names))(*[abi.FIELD_ID[n] for n in names])
        self.l.roms_hip_snapshot_begin.argtypes = [C.POINTER(C.c_int), C.c_int]
        self._chk(self.l.roms_hip_snapshot_begin(ids, len(names)), "snapshot_begin")

    def snapshot_end(self):
        self._chk(self.l.roms_hip_snapshot_end(), "snapshot_end")

    def ana_srflux(self, yday, hour):
        """ana_srflux (ALBEDO branch) for the day of the year and hour caldate gives; writes srflx."""
        self.l.roms_hip_ana_srflux.argtypes = [C.c_double, C.c_double]
        self._chk(self.l.roms_hip_ana_srflux(float(yday), float(hour)), "ana_srflux")

    def diag(self, s):
        """Tile-local sums and maxima of diag_tile (diag.F:190-290) as a 12-vector, see roms_hip.h."""
        import numpy as np
        out = np.zeros(12)
        self.l.roms_hip_diag.argtypes = [C.POINTER(abi.StepIdx), _DP]
        self._chk(self.l.roms_hip_diag(C.byref(s), out.ctypes.data_as(_DP)), "diag")
        return out

    def step2d_loop(self, s, indx1):
        ii = C.c_int(indx1)
        self._chk(self.l.roms_hip_step2d_loop(C.byref(s), C.byref(ii)), "step2d_loop")
        return ii.value

    def to_device(self, names=None):
        if names is None:
            self._chk(self.l.roms_hip_sync_all_to_device(), "sync_all_to_device")
        else:
            for n in names:
                self._chk(self.l.roms_hip_sync_to_device(abi.FIELD_ID[n]), "sync_to_device")

    def to_host(self, names=None):
        if names is None:
            self._chk(self.l.roms_hip_sync_all_to_host(), "sync_all_to_host")
        else:
            for n in names:
                self._chk(self.l.roms_hip_sync_to_host(abi.FIELD_ID[n]), "sync_to_host")
        return self.st

    def sync(self):
        self._chk(self.l.roms_hip_device_synchronize(), "device_synchronize")

    def timing(self, on=True):
        self.l.roms_hip_timing_enable(int(on))

    def set_halo_relay_gloo(self, dist, torch):
        """Route the halo exchange through torch.distributed (gloo, host memory) instead of
        RCCL: roms_hip_set_halo_relay with a callback doing the isend/irecv of one exchange.  Used to
        rehearse N tiles on fewer GPUs and as the pattern for a host-MPI relay.  Deliberately WITHOUT
        message tags: like RCCL, gloo then pairs the k-th send to a rank with the k-th receive from it,
        so the multi-process tests also check the order in which the library lists its messages (two
        tiles in a periodic direction exchange up to four messages per call)."""
        def cb(_user, nsend, send, nrecv, recv):
            try:
                reqs = []
                view = lambda m: torch.from_numpy(np.ctypeslib.as_array(m.buf, shape=(m.count,)))
                for q in range(nsend):
                    reqs.append(dist.isend(view(send[q]), dst=send[q].peer))
                for q in range(nrecv):
                    reqs.append(dist.irecv(view(recv[q]), src=recv[q].peer))
                for r in reqs:
                    r.wait()
                return 0
            except Exception as e:      # an exception must not unwind through the C frame
                import sys
                print(f"halo relay failed: {e!r}", file=sys.stderr, flush=True)
                return 1
        self._relay = RELAY_FN(cb)      # keep the thunk alive
        self._chk(self.l.roms_hip_set_halo_relay(self._relay, None), "set_halo_relay")

    def calib_stream(self, n_doubles):
        self._chk(self.l.roms_hip_calib_stream(n_doubles), "calib_stream")

    def last_ms(self, entry):
        return self.l.roms_hip_timing_last_ms(entry.encode())

    def row_metrics_state(self):
        """roms_hip_row_metrics_state: 0 not examined, 1 metric arrays independent of i (row table), 2 not."""
        self.l.roms_hip_row_metrics_state.restype = C.c_int
        return int(self.l.roms_hip_row_metrics_state())

    def graph_exchanges(self, on=True):
        """Several tiles over RCCL: replay LOOP_2D with its exchanges as one hipGraph (roms_hip.h)."""
        self._chk(self.l.roms_hip_graph_exchanges(int(on)), "graph_exchanges")

    def graph_exchanges_state(self):
        self.l.roms_hip_graph_exchanges_state.restype = C.c_int
        return int(self.l.roms_hip_graph_exchanges_state())

    def check_guards(self):
        """Raise if a kernel stored outside one of the device arrays (roms_hip_check_guards)."""
        self._chk(self.l.roms_hip_check_guards(), "check_guards")

    def close(self):
        # the library holds ONE context per process: only its current owner may tear it down
        # (a stale object being garbage-collected must not finalize its successor's context)
        if RomsHip._live is self:
            import sys
            import warnings
            RomsHip._live = None
            rc = self.l.roms_hip_check_guards()
            msg = self.l.roms_hip_last_error() if rc else None
            self.l.roms_hip_finalize()
            if rc:
                text = msg.decode() if msg else ""
                what = ("guard band damaged: " if "store outside" in text else "guard bands could not be checked: ") + text
                if sys.exc_info()[0] is not None:
                    # close() in a `finally:` while the body's exception is propagating: do not replace it
                    warnings.warn("roms_hip: " + what)
                else:
                    raise RuntimeError("roms_hip: " + what)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
