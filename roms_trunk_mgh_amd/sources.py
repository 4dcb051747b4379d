"""Point sources / sinks (rivers): the host-side image of SOURCES(ng) (ROMS/Modules/mod_sources.F:56-80) with LuvSrc / LwSrc,
as an application's ana_psource.h or its river forcing file fills it.  A TileState carries one as `state.sources`;
both backends hand it to their library when they are built and whenever `set_sources` is called again (the reference
refreshes Qbar / Qsrc / Tsrc in set_data.F:124-160 every step)."""
import ctypes as C

import numpy as np

_IP = C.POINTER(C.c_int)
_DP = C.POINTER(C.c_double)


class Sources:
    def __init__(self, Isrc, Jsrc, Dsrc, Qbar, Qshape, Tsrc, LtracerSrc):
        """Isrc, Jsrc: grid indices of the u-face (Dsrc = 0) or v-face (Dsrc = 1) the source flows through, or of the
        cell it enters (Dsrc = 2, LwSrc); Qbar (m3/s, positive in the direction of increasing index, or into the
        cell); Qshape (Nsrc, N), the vertical distribution (sums to one);
        Tsrc (Nsrc, N, NT); LtracerSrc (NT)."""
        self.Isrc = np.ascontiguousarray(Isrc, dtype=np.int32)
        self.Jsrc = np.ascontiguousarray(Jsrc, dtype=np.int32)
        self.Dsrc = np.ascontiguousarray(Dsrc, dtype=np.float64)
        self.Qbar = np.ascontiguousarray(Qbar, dtype=np.float64)
        self.Qshape = np.asfortranarray(Qshape, dtype=np.float64)
        self.Tsrc = np.asfortranarray(Tsrc, dtype=np.float64)
        self.LtracerSrc = np.ascontiguousarray(LtracerSrc, dtype=np.int32)
        n = self.Isrc.size
        assert self.Jsrc.size == n and self.Dsrc.size == n and self.Qbar.size == n
        assert self.Qshape.shape[0] == n and self.Tsrc.shape[:2] == self.Qshape.shape
        assert self.Tsrc.shape[2] == self.LtracerSrc.size

    @property
    def n(self):
        return int(self.Isrc.size)

    def qsrc(self):
        """Qsrc(is,k) = Qbar(is) * Qshape(is,k), set_data.F:136-143"""
        return np.asfortranarray(self.Qbar[:, None] * self.Qshape)

    def c_args(self):
        """(Nsrc, Isrc, Jsrc, Dsrc, Qbar, Qsrc, Tsrc, LtracerSrc) as the C ABI takes them; keeps the buffers alive"""
        self._q = self.qsrc()
        return (self.n, self.Isrc.ctypes.data_as(_IP), self.Jsrc.ctypes.data_as(_IP), self.Dsrc.ctypes.data_as(_DP),
                self.Qbar.ctypes.data_as(_DP), self._q.ctypes.data_as(_DP), self.Tsrc.ctypes.data_as(_DP),
                self.LtracerSrc.ctypes.data_as(_IP))
