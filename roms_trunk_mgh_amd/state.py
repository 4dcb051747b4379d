"""Host-side mirror of the reference's tile state: the module arrays of
mod_ocean / mod_grid / mod_coupling / mod_mixing / mod_forces for ONE tile, as
Fortran-ordered float64 numpy arrays with the common horizontal extents
LBi:UBi, LBj:UBj (ROMS/Modules/mod_ocean.F:318-411 allocate them that way).
"""
import ctypes as C

import numpy as np

from . import abi


class TileState:
    """All hot-path arrays of one tile + its bounds and parameter block."""

    def __init__(self, bounds, params):
        self.b = bounds
        self.p = params
        self.ni = bounds.UBi - bounds.LBi + 1
        self.nj = bounds.UBj - bounds.LBj + 1
        self.arr = {}
        for name, kind, _ in abi.FIELDS:
            trail = abi.trailing_shape(kind, bounds.N, bounds.NT, bounds.NAT)
            self.arr[name] = np.zeros((self.ni, self.nj) + trail, dtype=np.float64, order="F")

    # -- index helpers (Fortran index -> numpy offset) ---------------------
    def I(self, a, b=None):
        """slice for Fortran i-range a:b (inclusive) or scalar offset."""
        if b is None:
            return a - self.b.LBi
        return slice(a - self.b.LBi, b - self.b.LBi + 1)

    def J(self, a, b=None):
        if b is None:
            return a - self.b.LBj
        return slice(a - self.b.LBj, b - self.b.LBj + 1)

    def __getitem__(self, name):
        return self.arr[name]

    def __setitem__(self, name, value):
        self.arr[name][...] = value

    def fields_struct(self):
        f = abi.Fields()
        for name, _, _ in abi.FIELDS:
            a = self.arr[name]
            assert a.flags["F_CONTIGUOUS"] and a.dtype == np.float64
            setattr(f, name, a.ctypes.data_as(C.POINTER(C.c_double)))
        return f

    def copy(self):
        other = TileState.__new__(TileState)
        other.b, other.p, other.ni, other.nj = self.b, self.p, self.ni, self.nj
        other.arr = {k: v.copy(order="F") for k, v in self.arr.items()}
        for extra in ("cfg", "lonr", "latr", "z_r0", "z_w0", "sources"):
            if hasattr(self, extra):
                setattr(other, extra, getattr(self, extra))
        return other

    def interior(self, name):
        """View of the interior RHO range Istr:Iend,Jstr:Jend (all trailing)."""
        b = self.b
        return self.arr[name][self.I(b.Istr, b.Iend), self.J(b.Jstr, b.Jend)]


def rel_rms(x, ref, floor=0.0):
    """relRMS(x) = sqrt(sum((x-ref)^2)/sum(ref^2));
    `floor` is the stated magnitude used when the reference RMS is ~0."""
    num = float(np.sqrt(np.mean((np.asarray(x) - np.asarray(ref)) ** 2)))
    den = float(np.sqrt(np.mean(np.asarray(ref) ** 2)))
    return num / max(den, floor, 1e-300)
