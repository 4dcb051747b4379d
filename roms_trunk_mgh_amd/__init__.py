"""roms_trunk_mgh_amd -- MI355X-native ROMS nonlinear 3-D time-stepping hot path.

Host-side (Python) mirror of the reference interface for this path: tile
bounds (get_bounds.F), the module-array state of one tile, the main3d step
sequencer, and the ctypes binding of the C-ABI library libroms_hip.so whose
kernels are hand-written HIP for gfx950.
"""
from . import abi, bounds, state, ana  # noqa: F401
