"""Developer tool (GPU): one step3d_t call with six MPDATA tracers, exact vs refined-reciprocal quotients in
mpdata_adiff (roms_params_t.mpdata_fast): prints how far the two results are apart."""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
import util  # noqa: E402
from roms_trunk_mgh_amd import hip  # noqa: E402

out = {}
for fast in (0, 1):
    st = util.prepared_state("BENCHMARK_TINY", NT=6, overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"})
    util.hz_weighted_tnew(st)
    st.p.mpdata_fast = fast
    h = hip.RomsHip(st)
    h.call("step3d_t", util.step_idx())
    h.to_host(["t"])
    h.close()
    out[fast] = st["t"][:, :, :, 1, :].copy()
d = np.abs(out[1] - out[0])
print("max |t_fast - t_exact| =", d.max(), " cells differing:", int((d > 0).sum()), "of", d.size,
      " max |t| =", np.abs(out[0]).max())
