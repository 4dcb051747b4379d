"""Developer tool (GPU box): configuration 5 (BENCHMARK3 + 4 passive tracers, MPDATA for all six) --
hipEvent times of pre_step3d and step3d_t, for rocprofv3 --kernel-trace --stats."""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
import util  # noqa: E402
from roms_trunk_mgh_amd import hip  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "BENCHMARK3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
st = util.prepared_state(cfg, NT=6, overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"})
util.hz_weighted_tnew(st)
h = hip.RomsHip(st)
s = util.step_idx()
b = st.b
cells = b.Lm * b.Mm * b.N
for k in ("pre_step3d", "step3d_t"):
    h.timing(False)
    for _ in range(2):
        h.call(k, s)
    h.sync()
    h.timing(True)
    ms = []
    for _ in range(reps):
        h.call(k, s)
        ms.append(h.last_ms(k))
    ms.sort()
    med = ms[len(ms) // 2]
    comp = 8.0 * (4 * b.NT + 4) * cells
    print(f"{k}: median {med:.3f} ms  NT={b.NT}  compulsory {comp/1e9:.2f} GB -> {comp/med/1e6/8000:.3f} of 8 TB/s", flush=True)
h.close()
