import os, sys
import numpy as np
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
import util, oracle
from roms_trunk_mgh_amd import hip
config, kernel = sys.argv[1], sys.argv[2]
st0 = util.prepared_state(config, mask="island")
if kernel == "step3d_t":
    util.hz_weighted_tnew(st0)
st_o, st_h = st0.copy(), st0.copy()
s = util.step_idx(iic=5)
oracle.Oracle(st_o).call(kernel, s)
h = hip.RomsHip(st_h); h.call(kernel, s); h.to_host(); h.close()
b = st0.b
for name in st0.arr:
    d = np.abs(st_h[name] - st_o[name])
    if d.max() > 0:
        idx = np.unravel_index(np.argmax(d), d.shape)
        i, j = idx[0] + b.LBi, idx[1] + b.LBj
        print(name, "maxdiff", d.max(), "at", (i, j) + idx[2:], "hip", st_h[name][idx], "oracle", st_o[name][idx], "n_diff", int((d > 0).sum()))
        print(" rmask 3x3 around (rows j+1..j-1):")
        for jj in (j + 1, j, j - 1):
            print("  ", [st0["rmask"][ii - b.LBi, jj - b.LBj] for ii in range(i - 2, i + 3)])
        nz = np.argwhere(d > 0)[:12]
        print(" first diffs:", [tuple(int(x) for x in q) for q in nz])
