"""Child process of tests/test_ref_pinning.py: loads oracle/_ref/<APP>/libref.so (the
reference's own Fortran, compiled from /root/reference by oracle/build_ref.sh) and
compares it with the C oracle on the same seeded inputs.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(config):
    import oracle
    import util
    from oracle import ref
    ov = {"tnu2": 300.0, "visc2": 800.0} if config != "SEAMOUNT" else {"tnu2": 300.0}
    st0 = util.prepared_state(config, overrides=ov)
    out = {}
    r = ref.Ref(st0.copy())
    bb = r.bounds()
    mine = st0.b.as_dict()
    out["bounds_mismatch"] = {k: (v, mine[k]) for k, v in bb.items() if mine[k] != v}
    nf, w1, w2 = r.set_weights(st0.p.ndtfast)
    n2 = 2 * st0.p.ndtfast
    out["nfast"] = [nf, st0.p.nfast]
    out["weights_maxdiff"] = float(max(max(abs(w1[i] - st0.p.weight1[i]), abs(w2[i] - st0.p.weight2[i])) for i in range(n2)))
    kernels = ["set_depth", "set_massflux", "set_zeta", "rho_eos", "prsgrd", "t3dmix2"]
    if config != "SEAMOUNT":
        kernels.append("uv3dmix2")
    s = util.step_idx()
    out["kernels"] = {}
    for k in kernels:
        st_r, st_o = st0.copy(), st0.copy()
        # detune so that every kernel has something to do
        for st in (st_r, st_o):
            st["Zt_avg1"] *= 1.3
            st["u"] *= 1.1
        rr = ref.Ref(st_r)
        rr.call(k, s)
        oracle.Oracle(st_o).call(k, s)
        diffs = util.compare_states(st_o, st_r)
        changed = util.compare_states(st_r, st0)
        out["kernels"][k] = {"max_rel_diff": max(diffs.values()) if diffs else 0.0, "fields_diff": sorted(diffs),
                             "changed": sorted(changed)}
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
