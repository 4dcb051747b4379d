"""Developer tool: time single hot-path entries on a named configuration with
the library's hipEvent timers.  Usage: python tools/bench_kernel.py [--lib PATH] BENCHMARK3 step3d_t [reps]"""
import sys
import time

import os  # noqa: E402
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
import util  # noqa: E402
from roms_trunk_mgh_amd import hip  # noqa: E402


def main():
    if "--lib" in sys.argv:                 # A/B of another build of the same ABI (python -m roms_trunk_mgh_amd._build <variant>)
        q = sys.argv.index("--lib")
        hip.use_library(sys.argv[q + 1])
        del sys.argv[q:q + 2]
    config = sys.argv[1] if len(sys.argv) > 1 else "BENCHMARK3"
    kernels = sys.argv[2].split(",") if len(sys.argv) > 2 else ["step3d_t"]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    sets = dict(a.split("=") for a in sys.argv[4:])      # e.g. uv_vis2=0 uv_adv=0
    t0 = time.time()
    if config.endswith("_MPDATA"):      # BASELINE.json configuration 5: six tracers, all MPDATA
        st = util.prepared_state(config[:-7], NT=6, overrides={"Hadv": "MPDATA", "Vadv": "MPDATA"})
    else:
        st = util.prepared_state(config)
    util.hz_weighted_tnew(st)
    print(f"state built in {time.time()-t0:.1f}s", flush=True)
    for k_, v_ in sets.items():
        setattr(st.p, k_, type(getattr(st.p, k_))(float(v_)))
    h = hip.RomsHip(st)
    s = util.step_idx()
    b = st.b
    cells = b.Lm * b.Mm * b.N
    for k in kernels:
        if k == "calib_stream":
            # known byte count for the PMC counters: 8*n read + 8*n written per launch
            n = (b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * b.N
            h.timing(True)
            for _ in range(reps):
                h.calib_stream(n)
            print(f"calib_stream: n={n} doubles, {8*n} B read + {8*n} B written per launch, "
                  f"{h.last_ms('calib_stream'):.4f} ms", flush=True)
            continue
        h.timing(False)
        run = (lambda: h.diag(s)) if k == "diag" else (lambda: h.call(k, s))
        for _ in range(3):
            run()
        h.sync()
        h.timing(True)
        ms = []
        for _ in range(reps):
            run()
            ms.append(h.last_ms(k))
        ms.sort()
        med = ms[len(ms) // 2]
        print(f"{k}: median {med:.4f} ms  min {ms[0]:.4f}  cells={cells}", flush=True)
        if k == "step3d_t":
            byts = 8.0 * (4 * b.NT + 4) * cells
            print(f"   algorithmic {byts/1e9:.3f} GB -> {byts/med/1e6:.1f} GB/s = {byts/med/1e6/8000:.3f} of 8 TB/s")
    h.close()


if __name__ == "__main__":
    main()
