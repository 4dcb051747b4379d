"""Build libroms_hip.so (gfx950) with hipcc, in-tree, one object per .hip file."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libroms_hip.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: keep the reference's operation order (no FMA contraction);
# these kernels are HBM-bound, FMA throughput is irrelevant.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-variable", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stale(src, obj, deps):
    if not os.path.exists(obj):
        return True
    mt = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > mt for d in [src] + deps)


def build(verbose=False, force=False, variant=None, extra=()):
    """variant / extra: developer A/B builds -- objects in csrc/_obj_<variant>, library libroms_hip_<variant>.so,
    compiled with the extra flags (e.g. -DUVCOL_FLAT); the product build is variant=None."""
    global OBJ, LIB
    obj0, lib0 = OBJ, LIB
    if variant:
        OBJ = os.path.join(HERE, "csrc", "_obj_" + variant)
        LIB = os.path.join(HERE, f"libroms_hip_{variant}.so")
    try:
        return _build(verbose, force, list(extra))
    finally:
        OBJ, LIB = obj0, lib0


def _build(verbose, force, extra):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    jobs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        if force or _stale(src, obj, deps):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r

    with ThreadPoolExecutor(max_workers=6) as ex:
        for src, r in ex.map(cc, jobs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed on {src}")
            if verbose and r.stderr.strip():
                sys.stderr.write(r.stderr)
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in srcs]
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    if args:      # python -m roms_trunk_mgh_amd._build <variant> [-DFLAG ...]
        print(build(verbose=False, force="--force" in sys.argv, variant=args[0], extra=args[1:]))
    else:
        print(build(verbose=True, force="--force" in sys.argv))
