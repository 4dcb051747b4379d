#!/usr/bin/env python3
"""Developer tool: instruction mix of the kernels of one .hip file (device assembly via hipcc -S).
Usage: tools/isa_count.py roms_trunk_mgh_amd/csrc/k_mpdata.hip [name-filter]"""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                "-I" + R + "/include", "-I" + R + "/roms_trunk_mgh_amd/csrc", "-S", "--cuda-device-only", src, "-o", out] + sys.argv[3:],
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
funcs = list(re.finditer(r'^(_Z[^:\n]*):', s, re.M))
for q, m in enumerate(funcs):
    name = m.group(1)
    if filt not in name:
        continue
    body = s[m.end(): funcs[q + 1].start() if q + 1 < len(funcs) else len(s)]
    body = body.split("s_endpgm")[0]
    lines = [l.strip() for l in body.split("\n")]
    lines = [l for l in lines if l and not l.startswith((".", ";")) and not l.endswith(":")]
    v = [l for l in lines if l.startswith("v_")]
    cnt = lambda pred: sum(1 for l in lines if pred(l))
    print(f"{name[:70]}: total {len(lines)} valu {len(v)} f64 {cnt(lambda l: '_f64' in l)} "
          f"gload {cnt(lambda l: l.startswith(('global_load', 'flat_load')))} gstore {cnt(lambda l: l.startswith(('global_store', 'flat_store')))} "
          f"ds {cnt(lambda l: l.startswith('ds_'))} div_scale {cnt(lambda l: 'v_div_scale' in l)} rcp {cnt(lambda l: 'v_rcp' in l)} "
          f"barrier {cnt(lambda l: 's_barrier' in l)} scratch {cnt(lambda l: 'scratch_' in l)}")
