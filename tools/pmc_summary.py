#!/usr/bin/env python3
"""Developer tool: condense the two rocprofv3 PMC passes written by tools/gpu_profile.sh
(gpurun_out/prof_TAG/pmc_fetch, pmc_write) into profiles/TAG/pmc_summary.csv and copy the
kernel statistics of the bench pass next to it.  FETCH_SIZE is in KiB and under-counts 8-byte
per-lane loads by 2 on gfx950 (calibrated by k_calib_stream in the same run: its corrected
fetch equals its write), WRITE_SIZE is in KiB and exact."""
import csv, collections, os, shutil, sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
dst = f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)


def per_kernel(path, counter):
    tot = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            d = tot.setdefault(k, [set(), 0.0])
            d[0].add(r["Dispatch_Id"])
            d[1] += float(r["Counter_Value"])
    return {k: (len(v[0]), v[1]) for k, v in tot.items()}


fe = per_kernel(f"{src}/pmc_fetch/{tag}_counter_collection.csv", "FETCH_SIZE")
wr = per_kernel(f"{src}/pmc_write/{tag}_counter_collection.csv", "WRITE_SIZE")
with open(f"{dst}/pmc_summary.csv", "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KiB_raw,FETCH_bytes_corrected_x2,WRITE_SIZE_KiB,WRITE_bytes,total_bytes\n")
    for k, (n, fv) in fe.items():
        wv = wr.get(k, (n, 0.0))[1]
        fb = fv / n * 1024 * 2
        wb = wv / n * 1024
        f.write(f'"{k}",{n},{fv / n:.1f},{fb:.0f},{wv / n:.1f},{wb:.0f},{fb + wb:.0f}\n')
shutil.copy(f"{src}/stats/{tag}_kernel_stats.csv", f"{dst}/kernel_stats.csv")
shutil.copy(f"{src}/bench_under_rocprof.json", f"{dst}/bench_under_rocprof.json")
print(open(f"{dst}/pmc_summary.csv").read())
