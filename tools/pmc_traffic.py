#!/usr/bin/env python3
"""HBM-side traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are collected in
separate passes: they do not fit one, MI355X_MICROARCH.md "rocprofv3 PMC slots").

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <workload> <build note>

Correction (same guide, HBM section): on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes -> doubled.  Units are
KiB.  k_hist streams exactly one byte per residue and serves as the check of the correction."""
import csv, glob, json, sys
from collections import defaultdict


def collect(root, counter):
    tot, launches = defaultdict(float), defaultdict(set)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(launches[k]), len(launches[k])) for k in tot}


def main():
    fdir, wdir, workload, build = sys.argv[1:5]
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    keep = ("k_join", "k_rs_scatter", "k_rs_hist", "k_gather_ranges", "k_hist", "k_order_rows", "k_scan_apply")
    out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --output-format csv -- "
                      "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "note": "FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); k_hist "
                   "streams exactly one byte per residue and checks the correction",
           "workload": workload, "build": build, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not any(s in k for s in keep):
            continue
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": f, "launches_FETCH_SIZE": nf, "WRITE_SIZE_KB_per_launch": w,
                             "launches_WRITE_SIZE": nw, "hbm_bytes_per_launch_corrected": (2.0 * f + w) * 1024.0}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
