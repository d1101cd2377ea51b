#!/usr/bin/env python3
"""HBM traffic per launch of the library kernels from separate rocprofv3 --pmc passes.

usage: pmc_traffic.py <dir with fetch/ and write/ counter_collection CSVs> <out.json>

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 128-byte read
requests of wide coalesced streams as 64 bytes, so the read side is doubled before it is compared
with a byte count; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv
import glob
import json
import os
import sys


def mean_per_kernel(path, counter):
    agg = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "ic::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = name[:name.find("(")] if "(" in name else name
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in agg.items()}


def main():
    src, out = sys.argv[1], sys.argv[2]
    fetch, write = {}, {}
    for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        fetch.update(mean_per_kernel(path, "FETCH_SIZE"))
        write.update(mean_per_kernel(path, "WRITE_SIZE"))
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        res[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                  "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "correction": "read side doubled (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)"}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
