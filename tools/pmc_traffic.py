#!/usr/bin/env python
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide prescribes) of
`bench.py` into per-launch HBM traffic of the dominant kernel and writes profiles/traffic_<round>.json.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores; both are in KiB.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <min duration us> out.json
"""
import csv
import json
import sys
from collections import defaultdict


def per_dispatch(path, counter, needle, min_us):
    """Counter values of the dispatches of kernel `needle` that ran for at least `min_us` (separates the visual-tower
    launches of a kernel instantiation from the much shorter text-tower ones)."""
    vals = []
    for r in csv.DictReader(open(path)):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"] and dur >= min_us:
            vals.append(float(r["Counter_Value"]))
    return vals


def main():
    fetch_csv, write_csv, needle, min_us, out = sys.argv[1:6]
    min_us = float(min_us)
    f = per_dispatch(fetch_csv, "FETCH_SIZE", needle, min_us)
    w = per_dispatch(write_csv, "WRITE_SIZE", needle, min_us)
    assert f and w, (len(f), len(w))
    fetch_b = 2.0 * 1024.0 * sum(f) / len(f)
    write_b = 1024.0 * sum(w) / len(w)
    res = {"kernel_substring": needle, "min_duration_us": min_us, "launches_fetch": len(f), "launches_write": len(w),
           "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
           "hbm_bytes_per_launch": fetch_b + write_b,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE x2 (gfx950), KiB units"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
