#!/usr/bin/env python
"""Turns rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE and optionally SQ_* counters; SEPARATE runs, as the MI355X guide
prescribes) of `bench.py` into per-launch HBM traffic of the dominant kernel and writes profiles/traffic_<round>.json,
which bench.py reads for `roofline.traffic`.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores; both are in KiB.

    python tools/pmc_traffic.py --fetch F.csv --write W.csv [--sq S.csv] --kernel <substring> --min-us 250 \
        --shape 100864 3072 768 --precision bf16 --epilogue bias_quickgelu --out profiles/traffic_r01.json
"""
import argparse
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_dispatch(path, needle, min_us):
    """counter -> values over the dispatches of kernel `needle` that ran for at least `min_us` (separates the
    visual-tower launches of a kernel instantiation from the much shorter text-tower ones); also their durations."""
    vals, durs = defaultdict(list), []
    seen = set()
    for r in csv.DictReader(open(path)):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if needle in r["Kernel_Name"] and dur >= min_us:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                durs.append(dur)
    return vals, durs


def mean(v):
    return sum(v) / len(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--sq")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--min-us", type=float, default=250.0)
    ap.add_argument("--shape", type=int, nargs=3, required=True, metavar=("M", "N", "K"))
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--epilogue", default="bias_quickgelu")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    f, _ = per_dispatch(a.fetch, a.kernel, a.min_us)
    w, _ = per_dispatch(a.write, a.kernel, a.min_us)
    assert f["FETCH_SIZE"] and w["WRITE_SIZE"], (len(f), len(w))
    fetch_b = 2.0 * 1024.0 * mean(f["FETCH_SIZE"])
    write_b = 1024.0 * mean(w["WRITE_SIZE"])
    M, N, K = a.shape
    esz = 2 if a.precision == "bf16" else 4
    res = {"kernel_substring": a.kernel, "min_duration_us": a.min_us, "launches_fetch": len(f["FETCH_SIZE"]),
           "launches_write": len(w["WRITE_SIZE"]), "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
           "hbm_bytes_per_launch": fetch_b + write_b,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE x2 (gfx950), KiB units",
           "shape": [M, N, K], "precision": a.precision, "epilogue": a.epilogue,
           "algorithmic_bytes_per_launch": (M * K + N * K + M * N) * esz}
    from fitclip_amd.build import source_fingerprint
    res["source_fingerprint"] = source_fingerprint()  # bench.py refuses the file once the kernel sources change
    if a.sq:
        s, durs = per_dispatch(a.sq, a.kernel, a.min_us)
        sq = {k: round(mean(v)) for k, v in s.items()}
        sq["duration_us"] = round(mean(durs), 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
            # rocprofv3 sums both over the chip: busy cycles over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            cycles = sq["GRBM_GUI_ACTIVE"] / 8.0
            sq["sclk_ghz"] = round(cycles / (sq["duration_us"] * 1e3), 3)
            sq["mfma_busy_fraction"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles, 4)
        res["sq_counters_per_launch"] = sq
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
