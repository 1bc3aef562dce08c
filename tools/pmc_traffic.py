#!/usr/bin/env python
"""Turns rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE and optionally SQ_* counters; SEPARATE runs, as the MI355X guide
prescribes) of `bench.py` into per-launch HBM traffic of the dominant kernel and writes profiles/traffic_<round>.json,
which bench.py reads for `roofline.traffic`.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores; both are in KiB.

    python tools/pmc_traffic.py --fetch F.csv --write W.csv [--sq S.csv] --precision fp32 --out profiles/traffic_r02_fp32.json \
        --spec "gemm_pipelined_kernel<float, 256, 256, 2, 4, 1,|4000|1e9|201728|3072|768|bias_quickgelu" --spec ...
"""
import argparse
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_dispatch(path, needle, min_us, max_us=1e18):
    """counter -> values over the dispatches of kernel `needle` that ran for min_us <= t < max_us (separates the
    visual-tower launches of a kernel instantiation from the much shorter text-tower ones, and the shapes that share one
    instantiation); also their durations."""
    vals, durs = defaultdict(list), []
    seen = set()
    for r in csv.DictReader(open(path)):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if needle in r["Kernel_Name"] and min_us <= dur < max_us:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                durs.append(dur)
    return vals, durs


def mean(v):
    return sum(v) / len(v)


def one_kernel(a, needle, min_us, max_us, shape, epilogue):
    f, _ = per_dispatch(a.fetch, needle, min_us, max_us)
    w, _ = per_dispatch(a.write, needle, min_us, max_us)
    assert f["FETCH_SIZE"] and w["WRITE_SIZE"], (needle, len(f), len(w))
    fetch_b = 2.0 * 1024.0 * mean(f["FETCH_SIZE"])
    write_b = 1024.0 * mean(w["WRITE_SIZE"])
    M, N, K = shape
    esz = 2 if a.precision in ("bf16", "fp32x6", "fp32x3") else 4
    # split-fp32 GEMMs (precision fp32x6; K is given as 6 K, the bf16 products): three bf16 planes per operand value = 6 bytes;
    # fp32 or three-plane rows out
    out_b = M * N * (4 if epilogue.endswith("f32_out") or epilogue.endswith("x2_out") else 6 if epilogue.endswith("x3_out") else esz)
    if "residual" in epilogue:  # C += ..: the fp32 tile is read as well as written
        out_b = 2 * M * N * 4
    res = {"kernel_substring": needle, "min_duration_us": min_us, "max_duration_us": max_us,
           "launches_fetch": len(f["FETCH_SIZE"]), "launches_write": len(w["WRITE_SIZE"]),
           "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
           "shape": [M, N, K], "precision": a.precision, "epilogue": epilogue,
           # (fp32x6: K is 6 K, three bf16 planes = 6 bytes per operand value; fp32x3: K is 3 K, two fp16 planes = 4 bytes)
           "algorithmic_bytes_per_launch": ((M + N) * (K // 6) * 6 if a.precision == "fp32x6" else (M + N) * (K // 3) * 4
                                            if a.precision == "fp32x3" else (M * K + N * K) * esz) + out_b}
    if a.sq:
        s, durs = per_dispatch(a.sq, needle, min_us, max_us)
        sq = {k: round(mean(v)) for k, v in s.items()}
        sq["duration_us"] = round(mean(durs), 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
            # rocprofv3 sums both over the chip: busy cycles over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            cycles = sq["GRBM_GUI_ACTIVE"] / 8.0
            sq["sclk_ghz"] = round(cycles / (sq["duration_us"] * 1e3), 3)
            sq["mfma_busy_fraction"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles, 4)
        res["sq_counters_per_launch"] = sq
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--sq")
    ap.add_argument("--spec", action="append", required=True,
                    help="kernel-name substring|min_us|max_us|M|N|K|epilogue  (one per kernel; the duration window "
                         "separates the shapes that share one instantiation and the short text-tower launches)")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    kernels = []
    for spec in a.spec:
        needle, lo, hi, M, N, K, epi = spec.split("|")
        kernels.append(one_kernel(a, needle, float(lo), float(hi), (int(M), int(N), int(K)), epi))
    from fitclip_amd.build import source_fingerprint
    res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE x2 (gfx950), KiB units",
           "precision": a.precision, "kernels": kernels,
           "source_fingerprint": source_fingerprint()}  # bench.py refuses the file once the kernel sources change
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
