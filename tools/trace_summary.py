#!/usr/bin/env python
"""Per-shape summary of a rocprofv3 --kernel-trace CSV of bench.py: labels the block GEMMs by their position in the
layer sequence (QKV, out_proj | c_fc | c_proj) and prints average durations / TFLOP/s per shape.

    python tools/trace_summary.py <..._kernel_trace.csv> [frames_per_chunk]
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    agg = defaultdict(lambda: [0.0, 0])
    seq = 0
    M = chunk * 197
    shapes = {"qkv": (2304, 768), "out_proj": (768, 768), "c_fc": (3072, 768), "c_proj": (768, 3072)}
    total = 0.0
    for r in rows:
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        total += dur
        wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else 0
        if "gemm_pipelined_kernel" in name and "Li256ELi256ELi2ELi4ELi0E" in name and "DF16b" in name:
            label = ("qkv", "out_proj", "c_proj")[seq % 3]
            seq += 1
        elif "gemm_pipelined_kernel" in name and "Li256ELi256ELi2ELi4ELi1E" in name:
            label = "c_fc"
        elif "attn_" in name:
            label = "attention"
        elif "add_layernorm" in name:
            label = "add_layernorm"
        elif "layernorm" in name:
            label = "layernorm"
        elif "gemm" in name:
            label = "gemm(other)"
        else:
            label = "other"
        agg[label][0] += dur
        agg[label][1] += 1
    print(f"total kernel time {total / 1e3:.2f} ms over {len(rows)} dispatches")
    for k, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        extra = ""
        if k in shapes:
            N, K = shapes[k]
            extra = f"  (if M={M}: {2.0 * M * N * K / (us / n * 1e-6) / 1e12:7.1f} TF/s)"
        print(f"{k:14s} n={n:5d} avg={us / n:9.1f} us total={us / 1e3:8.2f} ms {100 * us / total:5.1f}%{extra}")


if __name__ == "__main__":
    main()
