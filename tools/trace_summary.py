#!/usr/bin/env python
"""Per-shape summary of a rocprofv3 --kernel-trace CSV of bench.py: labels the block GEMMs of the VISUAL tower by their
position in the layer sequence (QKV, out_proj | c_fc | c_proj), splits off the much shorter text-tower launches of the
same kernels by duration, and prints average durations / TFLOP/s per shape plus the dispatch gaps (wall span of the
trace minus the summed kernel time).

    python tools/trace_summary.py <..._kernel_trace.csv> [frames_per_chunk] [fp32|bf16]

With a precision given, launches of the OTHER precision (bench.py plants the retrieval task with an fp32 model before a
bf16 run) are summed under "setup (other precision)" and otherwise ignored.
"""
import csv
import re
import statistics
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    want = sys.argv[3] if len(sys.argv) > 3 else None
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    M = chunk * 197
    shapes = {"qkv": (2304, 768), "out_proj": (768, 768), "c_fc": (3072, 768), "c_proj": (768, 3072)}
    labelled = []
    fp32 = False  # the exact-fp32 mode runs the same kernels ~8x longer: other duration cuts
    setup_other = 0.0
    for r in rows:
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        is_f32 = "<float" in name or "kernelIf" in name or "_f32_" in name or ", float>" in name
        is_b16 = "DF16b" in name or "__bf16" in name or "attn_bf16" in name
        if want and ((want == "bf16" and is_f32 and "fc::" in name) or (want == "fp32" and is_b16)):
            setup_other += dur
            continue
        if "gemm_pipelined_kernelIf" in name or "gemm_pipelined_kernel<float" in name:
            fp32 = True
        # rocprofv3 prints mangled or demangled names depending on its version: accept both spellings
        pipelined = re.search(r"gemm_pipelined_kernel<[^,]+, 256, 256, 2, 4, (\d),", name)
        if "gemm_pipelined_kernel" in name and ("Li256ELi256ELi2ELi4ELi0E" in name or (pipelined and pipelined.group(1) == "0")):
            label = "bias_gemm"
        elif "gemm_pipelined_kernel" in name and ("Li256ELi256ELi2ELi4ELi2E" in name or (pipelined and pipelined.group(1) == "2")):
            label = "resid_gemm"  # fp32: out_proj and c_proj update the residual stream in their epilogue
        elif "gemm_pipelined_kernel" in name and ("Li256ELi256ELi2ELi4ELi1E" in name or (pipelined and pipelined.group(1) == "1")):
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
        labelled.append([label, dur])
    # visual-tower launches of the pipelined GEMMs run for > 80 us at chunk >= 256 frames; the text tower's for < 60 us
    # (fp32: visual >= 900 us, text <= 350 us)
    # (fp32 bench step: one pass of 2048 frames; the shortest visual launch, out_proj, runs 3.5 ms - or 0.6 ms in the short
    # pass of a step that is split, e.g. 768 + 768 + 512 with `chunk_frames=768`)
    # (the reference-shaped call of bench.py --config c3, 128 frames: out_proj runs 0.26 ms, the text tower's launches < 0.1 ms)
    cut = (500.0 if chunk >= 512 else 150.0) if fp32 else 70.0
    cut_small = 100.0 if fp32 else 40.0
    seq = seq2 = 0
    fused = any(item[0] == "resid_gemm" for item in labelled)
    for item in labelled:
        if item[0] == "bias_gemm":
            if item[1] < cut:
                item[0] = "text / short-pass gemm"
            elif fused:
                item[0] = "qkv"
            else:
                item[0] = ("qkv", "out_proj", "c_proj")[seq % 3]
                seq += 1
        elif item[0] == "resid_gemm":
            if item[1] < cut:
                item[0] = "text / short-pass gemm"
            else:
                item[0] = ("out_proj", "c_proj")[seq2 % 2]
                seq2 += 1
        elif item[0] == "c_fc" and item[1] < cut:
            item[0] = "text / short-pass gemm"
        elif item[0] in ("attention", "add_layernorm") and item[1] < cut_small:
            item[0] = "text " + item[0]
    # a step may run passes of different sizes (fp32: whole tile rounds first, the rest after): launches well below the
    # label's longest belong to the short pass and are listed apart, so that the TF/s of the main pass are not diluted
    # (the reference level of a label is its 90th-percentile duration, not its maximum: the first launch of a process can run
    # twice as long as the others)
    by_label = defaultdict(list)
    for label, dur in labelled:
        by_label[label].append(dur)
    longest = defaultdict(float, {k: sorted(v)[int(0.9 * (len(v) - 1))] for k, v in by_label.items()})
    agg = defaultdict(list)
    for label, dur in labelled:
        if label in shapes and dur < 0.6 * longest[label]:
            label += " (short pass)"
        agg[label].append(dur)
    total = sum(d for _, d in labelled)
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    print(f"total kernel time {total / 1e3:.2f} ms over {len(labelled)} dispatches; trace span {span / 1e3:.2f} ms "
          f"(includes host-side setup between steps)")
    if setup_other:
        print(f"setup (other precision) {setup_other / 1e3:.2f} ms, not in the table")
    for k, durs in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        us, n = sum(durs), len(durs)
        extra = ""
        if k in shapes:
            N, K = shapes[k]
            extra = f"  (M={M}: {2.0 * M * N * K / (us / n * 1e-6) / 1e12:7.1f} TF/s avg, median {statistics.median(durs):.1f} us)"
        print(f"{k:18s} n={n:5d} avg={us / n:9.1f} us total={us / 1e3:8.2f} ms {100 * us / total:5.1f}%{extra}")


if __name__ == "__main__":
    main()
