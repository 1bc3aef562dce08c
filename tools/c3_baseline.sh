#!/bin/bash
# Baseline of the reference-shaped call (32 clips x 4 frames per encoder call, encoder=wise): config bench + kernel trace.
set -e
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
python3 tools/config_bench.py --precision fp32 --eval-batch-size 32 256 > "$out/c3_base.jsonl" 2> "$out/c3_base.err"
cd /tmp && export TMPDIR=/tmp
export FITCLIP_OVERLAP_TEXT=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_c3_base" -o c3 -- python3 "$repo/tools/config_bench.py" --precision fp32 --eval-batch-size 32 > "$out/c3_base_prof.jsonl" 2> "$out/c3_base_prof.err"
cp "$out"/prof_c3_base/*/c3_kernel_stats.csv "$out/c3_base_kernel_stats.csv" 2>/dev/null || find "$out/prof_c3_base" -name '*kernel_stats.csv' -exec cp {} "$out/c3_base_kernel_stats.csv" \;
rm -rf "$out/prof_c3_base"
