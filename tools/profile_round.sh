#!/bin/bash
# Reproduces the rocprofv3 evidence under profiles/ (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh r01
# 1. kernel trace + stats of the default bench command (CPU leg off);  2./3. separate PMC passes (FETCH_SIZE,
# WRITE_SIZE) and 4. an SQ pass for the dominant kernel.  Raw output goes to gpurun_out/prof_*; summaries to profiles/.
set -e
tag=${1:-r01}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# The profiled runs keep the two towers on ONE stream (FITCLIP_OVERLAP_TEXT=0): with the default two-stream forward the
# text-tower kernels overlap the visual tower's LayerNorm / attention kernels in time, and a per-kernel duration then
# no longer describes a kernel that owns the chip.  The dominant GEMM is not affected either way (nothing fits next to
# a persistent GEMM workgroup).
export FITCLIP_OVERLAP_TEXT=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_trace" -o bench -- python3 "$repo/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$out/prof_trace.json" 2> "$out/prof_trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/prof_fetch" -o bench -- python3 "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$out/prof_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/prof_write" -o bench -- python3 "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$out/prof_write.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/prof_sq" -o bench -- python3 "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$out/prof_sq.err" || echo "SQ pass failed (counters may need separate passes)"
cd "$repo"
find "$out/prof_trace" -name "*kernel_stats.csv" -exec cp {} "profiles/${tag}_final_bench_kernel_stats.csv" \;
trace=$(find "$out/prof_trace" -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py "$trace" > "profiles/${tag}_final_bench_trace_summary.txt"
cp "$out/prof_trace.json" "profiles/${tag}_final_bench_under_rocprof.json"
f=$(find "$out/prof_fetch" -name "*counter_collection.csv" | head -1)
w=$(find "$out/prof_write" -name "*counter_collection.csv" | head -1)
q=$(find "$out/prof_sq" -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py --fetch "$f" --write "$w" ${q:+--sq "$q"} --kernel gemm_pipelined_kernelIDF16bLi256ELi256ELi2ELi4ELi1 \
  --min-us 250 --shape 100864 3072 768 --precision bf16 --epilogue bias_quickgelu --out "profiles/traffic_${tag}.json"
cp "profiles/traffic_${tag}.json" "$out/"
