#!/bin/bash
set -e
repo=$(pwd); out=$repo/gpurun_out; cd /tmp; export TMPDIR=/tmp
for s in 0 2 8; do
  export FITCLIP_GEMM_SCHED=$s
  rocprofv3 --kernel-trace --output-format csv -d "$out/ab_s$s" -o bench -- python3 "$repo/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$out/ab_s$s.json" 2> "$out/ab_s$s.err"
  python3 "$repo/tools/trace_summary.py" "$out/ab_s$s/bench_kernel_trace.csv" > "$out/ab_s$s.txt"
  echo "== sched $s"; grep -E "^(c_fc|c_proj|qkv|out_proj) " "$out/ab_s$s.txt"
done
