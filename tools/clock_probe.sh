#!/bin/bash
# Samples sclk / power with rocm-smi while a sustained GEMM loop runs (tools/gemm_lab with many reps).
# usage: tools/clock_probe.sh <out.log> -- <command...>
out=$1; shift; shift
"$@" > "${out%.log}_cmd.log" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' ' >> "$out"
  echo >> "$out"
  sleep 0.2
done
wait $pid
