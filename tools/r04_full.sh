#!/bin/bash
# full GPU suite + the c3 line
set -e
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
timeout -k 10 1500 python3 -m pytest tests -x -q -m gpu > "$out/full_tests.log" 2>&1 || { tail -40 "$out/full_tests.log"; exit 1; }
tail -3 "$out/full_tests.log"
timeout -k 10 900 python3 bench.py --config c3 --steps 2 --warmup 1 > "$out/c3_bench.json" 2> "$out/c3_bench.err" || { tail -30 "$out/c3_bench.err"; exit 1; }
cat "$out/c3_bench.json"
