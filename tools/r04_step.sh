#!/bin/bash
# round-4 work loop on the GPU box: selected parity tests, then the reference-shaped call (config bench)
set -e
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > "$out/step_tests.log" 2>&1 || { tail -30 "$out/step_tests.log"; exit 1; }
tail -3 "$out/step_tests.log"
timeout -k 10 900 python3 -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "invariance or fixtures or pruned" >> "$out/step_tests.log" 2>&1 || { tail -30 "$out/step_tests.log"; exit 1; }
tail -3 "$out/step_tests.log"
timeout -k 10 600 python3 tools/config_bench.py --precision fp32 --eval-batch-size 32 256 > "$out/c3_step.jsonl" 2> "$out/c3_step.err" || { tail -20 "$out/c3_step.err"; exit 1; }
cat "$out/c3_step.jsonl"
