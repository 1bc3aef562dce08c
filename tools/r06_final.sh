#!/bin/bash
# Round-6 evidence, part 1 (run from the repo root on the GPU box): the GPU suite, smoke(), and the bench lines of every config.
# Part 2 is tools/profile_round.sh r06 {fp32,fp32x3,bf16,c3}.  Everything lands in gpurun_out/profiles_r06/.
set -e
repo=$(pwd); keep=$repo/gpurun_out/profiles_r06; mkdir -p "$keep"
part=${1:-all}   # tests | bench | configs | train | all
if [ "$part" = tests ] || [ "$part" = all ]; then
  timeout -k 10 1500 python3 -m pytest tests -q -m gpu > "$keep/r06_gpu_tests.log" 2>&1 || { tail -30 "$keep/r06_gpu_tests.log"; exit 1; }
  tail -2 "$keep/r06_gpu_tests.log"
  python3 __graft_entry__.py smoke > "$keep/r06_smoke.log" 2>&1 || { tail -20 "$keep/r06_smoke.log"; exit 1; }
  tail -2 "$keep/r06_smoke.log"
fi
if [ "$part" = bench ] || [ "$part" = all ]; then
  python3 bench.py --split6 > "$keep/r06_bench_default_run.json" 2> "$keep/r06_bench_default_run.err" || echo "default bench rc=$?"
  echo "c2 done"
fi
if [ "$part" = configs ] || [ "$part" = all ]; then
  python3 bench.py --config c3 --steps 3 --warmup 1 > "$keep/r06_bench_c3.json" 2> "$keep/r06_bench_c3.err"
  echo "c3 done"
  python3 bench.py --config c5 --total-clips 64 --steps 3 --warmup 1 > "$keep/r06_bench_c5_share.json" 2> "$keep/r06_bench_c5_share.err"
  echo "c5 (one rank's share of 8) done"
  python3 bench.py --config c5 --steps 2 --warmup 1 > "$keep/r06_bench_c5_n1.json" 2> "$keep/r06_bench_c5_n1.err"
  echo "c5 (512 clips on one GPU) done"
  python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > "$keep/r06_bench_c4_n1.json" 2> "$keep/r06_bench_c4_n1.err"
  echo "c4 done"
  for p in fp32x3 fp32x6; do python3 tools/config_bench.py --precision $p >> "$keep/r06_config_bench_split_modes.jsonl" 2>> "$keep/config_bench.err"; done
  echo "config bench (split modes) done"
  rm -f "$keep"/*.err
fi
