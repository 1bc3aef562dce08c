#!/bin/bash
# Round-4 evidence, part 1 (run from the repo root on the GPU box): the GPU suite, smoke(), and the bench lines of every config.
# Part 2 is tools/profile_round.sh r04 {fp32,fp32x6,bf16,c3}.  Everything lands in gpurun_out/profiles_r04/.
set -e
repo=$(pwd); keep=$repo/gpurun_out/profiles_r04; mkdir -p "$keep"
part=${1:-all}   # tests | bench | train | all
if [ "$part" = tests ] || [ "$part" = all ]; then
  timeout -k 10 1500 python3 -m pytest tests -q -m gpu > "$keep/r04_gpu_tests.log" 2>&1 || { tail -30 "$keep/r04_gpu_tests.log"; exit 1; }
  tail -2 "$keep/r04_gpu_tests.log"
  python3 __graft_entry__.py smoke > "$keep/r04_smoke.log" 2>&1 || { tail -20 "$keep/r04_smoke.log"; exit 1; }
  tail -2 "$keep/r04_smoke.log"
fi
if [ "$part" = bench ] || [ "$part" = all ]; then
  python3 bench.py > "$keep/r04_bench_default_run.json" 2> "$keep/r04_bench_default_run.err" || echo "default bench rc=$?"
  echo "c2 done"
  python3 bench.py --config c3 --steps 3 --warmup 1 > "$keep/r04_bench_c3.json" 2> "$keep/r04_bench_c3.err"
  echo "c3 done"
  python3 bench.py --config c5 --steps 2 --warmup 1 > "$keep/r04_bench_c5_n1.json" 2> "$keep/r04_bench_c5_n1.err"
  echo "c5 (512 clips on one GPU) done"
  python3 bench.py --config c5 --total-clips 64 --steps 3 --warmup 1 > "$keep/r04_bench_c5_share.json" 2> "$keep/r04_bench_c5_share.err"
  echo "c5 (one rank's share of 8) done"
  python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > "$keep/r04_bench_c4_n1.json" 2> "$keep/r04_bench_c4_n1.err"
  echo "c4 done"
  rm -f "$keep"/*.err
fi
if [ "$part" = train ] || [ "$part" = all ]; then
  # the KD training step at one rank's share of configs[4], plain and under the kernel trace (program directly after --)
  python3 tools/train_bench.py --steps 4 > "$keep/r04_train_step_plain.json" 2> "$keep/train_plain.err"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$keep/prof_train" -o train -- python3 "$repo/tools/train_bench.py" --steps 3 > "$keep/r04_train_step.json" 2> "$keep/train_prof.err")
  cp "$keep"/prof_train/*/train_kernel_stats.csv "$keep/r04_train_step_kernel_stats.csv" 2>/dev/null || cp "$keep"/prof_train/train_kernel_stats.csv "$keep/r04_train_step_kernel_stats.csv"
  rm -rf "$keep/prof_train" "$keep"/train_*.err
  tail -c 400 "$keep/r04_train_step_plain.json"
fi
