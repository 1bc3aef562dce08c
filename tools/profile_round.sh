#!/bin/bash
# Reproduces the rocprofv3 evidence under profiles/ (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh r04 fp32        # the headline precision
#   tools/profile_round.sh r04 bf16        # the secondary mode
#   tools/profile_round.sh r05 fp32x3      # the split-fp32 leg (fp32 bench run with `fp32_split_mode` on: three fp16 products)
#   tools/profile_round.sh r05 fp32x6      # the six-product split mode of rounds 2-4 (fp32 bench run with --split6)
#   tools/profile_round.sh r04 c3          # bench.py --config c3 (encoder=wise, eval batches of 32 clips = 128 frames per call)
# 1. kernel trace + stats of the bench command for that precision (CPU leg and the other precision off);
# 2./3. separate PMC passes (FETCH_SIZE, WRITE_SIZE) and 4. an SQ pass for the dominant kernel (c_fc + QuickGELU GEMM).
# Raw output goes to gpurun_out/prof_*; summaries to profiles/ AND gpurun_out/profiles_<tag>/ (the latter travels back).
# The program stays directly after `--` (no env / bash -c hop: the profiler has already initialised the GPU).
set -e
tag=${1:-r04}
prec=${2:-fp32}
repo=$(pwd)
out=$repo/gpurun_out
keep=$out/profiles_${tag}
mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
# The profiled runs keep the two towers on ONE stream (FITCLIP_OVERLAP_TEXT=0): with the default two-stream forward the
# text-tower kernels overlap the visual tower's LayerNorm / attention kernels in time, and a per-kernel duration then
# no longer describes a kernel that owns the chip.  The dominant GEMM is not affected either way (nothing fits next to
# a persistent GEMM workgroup).
export FITCLIP_OVERLAP_TEXT=0
# kernel-name substring of the dominant kernel (c_fc + QuickGELU, pipelined 256x256) as rocprofv3 prints it: demangled for
# float, still mangled for __bf16 instantiations
# (fp32 and bf16: the 2048 frames of a bench step run as ONE pass)
# c_fc (+QuickGELU) has its own instantiation; c_proj shares one with out_proj (the residual epilogue, 2) and is told apart by
# its duration window (fp32 @ 2048 frames: c_proj 13.2 ms, out_proj 3.5; bf16 @ 2048: 1.8 / 0.55)
bench_args=""
if [ "$prec" = c3 ]; then
  # the reference-shaped call: 128 frames per encoder call (M = 25 216).  c_proj and out_proj share one instantiation (residual
  # epilogue, tail of 64-row tiles: HT = 1) and are told apart by duration (c_proj 0.93 ms, out_proj 0.26 ms); c_fc (QuickGELU)
  # has its own (0.87 ms; the text tower's c_fc runs on the 64 x 64 ring kernel, another name)
  steps=1; chunk=128; rows=$((chunk * 197)); bench_args="--config c3"
  spec_fc="gemm_pipelined_kernel<float, 256, 256, 2, 4, 1,|600|2000|$rows|3072|768|bias_quickgelu"
  spec_proj="gemm_pipelined_kernel<float, 256, 256, 2, 4, 2,|700|2000|$rows|768|3072|bias_residual"
elif [ "$prec" = fp32 ]; then
  steps=3; chunk=2048; rows=$((chunk * 197))
  spec_fc="gemm_pipelined_kernel<float, 256, 256, 2, 4, 1,|9000|1e9|$rows|3072|768|bias_quickgelu"
  spec_proj="gemm_pipelined_kernel<float, 256, 256, 2, 4, 2,|9800|1e9|$rows|768|3072|bias_residual"
elif [ "$prec" = bf16 ]; then
  steps=5; chunk=2048; rows=$((chunk * 197))
  spec_fc="gemm_pipelined_kernelIDF16bLi256ELi256ELi2ELi4ELi1E|1000|1e9|$rows|3072|768|bias_quickgelu"
  spec_proj="gemm_pipelined_kernelIDF16bLi256ELi256ELi2ELi4ELi2E|1000|1e9|$rows|768|3072|bias_residual"
elif [ "$prec" = fp32x3 ]; then
  # fp32x3 = the split-fp32 leg of the fp32 bench run: the 2048 frames of a step run as ONE pass (round 6; 1024 + 1024 in round 5);
  # two-plane fp16 operands, three fp16 products per fp32 product (the K below is 3 K).  c_fc (QuickGELU + x2 rows) has its own
  # instantiation (epilogue 10; 4.45 ms at 2048 frames); c_proj shares epilogue 8 with out_proj and is the only one of the two
  # above 3 ms (at 2048 frames: c_proj 4.0, out_proj 1.2)
  steps=3; chunk=2048; rows=$((chunk * 197))
  spec_fc="gemm_split2_kernel<10,|3500|1e9|$rows|3072|2304|bias_quickgelu_x2_out"
  spec_proj="gemm_split2_kernel<8,|3000|1e9|$rows|768|9216|bias_residual_f32_out"
else
  # fp32x6 = the split-fp32 leg of the fp32 bench run (the 2048 frames of a step run as 768 + 768 + 512; three-plane operands,
  # six bf16 products per fp32 product: the K below is 6 K): c_fc with the QuickGELU + x3 epilogue has its own instantiation
  # (epilogue 7; 3.4 ms at 768 frames, 2.3 at 512); c_proj shares epilogue 8 (residual update in place) with out_proj and is the
  # only one of the two above 2.9 ms (at 768 frames: c_proj 3.2, out_proj 0.85; QKV, epilogue 6: 2.55)
  steps=3; chunk=768; rows=$((chunk * 197))
  spec_fc="gemm_split3_kernel<7,|3000|1e9|$rows|3072|4608|bias_quickgelu_x3_out"
  spec_proj="gemm_split3_kernel<8,|2900|1e9|$rows|768|18432|bias_residual_f32_out"
fi
pmc_extra=""
if [ "$prec" = c3 ]; then
  common="--config c3 --no-cpu-baseline --headline-only"
  # (the evaluate loop enqueues a whole epoch without a host synchronisation; under --pmc a 1024-clip run of it aborted the queue
  # with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT in round 5, and the SAME run with the queue drained after every eval batch
  # (--sync-batches) passes - 18.7 k dispatches profiled: it is the depth of the host's run-ahead over the intercepted queue that
  # breaks, not a kernel and not the dispatch count (tools/pmc_scaling_probe.py, profiles/r06_pmc_scaling_probe.log).  The PMC
  # passes therefore run 8 batches of 32 clips - the same kernels on the same shapes - and synchronise per batch)
  pmc_extra="--total-clips 256 --sync-batches"
elif [ "$prec" = fp32x3 ]; then
  common="--precision fp32 --no-bf16-mode --no-cpu-baseline --no-train-leg"
elif [ "$prec" = fp32x6 ]; then
  common="--precision fp32 --no-bf16-mode --no-cpu-baseline --no-train-leg --split6"
else
  common="--precision $prec --no-bf16-mode --no-split-mode --no-cpu-baseline --no-train-leg"
fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_trace_$prec" -o bench -- python3 "$repo/bench.py" --steps $steps --warmup $([ "$prec" = c3 ] && echo 1 || echo 2) $common > "$out/prof_trace_$prec.json" 2> "$out/prof_trace_$prec.err"
echo "trace pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/prof_fetch_$prec" -o bench -- python3 "$repo/bench.py" --steps $([ "$prec" = c3 ] && echo 1 || echo 2) --warmup 1 --no-plant $common $pmc_extra > /dev/null 2> "$out/prof_fetch_$prec.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/prof_write_$prec" -o bench -- python3 "$repo/bench.py" --steps $([ "$prec" = c3 ] && echo 1 || echo 2) --warmup 1 --no-plant $common $pmc_extra > /dev/null 2> "$out/prof_write_$prec.err"
echo "write pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/prof_sq_$prec" -o bench -- python3 "$repo/bench.py" --steps $([ "$prec" = c3 ] && echo 1 || echo 2) --warmup 1 --no-plant $common $pmc_extra > /dev/null 2> "$out/prof_sq_$prec.err" || echo "SQ pass failed (counters may need separate passes)"
echo "sq pass done"
cd "$repo"
find "$out/prof_trace_$prec" -name "*kernel_stats.csv" -exec cp {} "$keep/${tag}_bench_${prec}_kernel_stats.csv" \;
trace=$(find "$out/prof_trace_$prec" -name "*kernel_trace.csv" | head -1)
if [ "$prec" != fp32x6 ] && [ "$prec" != fp32x3 ]; then  # (the split legs share their trace with the fp32 leg: only the kernel table is kept)
  python3 tools/trace_summary.py "$trace" $chunk $([ "$prec" = c3 ] && echo fp32 || echo $prec) > "$keep/${tag}_bench_${prec}_trace_summary.txt"
fi
cp "$out/prof_trace_$prec.json" "$keep/${tag}_bench_${prec}_under_rocprof.json"
f=$(find "$out/prof_fetch_$prec" -name "*counter_collection.csv" | head -1)
w=$(find "$out/prof_write_$prec" -name "*counter_collection.csv" | head -1)
q=$(find "$out/prof_sq_$prec" -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py --fetch "$f" --write "$w" ${q:+--sq "$q"} --spec "$spec_fc" --spec "$spec_proj" \
  --precision $([ "$prec" = c3 ] && echo fp32 || echo $prec) --out "$keep/traffic_${tag}_$([ "$prec" = c3 ] && echo fp32_c3 || echo $prec).json"
cp "$keep"/* profiles/
# the raw traces are large: keep only the summaries for the trip back
rm -rf "$out/prof_trace_$prec" "$out/prof_fetch_$prec" "$out/prof_write_$prec" "$out/prof_sq_$prec"
