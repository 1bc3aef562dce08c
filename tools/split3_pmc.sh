#!/bin/bash
# Lab: time and HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate PMC passes) of the split-fp32 GEMM variants of
# tools/split3_probe.py, one shape.   tools/split3_pmc.sh c_fc "0 1 2 3 4 5"
set -e
shape=${1:-c_fc}; variants=${2:-"0 1"}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
export FITCLIP_HIP_LIB=$repo/tools/bin/libfitclip_hip_lab.so
cd /tmp && export TMPDIR=/tmp
for v in $variants; do
  export FITCLIP_LAB_SPLIT3=$v
  python3 "$repo/tools/split3_probe.py" 768 10 $shape 2>&1 | grep variant
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$out/pmc_s3"
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_s3" -o p -- python3 "$repo/tools/split3_probe.py" 768 3 $shape > /dev/null 2> "$out/pmc_s3.err" || { tail -5 "$out/pmc_s3.err"; exit 1; }
    f=$(find "$out/pmc_s3" -name "*counter_collection.csv" | head -1)
    python3 - "$f" $c $v <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "gemm_split3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
scale = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0   # gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md)
print(f"variant {sys.argv[3]} {sys.argv[2]}: {scale * 1024 * sum(vals) / len(vals) / 1e9:.3f} GB per launch over {len(vals)} launches")
PY
  done
done
rm -rf "$out/pmc_s3"
