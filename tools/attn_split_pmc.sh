#!/bin/bash
# HBM traffic of the split attention (attn_split_kernel<0>) from two rocprofv3 PMC passes over tools/bin/attn_split_lab
# (run on the GPU box from the repo root; writes gpurun_out/r03_attn_split_pmc.txt).  FETCH_SIZE x2 (gfx950), KiB units.
set -e
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/prof_attn_fetch" -o lab -- "$repo/tools/bin/attn_split_lab" 768 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/prof_attn_write" -o lab -- "$repo/tools/bin/attn_split_lab" 768 3 > /dev/null 2>&1
cd "$repo"
python3 - "$out" <<'PY'
import csv, glob, sys, statistics
out = sys.argv[1]
res = {}
for what, col, scale in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
    f = glob.glob(f"{out}/prof_attn_{what}/**/*counter_collection.csv", recursive=True)[0]
    for name in ("attn_split_kernel<0>", "attn_f32_blocks_kernel<8, 0, true>"):
        vals = [float(r["Counter_Value"]) * scale for r in csv.DictReader(open(f))
                if name in r["Kernel_Name"] and r["Counter_Name"] == col]
        res[(name, what)] = (statistics.mean(vals), len(vals))
with open(f"{out}/r03_attn_split_pmc.txt", "w") as o:
    o.write("768 frames x 197 tokens x 12 heads: algorithmic bytes fp32 q|k|v in 1.394 GB, x3 rows out 0.930 GB\n")
    for name in ("attn_split_kernel<0>", "attn_f32_blocks_kernel<8, 0, true>"):
        fb, nf = res[(name, "fetch")]; wb, nw = res[(name, "write")]
        o.write(f"{name}: fetch {fb / 1e9:.3f} GB ({nf} launches), write {wb / 1e9:.3f} GB ({nw} launches)\n")
print(open(f"{out}/r03_attn_split_pmc.txt").read())
PY
rm -rf "$out/prof_attn_fetch" "$out/prof_attn_write"
