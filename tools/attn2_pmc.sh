#!/bin/bash
# Lab: PMC passes over the three-product attention (attn_split2_kernel) at a bench pass (2048 sequences x 197 tokens x 12 heads) through
# tools/attn2_probe.py: matrix-pipe busy and shader clock, the wave-cycle split, vector / LDS instruction counts, LDS bank conflicts, and
# the memory-side traffic (FETCH_SIZE x 2 on gfx950, KiB; separate passes).     tools/attn2_pmc.sh [n_seq] [S]   (repo root, GPU box)
set -e
n=${1:-2048}; S=${2:-197}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; rm -rf "$out/pmc_attn2_$name"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_attn2_$name" -o p -- python3 "$repo/tools/attn2_probe.py" $n $S > "$out/pmc_attn2_$name.log" 2> "$out/pmc_attn2_$name.err" || { tail -5 "$out/pmc_attn2_$name.err"; return 1; }; }
pass sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES || echo "sq2 pass failed"
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
agg, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
for name in ("sq", "sq2", "fetch", "write"):
    for f in glob.glob(f"{out}/pmc_attn2_{name}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            key = "attn_split2" if "attn_split2_kernel" in k else "attn_split (six products, x2 out)" if "attn_split_kernel" in k else "attn_f32" if "attn_f32" in k else None
            if not key:
                continue
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if name == "sq" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[key].items()}
    d = sum(dur[key]) / max(1, len(dur[key]))
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8            # (summed over the 8 XCDs)
    line = f"{key}: {d:8.1f} us  sclk {cyc / d / 1e3 if d else 0:.3f} GHz  mfma busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc) if cyc else 0:.3f}"
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        line += f"  wave cycles: active {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} wait-inst {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}"
    line += f"  valu insts {c.get('SQ_INSTS_VALU', 0):.3g}  lds insts {c.get('SQ_INSTS_LDS', 0):.3g}  lds conflict cycles {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g}"
    for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES"):
        if n in c:
            line += f"  {n} {c[n]:.3g}"
    line += f"  fetch {c.get('FETCH_SIZE', 0) * 2048 / 1e9:.3f} GB  write {c.get('WRITE_SIZE', 0) * 1024 / 1e9:.3f} GB"
    print(line)
PY
