#!/bin/bash
# Lab: PMC passes over tools/bin/split2_lab (the two-plane fp16 GEMM next to the shipped three-plane bf16 kernel): matrix-pipe
# busy fraction and shader clock (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE), LDS bank conflicts, L1 pending-miss stalls, and
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes; gfx950: FETCH_SIZE x 2, KiB) per launch.
#   tools/split2_pmc.sh [M] [variant filter]        (run from the repo root on the GPU box)
set -e
M=${1:-151296}; filt=${2:-spread3}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  name=$1; shift
  rm -rf "$out/pmc_s2_$name"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_s2_$name" -o p -- "$repo/tools/bin/split2_lab" $M 3 1 1 1 "$filt" > "$out/pmc_s2_$name.log" 2> "$out/pmc_s2_$name.err" || { tail -5 "$out/pmc_s2_$name.err"; return 1; }
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcp TCP_PENDING_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES || echo "tcp pass failed"
python3 - "$out" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for name in ("sq", "fetch", "write", "tcp"):
    for f in glob.glob(f"{out}/pmc_s2_{name}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_split" not in k:
                continue
            m = re.search(r"gemm_split(\d)_kernel<(\d+)", k) or re.search(r"gemm_split(\d)_kernelILi(\d+)", k)
            # shapes share instantiations: key by (kernel, epilogue, grid work = duration bucket is not needed: one shape per epilogue
            # except epilogue 8 = out_proj and c_proj, told apart by duration)
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            key = (f"split{m.group(1)}", int(m.group(2)), "long" if d > 1000 else "short")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if name == "sq" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[key].append(d)
for key in sorted(agg):
    c = {k: sum(v) / len(v) for k, v in agg[key].items()}
    d = sum(dur[key]) / max(1, len(dur[key]))
    line = f"{key[0]} epilogue {key[1]:2d} {key[2]:5s}: {d:8.1f} us"
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        line += f"  sclk {cyc / (d * 1e3):.3f} GHz  mfma busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc:.3f}  lds conflicts {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g}"
        line += f"  wait/active inst {c.get('SQ_WAIT_INST_ANY', 0) / max(1.0, c.get('SQ_ACTIVE_INST_ANY', 1)):.2f}"
    if "FETCH_SIZE" in c:
        line += f"  fetch {2 * 1024 * c['FETCH_SIZE'] / 1e9:.3f} GB"
    if "WRITE_SIZE" in c:
        line += f"  write {1024 * c['WRITE_SIZE'] / 1e9:.3f} GB"
    if "TCP_PENDING_STALL_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        line += f"  tcp pending stall {c['TCP_PENDING_STALL_CYCLES'] / 256.0 / (c['GRBM_GUI_ACTIVE'] / 8.0):.3f}  ta stalled by tc {c.get('TA_ADDR_STALLED_BY_TC_CYCLES', 0) / 256.0 / (c['GRBM_GUI_ACTIVE'] / 8.0):.3f}"
    print(line)
PY
rm -rf "$out"/pmc_s2_sq "$out"/pmc_s2_fetch "$out"/pmc_s2_write "$out"/pmc_s2_tcp
