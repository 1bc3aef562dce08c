#!/bin/bash
# Lab: PMC passes over tools/bin/x2k_lab (the two MFMA shapes of the two-plane fp16 GEMM): matrix-pipe busy fraction and shader clock
# (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE), wave-cycle split, LDS bank conflicts, L1 pending-miss stalls, and memory-side traffic
# (FETCH_SIZE / WRITE_SIZE, separate passes; gfx950: FETCH_SIZE x 2, KiB) per launch.  Few dispatches per pass (3 reps, 1 round).
#   tools/x2k_pmc.sh [M] [variant filter] [tag]        (run from the repo root on the GPU box)
set -e
M=${1:-151296}; filt=${2:-spread}; tag=${3:-x2k}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  name=$1; shift
  rm -rf "$out/pmc_${tag}_$name"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_${tag}_$name" -o p -- "$repo/tools/bin/x2k_lab" $M 3 1 1 1 "$filt" > "$out/pmc_${tag}_$name.log" 2> "$out/pmc_${tag}_$name.err" || { tail -5 "$out/pmc_${tag}_$name.err"; return 1; }
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcp TCP_PENDING_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES || echo "tcp pass failed"
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for name in ("sq", "fetch", "write", "tcp"):
    for f in glob.glob(f"{out}/pmc_{tag}_{name}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.search(r"gemm_split2(_m32)?_kernel<(\d+), (\d+), (\d+)", k) or re.search(r"gemm_split2(_m32)?_kernelILi(\d+)ELi(\d+)ELi(\d+)", k)
            if not m:
                continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            # (epilogue 8 serves out_proj and c_proj: told apart by duration)
            key = ("m32" if m.group(1) else "k32", int(m.group(2)), int(m.group(3)), int(m.group(4)), "long" if d > 900 else "short")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if name == "sq" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[key].append(d)
for key in sorted(agg):
    c = {k: sum(v) / len(v) for k, v in agg[key].items()}
    d = sum(dur[key]) / max(1, len(dur[key]))
    line = f"{key[0]} epi {key[1]:2d} abl {key[2]} spread {key[3]} {key[4]:5s}: {d:8.1f} us"
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        wc = max(1.0, c.get("SQ_WAVE_CYCLES", 1))
        line += f"  sclk {cyc / (d * 1e3):.3f} GHz  mfma busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc:.3f}  lds conflicts {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g}"
        line += f"  wave cycles: active {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} wait-inst {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} wait {c.get('SQ_WAIT_ANY', 0) / wc:.2f}  valu insts {c.get('SQ_INSTS_VALU', 0):.3g}"
    if "FETCH_SIZE" in c:
        line += f"  fetch {2 * 1024 * c['FETCH_SIZE'] / 1e9:.3f} GB"
    if "WRITE_SIZE" in c:
        line += f"  write {1024 * c['WRITE_SIZE'] / 1e9:.3f} GB"
    if "TCP_PENDING_STALL_CYCLES" in c:
        line += f"  tcp pending stall {c['TCP_PENDING_STALL_CYCLES']:.3g}  ta stalled by tc {c.get('TA_ADDR_STALLED_BY_TC_CYCLES', 0):.3g}"
    print(line)
PY
rm -rf "$out"/pmc_${tag}_sq "$out"/pmc_${tag}_fetch "$out"/pmc_${tag}_write "$out"/pmc_${tag}_tcp
