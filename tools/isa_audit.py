#!/usr/bin/env python
"""Register / spill / scratch summary of the kernels in a hipcc -save-temps .s file (build.py uses the same parser).

    python tools/isa_audit.py fitclip_amd/csrc/build/gemm-hip-amdgcn-amd-amdhsa-gfx950.s [substring]
"""
import re
import sys


def kernels(path):
    text = open(path).read()
    md = text[text.index("amdhsa.kernels:"):]
    out = []
    for k in md.split("  - .agpr_count:")[1:]:
        def num(key):
            m = re.search(r"\.%s:\s+(\d+)" % key, k)
            return int(m.group(1)) if m else -1
        out.append({"name": re.search(r"\.name:\s+(\S+)", k).group(1), "agpr": int(k.split("\n")[0].strip()),
                    "vgpr": num("vgpr_count"), "sgpr": num("sgpr_count"), "vgpr_spill": num("vgpr_spill_count"),
                    "sgpr_spill": num("sgpr_spill_count"), "scratch": num("private_segment_fixed_size"),
                    "lds": num("group_segment_fixed_size")})
    return out


if __name__ == "__main__":
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels(sys.argv[1]):
        if sub in k["name"]:
            print(f"{k['name'][:120]:120s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} "
                  f"spill {k['vgpr_spill']:3d} scratch {k['scratch']:4d}")
