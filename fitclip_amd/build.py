"""Builds fitclip_amd/csrc/libfitclip_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m fitclip_amd.build [--force] [--save-temps]
    python -m fitclip_amd.build --debug          # tools/bin/libfitclip_hip_debug.so: NaN / Inf scan behind every tower call
    python -m fitclip_amd.build --host-asan      # CPU sanitizer target: csrc/bpe.cpp under ASan + UBSan (tests/test_sanitize.py)
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
REPO = CSRC.parent.parent
LIB = CSRC / "libfitclip_hip.so"
SOURCES = ["api.hip", "gemm.hip", "gemm_split3.hip", "gemm_split2.hip", "attention.hip", "attention_split.hip", "attention_split2.hip", "rowops.hip", "score.hip", "wgrad.hip", "attention_bwd.hip", "backward.hip",
           "train.hip", "bpe.cpp"]
HEADERS = [CSRC / "common.h", CSRC / "gemm_kernel.h", CSRC / "gemm_split3.h", CSRC / "gemm_split2.h", CSRC / "handle.h", CSRC / "unicode_ranges.inc",
           REPO / "include" / "fitclip_hip.h"]
ARCH = "gfx950"

# Kernels the bench lines name (bench.py: `roofline.kernel`, the time split) must not touch scratch memory: a spilled register in
# a kernel that lives at 256 VGPRs turns into scratch traffic inside the K loop.  Checked on the generated ISA after every
# compile (source file -> substrings of the mangled kernel names); a violation fails the build.
NO_SCRATCH_AUDIT = {
    "gemm.hip": ["gemm_pipelined_kernel", "gemm_kernel"],
    "gemm_split3.hip": ["gemm_split3_kernel"],
    "gemm_split2.hip": ["gemm_split2_kernel"],
    "attention_split.hip": ["attn_split_kernel"],
    "attention_split2.hip": ["attn_split2_kernel"],
    "attention.hip": ["attn_f32_blocks_kernel", "attn_f32_mfma_kernel", "attn_bf16_v2_kernel"],
    "rowops.hip": ["layernorm_kernel", "layernorm_pair_kernel"],
    "wgrad.hip": ["gemm_tn_kernel"],   # (the KD training step: 24 % of its kernel time)
}


def source_fingerprint() -> str:
    """Hash of every kernel / ABI source the library is built from.  Profiling artefacts (profiles/traffic_*.json) are
    stamped with it, and bench.py refuses a PMC pass that was made with other sources than the ones in the tree."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([*(CSRC / s for s in SOURCES), *CSRC.glob("*.h"), *CSRC.glob("*.inc"),
                        *(REPO / "include").glob("*.h")]):
        h.update(path.name.encode())
        h.update(path.read_bytes())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


# attention.hip loads the Q fragments with inline-asm `global_load_dwordx4` (so that hipcc does not drain vmcnt to 0
# while LDS-DMA pieces are in flight) and hands the registers to the compiler at a counted `s_waitcnt vmcnt`.  Nothing
# may read those registers in between -- hipcc cannot know that, so the generated ISA is checked after every compile.
ASYNC_LOAD_AUDIT = {"attention.hip": ["attn_bf16_v2_kernel"]}


def audit_async_loads(asm_path: Path, kernel: str) -> int:
    """Raises if, in any instantiation of `kernel`, an instruction between the asm loads and the first s_waitcnt vmcnt
    names a register those loads write.  Returns the number of instantiations checked."""
    import re
    text = asm_path.read_text()
    checked = 0
    for m in re.finditer(r"^(\S*%s\S*):\s*;" % re.escape(kernel), text, flags=re.M):
        body = text[m.end():]
        body = body[:body.find("s_endpgm")]
        loaded, waiting = set(), False
        for line in body.splitlines():
            code = line.split(";")[0]
            ld = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off", code)
            if ld:
                loaded.update(range(int(ld.group(1)), int(ld.group(2)) + 1))
                waiting = True
                continue
            if not waiting:
                continue
            if re.search(r"s_waitcnt vmcnt\(\d+\)", code):
                break
            used = set(int(x) for x in re.findall(r"\bv(\d+)\b", code))
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
                used.update(range(int(a), int(b) + 1))
            if used & loaded:
                raise RuntimeError(f"{asm_path.name}: {m.group(1)} touches a register of an asynchronous load before "
                                   f"its s_waitcnt: {line.strip()}")
        if not loaded:
            raise RuntimeError(f"{asm_path.name}: {m.group(1)}: no asynchronous loads found (audit out of date?)")
        checked += 1
    if checked == 0:
        raise RuntimeError(f"{asm_path.name}: kernel {kernel} not found for the asynchronous-load audit")
    return checked


def kernel_resources(asm_path: Path):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] from the `amdhsa.kernels` metadata of a hipcc .s file."""
    import re
    text = asm_path.read_text()
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


def audit_no_scratch(asm_path: Path, substrings) -> int:
    """Raises if a kernel whose name contains one of `substrings` spills a VGPR or owns scratch memory; returns the number checked."""
    checked = 0
    for k in kernel_resources(asm_path):
        if any(sub in k["name"] for sub in substrings):
            checked += 1
            if k["vgpr_spill"] != 0 or k["scratch"] != 0:
                raise RuntimeError(f"{asm_path.name}: {k['name']} spills {k['vgpr_spill']} VGPR(s), "
                                   f"{k['scratch']} bytes of scratch per lane (.vgpr_spill_count / .private_segment_fixed_size must be 0)")
    if checked == 0:
        raise RuntimeError(f"{asm_path.name}: no kernel matches {substrings} (audit out of date?)")
    return checked


# gemm_split2.h issues its LDS-DMA pieces as inline asm (`s_mov_b32 m0, ...; buffer_load_dwordx4 ... offen lds`): hipcc pads no
# hazard whose consumer sits inside an asm string (cdna_hip_programming.md section 5.7 item 2).  The one that can bite here: a VALU
# write of an SGPR (v_readlane_b32 reloading a spilled scalar, v_readfirstlane_b32) needs 5 wait states before a VMEM instruction
# reads that SGPR as descriptor or offset.  SALU writes are interlocked.  Checked on the generated ISA after every compile.
ASM_DMA_AUDIT = {"gemm_split2.hip": ["gemm_split2_kernel"]}


def audit_asm_dma_hazards(asm_path: Path, substrings) -> int:
    """Raises if an LDS-DMA `buffer_load ... lds` of a kernel whose name contains one of `substrings` reads an SGPR whose latest
    writer, fewer than 5 wait states upstream (straight-line scan; s_nop N counts N + 1), is a VALU instruction.  Returns the number
    of DMA instructions checked."""
    import re
    text = asm_path.read_text()
    checked = 0

    def sregs(tok):
        out = set()
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", tok):
            out.update(range(int(a), int(b) + 1))
        out.update(int(a) for a in re.findall(r"\bs(\d+)\b", tok))
        return out

    for m in re.finditer(r"^(\S+):\s*;\s*@", text, flags=re.M):
        if not any(sub in m.group(1) for sub in substrings):
            continue
        body = text[m.end():]
        body = body[:body.find("s_endpgm")]
        lines = [l.split(";")[0].strip() for l in body.splitlines()]
        lines = [l for l in lines if l and not l.endswith(":") and not l.startswith(".")]
        for n, c in enumerate(lines):
            if not (c.startswith("buffer_load_dword") and c.endswith(" lds")):
                continue
            checked += 1
            need, states, k = sregs(c), 0, n - 1
            while k >= 0 and states < 5 and need:
                p = lines[k]
                k -= 1
                dst = re.match(r"(\S+)\s+(s\[\d+:\d+\]|s\d+)\b", p)
                if dst:
                    written = sregs(dst.group(2)) & need
                    if written:
                        if p.startswith("v_"):
                            raise RuntimeError(f"{asm_path.name}: {m.group(1)}: `{p}` writes an SGPR {states} wait state(s) before `{c}` "
                                               f"reads it inside an asm statement (a VALU-written SGPR needs 5 before a VMEM read)")
                        need -= written   # an SALU write: interlocked, and it hides older writers
                nop = re.match(r"s_nop\s+(\d+)", p)
                states += int(nop.group(1)) + 1 if nop else 1
    if checked == 0:
        raise RuntimeError(f"{asm_path.name}: no LDS-DMA instruction found in {substrings} (audit out of date?)")
    return checked


LAB_LIB = REPO / "tools" / "bin" / "libfitclip_hip_lab.so"
DEBUG_LIB = REPO / "tools" / "bin" / "libfitclip_hip_debug.so"


def build(force: bool = False, save_temps: bool = False, verbose: bool = True, lab: bool = False, debug: bool = False) -> Path:
    """`lab=True` builds tools/bin/libfitclip_hip_lab.so with -DFITCLIP_LAB: the same sources plus the A/B switches the lab
    scripts under tools/ read from the environment (select it with FITCLIP_HIP_LIB=...).  The product library never defines it.
    `debug=True` builds tools/bin/libfitclip_hip_debug.so with -DFITCLIP_DEBUG: every tower call scans its output for NaN / Inf
    (a device-side count, one host synchronisation per call) and fails with the call's name."""
    hipcc = _hipcc()
    objdir = CSRC / ("build_lab" if lab else "build_debug" if debug else "build")
    objdir.mkdir(exist_ok=True)
    flags = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
             "-fvisibility=hidden", "-DFITCLIP_BUILD"] + (["-DFITCLIP_LAB"] if lab else []) + (["-DFITCLIP_DEBUG"] if debug else [])
    lib_path = LAB_LIB if lab else DEBUG_LIB if debug else LIB
    lib_path.parent.mkdir(exist_ok=True)

    def compile_one(src: str) -> Path:
        obj = objdir / (src.replace(".hip", ".o").replace(".cpp", ".o"))
        audited = src in ASYNC_LOAD_AUDIT or src in NO_SCRATCH_AUDIT or src in ASM_DMA_AUDIT
        asm = objdir / (src.replace(".hip", "") + f"-hip-amdgcn-amd-amdhsa-{ARCH}.s")
        # an audited source is compiled with -save-temps and its ISA checked on EVERY build: an up-to-date object whose .s file is
        # missing or older than the source (built before the source joined an audit table) is recompiled, never linked unaudited
        if force or _stale(obj, [CSRC / src, *HEADERS]) or (audited and _stale(asm, [CSRC / src, *HEADERS])):
            cmd = [hipcc, *flags, "-c", str(CSRC / src), "-o", str(obj)]
            if save_temps or audited:
                cmd += ["-save-temps=obj"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=str(objdir))
        if audited:
            try:
                for kernel in ASYNC_LOAD_AUDIT.get(src, []):
                    audit_async_loads(asm, kernel)
                if src in NO_SCRATCH_AUDIT:
                    n = audit_no_scratch(asm, NO_SCRATCH_AUDIT[src])
                    if verbose:
                        print(f"{src}: {n} kernel instantiation(s) audited: no VGPR spills, no scratch", flush=True)
                if src in ASM_DMA_AUDIT:
                    n = audit_asm_dma_hazards(asm, ASM_DMA_AUDIT[src])
                    if verbose:
                        print(f"{src}: {n} inline-asm LDS-DMA instruction(s) audited: no VALU-written SGPR within 5 wait states", flush=True)
            except Exception:
                obj.unlink(missing_ok=True)  # never link an object that failed an audit
                raise
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    if force or _stale(lib_path, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib_path), *map(str, objs), "-lz"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib_path


HOST_ASAN = CSRC / "build" / "bpe_asan"


def build_host_asan(verbose: bool = True) -> Path:
    """CPU sanitizer target (never the GPU build): the host-only part of the library - csrc/bpe.cpp, which parses external
    gzip files and arbitrary UTF-8 - with g++ -fsanitize=address,undefined and the driver tests/host/bpe_sanitize_driver.cpp.
    tests/test_sanitize.py runs the tokenizer fixtures and a seeded corpus of hostile inputs through it."""
    driver = REPO / "tests" / "host" / "bpe_sanitize_driver.cpp"
    HOST_ASAN.parent.mkdir(exist_ok=True)
    if _stale(HOST_ASAN, [CSRC / "bpe.cpp", CSRC / "unicode_ranges.inc", driver, REPO / "include" / "fitclip_hip.h"]):
        cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
               "-fno-sanitize-recover=all", "-Wall", str(CSRC / "bpe.cpp"), str(driver), "-o", str(HOST_ASAN), "-lz"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return HOST_ASAN


if __name__ == "__main__":
    if "--host-asan" in sys.argv:
        print(build_host_asan())
    else:
        print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, lab="--lab" in sys.argv, debug="--debug" in sys.argv))
