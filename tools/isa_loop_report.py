#!/usr/bin/env python3
"""Where the scratch (spill / private-array) traffic of a kernel sits: disassembles one kernel of a built object and lists
its natural loops (backward branches) with their instruction mix.

    python tools/isa_loop_report.py ops_k256 k256_mul_fast_kernelILi32ELi4E [min_loop_len]

The compiler parks rare paths (the carry ripples of the field arithmetic) behind the hot code and jumps back, so most
"loops" in the list are those; the real loops are the long ones.  The line to read is the innermost long loop that
contains the table reads (global_load) and no scratch instruction.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co], check=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout


def main():
    obj = os.path.join(ROOT, "rustcrypto-elliptic-curves_amd", "build", sys.argv[1] + ".o")
    want = sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    text = disassemble(obj).split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <", l) and want in l)
    end = next((i for i in range(start + 1, len(text)) if re.match(r"^[0-9a-f]+ <", text[i])), len(text))
    addr_re = re.compile(r"//\s*([0-9A-F]+):")
    instrs = []
    for l in text[start + 1:end]:
        m = addr_re.search(l)
        if m:
            instrs.append((int(m.group(1), 16), l.split("//")[0].strip()))
    amap = {a: i for i, (a, _) in enumerate(instrs)}
    print("%s: %d instructions, %d scratch, %d v_mad_u64_u32" % (text[start].strip(), len(instrs), sum(t.startswith("scratch_") for _, t in instrs),
                                                                  sum(t.startswith("v_mad_u64_u32") for _, t in instrs)))
    loops = []
    for i, (a, t) in enumerate(instrs):
        m = re.match(r"s_c?branch\w*\s+(\d+)", t)
        if m:
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in amap and i - amap[tgt] + 1 >= min_len:
                loops.append((amap[tgt], i))
    print("%8s %8s %7s %8s %6s %7s %7s" % ("start", "end", "length", "scratch", "mad", "gload", "gstore"))
    for s, e in sorted(set(loops)):
        body = [t for _, t in instrs[s:e + 1]]
        print("%8d %8d %7d %8d %6d %7d %7d" % (s, e, e - s + 1, sum(t.startswith("scratch_") for t in body), sum(t.startswith("v_mad_u64_u32") for t in body),
                                            sum(t.startswith("global_load") for t in body), sum(t.startswith("global_store") for t in body)))
    # where the scratch instructions sit: clusters of instruction indices (gap > 400 starts a new cluster)
    idx = [i for i, (_, t) in enumerate(instrs) if t.startswith("scratch_")]
    clusters = []
    for i in idx:
        if clusters and i - clusters[-1][1] <= 400:
            clusters[-1][1] = i
            clusters[-1][2] += 1
        else:
            clusters.append([i, i, 1])
    print("scratch instruction clusters (first..last index: count): " + ", ".join("%d..%d: %d" % tuple(c) for c in clusters))


if __name__ == "__main__":
    main()
