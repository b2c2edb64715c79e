#!/usr/bin/env python3
"""VGPR / spill / scratch / occupancy of every kernel in the built objects (build/*.o), from the code-object metadata.

    python tools/kernel_resources.py [substring]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "rustcrypto-elliptic-curves_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = []
    for fn in sorted(os.listdir(BUILD)):
        if not fn.endswith(".o"):
            continue
        with tempfile.TemporaryDirectory() as td:
            co, fat = os.path.join(td, "dev.co"), os.path.join(td, "fat.bin")
            subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", os.path.join(BUILD, fn), fat], check=True)
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + fat, "--output=" + co], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            def g(key):
                m = re.search(r"\.%s:\s*(\S+)" % key, blk)
                return m.group(1) if m else "?"
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name)
            if pat in name:
                rows.append((fn, name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    print("%-12s %-90s %5s %6s %5s %8s %7s" % ("object", "kernel", "vgpr", "spill", "sgpr", "scratchB", "ldsB"))
    for r in rows:
        print("%-12s %-90s %5s %6s %5s %8s %7s" % r)


if __name__ == "__main__":
    main()
