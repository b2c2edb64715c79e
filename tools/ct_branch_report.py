#!/usr/bin/env python3
"""Every conditional branch of the kernels in one built object, with the instruction that produced its condition.

    python tools/ct_branch_report.py ops_k256_ct [kernel-substring]

For the constant-time translation unit (csrc/ops_k256_ct.hip, compiled with ECGPU_K256_BRANCHFREE) the claim to check is: no
conditional branch depends on a value derived from a secret.  On this ISA a branch reads SCC (set by a scalar compare: wave-uniform
values - loop counters, kernel arguments), VCC or EXEC (set by vector compares: per-lane values).  The report classifies each
s_cbranch by the instruction that last wrote the register it tests and, for vector compares, prints the compare and a few instructions
of context, so that a reader can see what is compared: the loop counters and `index < n` of the public batch size are expected; a
compare of a field limb or a carry would be the data-dependent branch the throughput kernels have (fe_k256.hpp ECGPU_K256_RARE).
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_loop_report import disassemble, ROOT


def kernels(text):
    cur, out = None, {}
    for l in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None and "//" in l:
            out[cur].append(l.split("//")[0].strip())
    return out


def main():
    obj = os.path.join(ROOT, "rustcrypto-elliptic-curves_amd", "build", sys.argv[1] + ".o")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    import subprocess
    for name, ins in kernels(disassemble(obj)).items():
        if pat not in name or not ins:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        br = [i for i, t in enumerate(ins) if t.startswith("s_cbranch")]
        kinds = {}
        detail = []
        for i in br:
            t = ins[i]
            reg = "scc" if "scc" in t else "vcc" if "vcc" in t else "exec" if "exec" in t else "?"
            # last writer of the tested register
            src = None
            for j in range(i - 1, max(-1, i - 400), -1):
                u = ins[j]
                if reg == "scc" and re.match(r"s_(cmp|cmpk|bitcmp|and|or|xor|andn2|orn2|add|sub|lshl|lshr|ashr|bfe|cselect|not|ff1|flbit|min|max|abs|mul)", u):
                    src = (j, u); break
                if reg == "vcc" and re.search(r"\bvcc\b", u.split(" ", 1)[1] if " " in u else "") and not u.startswith("s_cbranch"):
                    src = (j, u); break
                if reg == "exec" and (re.match(r"s_(and|or|xor|andn2|mov|cmov)\w*\s+exec", u) or re.match(r"v_cmpx", u) or re.match(r"s_\w+_saveexec", u)):
                    src = (j, u); break
            kind = "%s <- %s" % (reg, (src[1].split()[0] if src else "?"))
            kinds[kind] = kinds.get(kind, 0) + 1
            if reg != "scc" or (src and not src[1].startswith("s_cmp")):
                detail.append((i, t, src))
        print("== %s\n   %d instructions, %d conditional branches: %s" % (dem, len(ins), len(br), ", ".join("%d x [%s]" % (v, k) for k, v in sorted(kinds.items()))))
        for i, t, src in detail:
            print("   @%d %s   condition from @%s %s" % (i, t, src[0] if src else "?", src[1] if src else "?"))
            if src:
                # the vector compare that fed a VCC / EXEC mask: show the nearest v_cmp before the writer
                for j in range(src[0], max(-1, src[0] - 60), -1):
                    if ins[j].startswith("v_cmp"):
                        print("        nearest vector compare @%d: %s" % (j, ins[j]))
                        break


if __name__ == "__main__":
    main()
