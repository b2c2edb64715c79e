"""The Rust side of the boundary is source only (no rustc in the image).  What CAN be checked without a compiler: its
`extern "C"` block declares exactly the functions of include/ecgpu.h, each with the header's number of parameters, and
the constants it mirrors have the header's values."""
import os
import re

from conftest import ROOT


def _header_functions():
    text = open(os.path.join(ROOT, "include", "ecgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(ecgpu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out, text


def _rust_functions():
    text = open(os.path.join(ROOT, "rustcrypto-elliptic-curves_amd", "rust", "ecgpu-sys", "src", "lib.rs")).read()
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (ecgpu_[a-z0-9_]+)\s*\(([^)]*)\)", block, flags=re.S):
        args = m.group(2).strip().rstrip(",")
        out[m.group(1)] = 0 if not args else args.count(":")
    return out, text


def test_extern_block_matches_header():
    h, htext = _header_functions()
    r, rtext = _rust_functions()
    assert set(h) == set(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    for name in h:
        assert h[name] == r[name], (name, h[name], r[name])
    # enum values mirrored as constants
    for name, val in re.findall(r"\b(ECGPU_[A-Z0-9_]+)\s*=\s*(-?\d+)u?", htext):
        m = re.search(r"pub const %s: \w+ = (-?\d+);" % name, rtext)
        assert m and int(m.group(1)) == int(val), name
