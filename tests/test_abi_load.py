"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/ecgpu.h declares (no compute calls without a GPU), and the product has no CPU fallback."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

import ecgpu


def declared_symbols():
    with open(os.path.join(ROOT, "include", "ecgpu.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ecgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    if not os.path.exists(ecgpu.LIB_PATH):
        pytest.skip("libecgpu.so not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = ctypes.CDLL(ecgpu.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ecgpu.h but not exported"
    assert set(names) == set(ecgpu.EXPORTED_SYMBOLS)
    # and nothing else leaves the library (csrc/ecgpu.map): the boundary is the C ABI only
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", ecgpu.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == set(names)


def test_python_binding_declares_all_prototypes():
    if not os.path.exists(ecgpu.LIB_PATH):
        pytest.skip("libecgpu.so not built")
    lib = ecgpu.load_library()
    assert lib.ecgpu_version().startswith(b"ecgpu")
    assert [lib.ecgpu_field_bytes(c) for c in (0, 1, 2, 9)] == [32, 32, 48, 0]


def test_no_cpu_fallback_without_gpu():
    """Without a gfx950 device the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not os.path.exists(ecgpu.LIB_PATH):
        pytest.skip("libecgpu.so not built")
    with pytest.raises(ecgpu.EcgpuError):
        ecgpu.Context(0)


def test_product_does_not_reference_oracle():
    """oracle/ is test infrastructure: nothing under the package may import, link or open it."""
    pkg = os.path.join(ROOT, "rustcrypto-elliptic-curves_amd")
    for d, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".hip", ".h", ".cpp", "Makefile")):
                with open(os.path.join(d, fn), errors="ignore") as f:
                    for ln in f:
                        code = ln.split("//")[0].split("#")[0] if not fn.endswith(".py") else ln.split("#")[0]
                        assert "oracle/" not in code and "import oracle" not in code and "from oracle" not in code, (fn, ln)


def _example_binary():
    import subprocess
    d = os.path.join(ROOT, "examples")
    if not os.path.exists(ecgpu.LIB_PATH):
        pytest.skip("libecgpu.so not built")
    subprocess.run(["make", "-s", "-C", d], check=True)
    return os.path.join(d, "abi_example")


def test_c_caller_links_and_fails_loudly_without_gpu():
    """examples/abi_example.c is a plain-C caller of include/ecgpu.h (gcc, no HIP headers): it must compile and link
    against the library, and without a gfx950 device stop with the no-device status instead of computing anything."""
    import subprocess
    import torch
    exe = _example_binary()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (test_gpu_api.py runs the example)")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "no usable gfx950 device" in r.stderr


def test_host_chunk_schedule():
    """The chunking of host-buffer batches (csrc/host_pipe.hpp) is host logic and needs no device: sizes add up, grow from an eighth
    of a pass to a whole pass, end small (short drain), and leave no crumbs."""
    if not os.path.exists(ecgpu.LIB_PATH):
        pytest.skip("libecgpu.so not built")
    P = 1 << 23
    assert ecgpu.host_chunk_schedule(1 << 24, P) == [1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 20]
    assert ecgpu.host_chunk_schedule(1 << 25, P) == [1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 23, 1 << 23, 1 << 20]
    assert ecgpu.host_chunk_schedule((1 << 21) + 12345, P) == [1 << 20, (1 << 20) + 12345]
    assert ecgpu.host_chunk_schedule((1 << 22) + (1 << 20) + 4321, P) == [1 << 20, 1 << 21, (1 << 20) + 4321, 1 << 20]
    import random
    rng = random.Random(5)
    for _ in range(300):
        n = rng.randrange(1, 1 << 27)
        pass_units = 1 << rng.randrange(18, 26)
        p = min(max(pass_units, 1 << 20), 1 << 23)
        s = ecgpu.host_chunk_schedule(n, pass_units)
        assert sum(s) == n and all(x > 0 for x in s)
        assert all(x <= p + p // 16 for x in s), (n, pass_units, s)              # a chunk is at most a pass (plus an absorbed crumb)
        if len(s) > 1:
            assert s[-1] <= max(p // 8, 0) + p // 16 or s[-1] < 2 * (p // 8), (n, pass_units, s)   # short drain
            assert all(x >= p // 16 for x in s), (n, pass_units, s)             # no crumbs


def test_shard_range_arithmetic():
    """ecgpu_shard_range (the index ranges a device group hands its members): balanced, contiguous, disjoint,
    covering - including n < parts and n = 0 - and the same as the Python launcher's parallel.shard_range for equal shares."""
    if not os.path.exists(ecgpu.LIB_PATH):
        pytest.skip("libecgpu.so not built")
    import random
    rng = random.Random(11)
    cases = [(0, 1), (0, 8), (5, 8), (8, 8), (9, 8), ((1 << 26), 8), ((1 << 23) + 777, 2), ((1 << 22) + 12345, 2), (1, 64)]
    cases += [(rng.randrange(0, 1 << 40), rng.randrange(1, 65)) for _ in range(200)]
    for n, k in cases:
        nxt, sizes = 0, []
        for i in range(k):
            lo, cnt = ecgpu.shard_range(n, k, i)
            assert lo == nxt
            nxt = lo + cnt
            sizes.append(cnt)
        assert nxt == n and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        ecgpu.shard_range(10, 0, 0)
    with pytest.raises(ValueError):
        ecgpu.shard_range(10, 4, 4)
    from ecgpu import parallel
    for world in (1, 2, 4, 8):
        for rank in range(world):
            lo, hi = parallel.shard_range((1 << 24) + 5, rank, world)
            assert (lo, hi - lo) == ecgpu.shard_range((1 << 24) + 5, world, rank)
