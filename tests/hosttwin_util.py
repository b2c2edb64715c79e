"""Loader for the test-only host build of the device arithmetic (tests/hosttwin)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        override = os.environ.get("ECGPU_HOSTTWIN_LIB")          # a sanitizer build of the same sources (tools/hosttwin_sanitize.sh)
        if override:
            _LIB = ctypes.CDLL(override)
            return _LIB
        d = os.path.join(HERE, "hosttwin")
        subprocess.run(["make", "-s", "-C", d], check=True)
        _LIB = ctypes.CDLL(os.path.join(d, "libechosttwin.so"))
    return _LIB


def buf(b: bytes):
    return (ctypes.c_uint8 * len(b)).from_buffer_copy(b)


def outbuf(n: int):
    return (ctypes.c_uint8 * n)()
