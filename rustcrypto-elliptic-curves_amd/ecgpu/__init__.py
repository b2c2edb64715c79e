"""Host-side mirror of the reference's trait surface over the libecgpu C ABI (include/ecgpu.h).

The reference is a Rust workspace; no Rust toolchain exists in this image, so the host side
above the C ABI that tests and bench.py drive is this thin ctypes binding.  Names follow the
reference: `Curve.mul_by_generator` (MulByGenerator), `Curve.mul` (`&P * &k`), `Curve.lincomb`
(LinearCombination), `Curve.add / add_mixed / double` (ProjectivePoint), `Curve.batch_normalize`
(BatchNormalize), `Curve.field_*` (FieldElement).  The Rust shim that a maintainer would add is
in ../rust/ (source only) and described in INTEGRATION.md.

There is no CPU fallback: if libecgpu.so is missing or no gfx950 device is present, every entry
point raises `EcgpuError`.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import numpy as np

try:  # one HIP runtime per process: let torch's bundled libamdhip64 load first so that libecgpu.so
    import torch  # noqa: F401  binds to the same copy (device pointers / streams are then shared)
except ImportError:  # torch is optional plumbing; the C ABI itself needs only the ROCm runtime
    torch = None

HOST, DEVICE = 0, 1
AFFINE, PROJECTIVE = 0, 1
EXACT_REFERENCE = 1
ECDSA_LOW_S = 2
PUBLIC_SCALARS = 4
SECRET_SCALARS = 8
K256, P256, P384 = 0, 1, 2
CURVE_IDS = {"k256": K256, "p256": P256, "p384": P384}
FIELD_BYTES = {K256: 32, P256: 32, P384: 48}
FE_MUL, FE_SQR, FE_ADD, FE_SUB, FE_NEG, FE_INV, FE_SQRT = range(7)
# ecgpu_option (per-context tuning / test knobs, include/ecgpu.h)
OPT_FB_WINDOW, OPT_FB_MAX_WINDOW, OPT_MSM_WINDOW_BITS, OPT_MSM_SLAB_TERMS, OPT_MSM_SMALL_PATH, OPT_MSM_ROUNDS, OPT_K256_WAVES, OPT_FB_MEMORY_BUDGET, OPT_LINCOMB_TERM_BY_TERM = range(9)

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ECGPU_LIB: another build of the library (A/B measurements of compile-time switches); default: the in-tree build
LIB_PATH = os.environ.get("ECGPU_LIB") or os.path.join(_PKG_ROOT, "lib", "libecgpu.so")


def host_chunk_schedule(n: int, pass_units: int) -> list:
    """chunk sizes of a host-buffer batch (ecgpu_host_chunk_schedule; no device needed)"""
    lib = load_library()
    cnt = lib.ecgpu_host_chunk_schedule(n, pass_units, None, 0)
    arr = (ctypes.c_size_t * max(cnt, 1))()
    lib.ecgpu_host_chunk_schedule(n, pass_units, arr, cnt)
    return list(arr[:cnt])


def shard_range(n: int, parts: int, index: int) -> tuple:
    """(first, count) of part `index` of n elements cut into `parts` balanced contiguous ranges (ecgpu_shard_range; no device needed)"""
    lib = load_library()
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    if lib.ecgpu_shard_range(n, parts, index, ctypes.byref(a), ctypes.byref(b)) != 0:
        raise ValueError("shard_range(%d, %d, %d)" % (n, parts, index))
    return a.value, b.value


class EcgpuError(RuntimeError):
    pass


_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen libecgpu.so and declare every prototype of include/ecgpu.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise EcgpuError(f"{p} not found: build it with `make -C rustcrypto-elliptic-curves_amd` (no CPU fallback exists)")
    lib = ctypes.CDLL(p)
    vp, sz, i, u8p = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p
    lib.ecgpu_create.argtypes = [ctypes.POINTER(vp), i]
    lib.ecgpu_destroy.argtypes = [vp]
    lib.ecgpu_destroy.restype = None
    lib.ecgpu_set_stream.argtypes = [vp, vp]
    lib.ecgpu_use_own_stream.argtypes = [vp]
    lib.ecgpu_last_error_copy.argtypes = [vp, ctypes.c_char_p, sz]
    lib.ecgpu_set_option.argtypes = [vp, i, ctypes.c_int64]
    lib.ecgpu_get_option.argtypes = [vp, i, ctypes.POINTER(ctypes.c_int64)]
    lib.ecgpu_fb_table_bytes.argtypes = [vp, i, ctypes.POINTER(sz), ctypes.POINTER(i)]
    lib.ecgpu_ecdh_batch.argtypes = [vp, i, u8p, u8p, u8p, u8p, sz, i]
    lib.ecgpu_sec1_encode_batch.argtypes = [vp, i, u8p, i, i, u8p, sz, i]
    lib.ecgpu_sec1_decode_batch.argtypes = [vp, i, u8p, sz, u8p, u8p, sz, i]
    lib.ecgpu_synchronize.argtypes = [vp]
    lib.ecgpu_last_error.argtypes = [vp]
    lib.ecgpu_last_error.restype = ctypes.c_char_p
    lib.ecgpu_version.restype = ctypes.c_char_p
    lib.ecgpu_field_bytes.argtypes = [i]
    lib.ecgpu_field_bytes.restype = sz
    lib.ecgpu_host_alloc.argtypes = [vp, sz, ctypes.POINTER(vp)]
    lib.ecgpu_host_free.argtypes = [vp, vp]
    lib.ecgpu_debug_workspace.argtypes = [vp, i, vp, sz, ctypes.POINTER(sz)]
    lib.ecgpu_host_chunk_schedule.argtypes = [sz, sz, ctypes.POINTER(sz), sz]
    pp = ctypes.POINTER(vp)
    lib.ecgpu_group_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i), i, ctypes.c_uint]
    lib.ecgpu_group_destroy.argtypes = [vp]
    lib.ecgpu_group_destroy.restype = None
    lib.ecgpu_group_size.argtypes = [vp]
    lib.ecgpu_group_context.argtypes = [vp, i]
    lib.ecgpu_group_context.restype = vp
    lib.ecgpu_group_last_error.argtypes = [vp]
    lib.ecgpu_group_last_error.restype = ctypes.c_char_p
    lib.ecgpu_group_gather_path.argtypes = [vp]
    lib.ecgpu_group_gather_path.restype = ctypes.c_char_p
    lib.ecgpu_group_synchronize.argtypes = [vp]
    lib.ecgpu_shard_range.argtypes = [sz, i, i, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.ecgpu_group_mul_batch.argtypes = [vp, i, u8p, u8p, i, u8p, i, u8p, sz, ctypes.c_uint]
    lib.ecgpu_group_lincomb_batch.argtypes = [vp, i, u8p, u8p, i, sz, u8p, i, u8p, sz, ctypes.c_uint]
    lib.ecgpu_group_lincomb_sharded.argtypes = [vp, i, pp, pp, i, sz, pp, i, pp, ctypes.POINTER(sz), ctypes.c_uint]
    lib.ecgpu_group_msm.argtypes = [vp, i, u8p, u8p, i, sz, u8p, i]
    lib.ecgpu_group_msm_sharded.argtypes = [vp, i, pp, pp, i, ctypes.POINTER(sz), u8p, i]
    lib.ecgpu_timer_start.argtypes = [vp]
    lib.ecgpu_timer_stop.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.ecgpu_field_op_batch.argtypes = [vp, i, i, u8p, u8p, u8p, sz, i]
    lib.ecgpu_point_add_batch.argtypes = [vp, i, u8p, u8p, u8p, sz, i]
    lib.ecgpu_point_add_mixed_batch.argtypes = [vp, i, u8p, u8p, u8p, sz, i]
    lib.ecgpu_point_double_batch.argtypes = [vp, i, u8p, u8p, sz, i]
    lib.ecgpu_batch_normalize.argtypes = [vp, i, u8p, u8p, u8p, sz, i]
    lib.ecgpu_point_eq_batch.argtypes = [vp, i, u8p, u8p, u8p, sz, i]
    lib.ecgpu_mul_batch_checked.argtypes = [vp, i, u8p, u8p, i, u8p, i, u8p, u8p, sz, i, ctypes.c_uint]
    lib.ecgpu_lincomb_batch_checked.argtypes = [vp, i, u8p, u8p, i, sz, u8p, i, u8p, u8p, sz, i, ctypes.c_uint]
    lib.ecgpu_mul_batch.argtypes = [vp, i, u8p, u8p, i, u8p, i, u8p, sz, i, ctypes.c_uint]
    lib.ecgpu_lincomb_batch.argtypes = [vp, i, u8p, u8p, i, sz, u8p, i, u8p, sz, i, ctypes.c_uint]
    lib.ecgpu_msm.argtypes = [vp, i, u8p, u8p, i, sz, u8p, i, i]
    lib.ecgpu_validate_scalars.argtypes = [vp, i, u8p, u8p, sz, i]
    lib.ecgpu_validate_points.argtypes = [vp, i, u8p, u8p, sz, i]
    lib.ecgpu_decompress_batch.argtypes = [vp, i, u8p, u8p, u8p, u8p, sz, i]
    lib.ecgpu_to_bytes_batch.argtypes = [vp, i, u8p, i, u8p, sz, i]
    lib.ecgpu_from_bytes_batch.argtypes = [vp, i, u8p, u8p, u8p, sz, i]
    lib.ecgpu_ecdsa_verify_batch.argtypes = [vp, i, u8p, u8p, u8p, u8p, sz, i, ctypes.c_uint]
    lib.ecgpu_map_to_curve_batch.argtypes = [vp, i, u8p, i, u8p, u8p, sz, i]
    lib.ecgpu_ecdsa_recover_batch.argtypes = [vp, i, u8p, u8p, u8p, u8p, u8p, sz, i, ctypes.c_uint]
    lib.ecgpu_schnorr_verify_batch.argtypes = [vp, i, u8p, u8p, u8p, u8p, sz, i]
    lib.ecgpu_ecdsa_sign_batch.argtypes = [vp, i, u8p, u8p, u8p, u8p, u8p, u8p, sz, i, ctypes.c_uint]
    lib.ecgpu_synth_scalars.argtypes = [vp, i, ctypes.c_uint64, ctypes.c_uint64, u8p, sz]
    lib.ecgpu_synth_points.argtypes = [vp, i, ctypes.c_uint64, ctypes.c_uint64, u8p, sz]
    for name in ("ecgpu_create", "ecgpu_set_stream", "ecgpu_synchronize", "ecgpu_timer_start", "ecgpu_timer_stop",
                 "ecgpu_field_op_batch", "ecgpu_point_add_batch", "ecgpu_point_add_mixed_batch",
                 "ecgpu_point_double_batch", "ecgpu_batch_normalize", "ecgpu_mul_batch", "ecgpu_lincomb_batch",
                 "ecgpu_msm", "ecgpu_validate_scalars", "ecgpu_validate_points", "ecgpu_decompress_batch",
                 "ecgpu_synth_scalars", "ecgpu_synth_points", "ecgpu_point_eq_batch", "ecgpu_mul_batch_checked",
                 "ecgpu_lincomb_batch_checked", "ecgpu_ecdsa_verify_batch", "ecgpu_ecdsa_sign_batch", "ecgpu_to_bytes_batch",
                 "ecgpu_from_bytes_batch", "ecgpu_host_alloc", "ecgpu_host_free", "ecgpu_schnorr_verify_batch",
                 "ecgpu_ecdsa_recover_batch", "ecgpu_map_to_curve_batch", "ecgpu_use_own_stream", "ecgpu_last_error_copy",
                 "ecgpu_set_option", "ecgpu_get_option", "ecgpu_fb_table_bytes", "ecgpu_sec1_encode_batch", "ecgpu_sec1_decode_batch",
                 "ecgpu_ecdh_batch", "ecgpu_debug_workspace", "ecgpu_host_chunk_schedule", "ecgpu_group_create", "ecgpu_group_size",
                 "ecgpu_group_synchronize", "ecgpu_shard_range", "ecgpu_group_mul_batch", "ecgpu_group_lincomb_batch", "ecgpu_group_lincomb_sharded",
                 "ecgpu_group_msm", "ecgpu_group_msm_sharded"):
        getattr(lib, name).restype = ctypes.c_int
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "ecgpu_create", "ecgpu_destroy", "ecgpu_set_stream", "ecgpu_synchronize", "ecgpu_last_error", "ecgpu_version",
    "ecgpu_field_bytes", "ecgpu_timer_start", "ecgpu_timer_stop", "ecgpu_field_op_batch", "ecgpu_point_add_batch",
    "ecgpu_point_add_mixed_batch", "ecgpu_point_double_batch", "ecgpu_batch_normalize", "ecgpu_mul_batch",
    "ecgpu_lincomb_batch", "ecgpu_msm", "ecgpu_validate_scalars", "ecgpu_validate_points", "ecgpu_decompress_batch",
    "ecgpu_synth_scalars", "ecgpu_synth_points", "ecgpu_ecdsa_verify_batch", "ecgpu_ecdsa_sign_batch",
    "ecgpu_to_bytes_batch", "ecgpu_from_bytes_batch", "ecgpu_host_alloc", "ecgpu_host_free", "ecgpu_schnorr_verify_batch", "ecgpu_ecdsa_recover_batch", "ecgpu_map_to_curve_batch",
    "ecgpu_point_eq_batch", "ecgpu_mul_batch_checked", "ecgpu_lincomb_batch_checked",
    "ecgpu_use_own_stream", "ecgpu_last_error_copy", "ecgpu_set_option", "ecgpu_get_option", "ecgpu_fb_table_bytes",
    "ecgpu_sec1_encode_batch", "ecgpu_sec1_decode_batch", "ecgpu_ecdh_batch", "ecgpu_debug_workspace", "ecgpu_host_chunk_schedule",
    "ecgpu_group_create", "ecgpu_group_destroy", "ecgpu_group_size", "ecgpu_group_context", "ecgpu_group_last_error", "ecgpu_group_gather_path",
    "ecgpu_group_synchronize", "ecgpu_shard_range", "ecgpu_group_mul_batch", "ecgpu_group_lincomb_batch", "ecgpu_group_lincomb_sharded",
    "ecgpu_group_msm", "ecgpu_group_msm_sharded",
)


def _ptr(x):
    """Host numpy array / bytes -> (pointer, keepalive); torch CUDA tensor or int -> device pointer."""
    if x is None:
        return None, None
    if isinstance(x, int):
        return ctypes.c_void_p(x), None
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("the C ABI takes dense buffers: pass a C-contiguous array")
        return ctypes.c_void_p(x.ctypes.data), x
    if hasattr(x, "data_ptr"):  # torch tensor
        if not x.is_contiguous():
            raise ValueError("the C ABI takes dense buffers: pass a contiguous tensor")
        return ctypes.c_void_p(x.data_ptr()), x
    raise TypeError(type(x))


class Context:
    """One context per device (include/ecgpu.h: context section)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.handle = ctypes.c_void_p()
        rc = self.lib.ecgpu_create(ctypes.byref(self.handle), device)
        if rc != 0:
            raise EcgpuError(f"ecgpu_create(device={device}) failed with {rc} (no usable gfx950 GPU? there is no CPU fallback)")
        self.device = device
        self._pinned = []

    def close(self):
        if self.handle:
            for p in self._pinned:
                self.lib.ecgpu_host_free(self.handle, p)
            self._pinned = []
            self.lib.ecgpu_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != 0:
            raise EcgpuError(f"ecgpu error {rc}: {self.lib.ecgpu_last_error(self.handle).decode()}")

    def set_stream(self, stream_handle: int):
        """All later launches go to this hipStream_t; 0 / None is the legacy default stream (PyTorch's default stream)."""
        self.check(self.lib.ecgpu_set_stream(self.handle, ctypes.c_void_p(stream_handle or None)))

    def use_own_stream(self):
        """Back to the context's own (blocking) stream."""
        self.check(self.lib.ecgpu_use_own_stream(self.handle))

    def set_option(self, option: int, value: int):
        self.check(self.lib.ecgpu_set_option(self.handle, option, value))

    def get_option(self, option: int) -> int:
        v = ctypes.c_int64()
        self.check(self.lib.ecgpu_get_option(self.handle, option, ctypes.byref(v)))
        return v.value

    def fb_table_bytes(self, curve) -> tuple:
        """(bytes of device memory held by the curve's generator tables, widest window among them)"""
        cid = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        b, w = ctypes.c_size_t(), ctypes.c_int()
        self.check(self.lib.ecgpu_fb_table_bytes(self.handle, cid, ctypes.byref(b), ctypes.byref(w)))
        return b.value, w.value

    def last_error(self) -> str:
        buf = ctypes.create_string_buffer(512)
        self.lib.ecgpu_last_error_copy(self.handle, buf, 512)
        return buf.value.decode()

    def debug_workspace(self, which: int) -> bytes:
        """contents of one of the context's device workspaces (ecgpu_debug_workspace): 0 table workspace, 1 ECDSA / ECDH
        intermediates, 2 MSM, 16 + i staging slot i.  b"" if it does not exist yet."""
        b = ctypes.c_size_t()
        self.check(self.lib.ecgpu_debug_workspace(self.handle, which, None, 0, ctypes.byref(b)))
        if not b.value:
            return b""
        buf = np.empty(b.value, dtype=np.uint8)
        self.check(self.lib.ecgpu_debug_workspace(self.handle, which, ctypes.c_void_p(buf.ctypes.data), b.value, ctypes.byref(b)))
        return buf.tobytes()

    def synchronize(self):
        self.check(self.lib.ecgpu_synchronize(self.handle))

    def pinned_array(self, shape, dtype=np.uint8) -> np.ndarray:
        """numpy array over page-locked host memory (ecgpu_host_alloc); freed with the context."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        self.check(self.lib.ecgpu_host_alloc(self.handle, nbytes, ctypes.byref(p)))
        self._pinned.append(p)
        buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def timer_start(self):
        self.check(self.lib.ecgpu_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = ctypes.c_float()
        self.check(self.lib.ecgpu_timer_stop(self.handle, ctypes.byref(ms)))
        return ms.value

    def curve(self, name_or_id) -> "Curve":
        cid = CURVE_IDS[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        return Curve(self, cid)


def _host_out(n: int, width: int) -> np.ndarray:
    return np.zeros((n, width), dtype=np.uint8)


def _as_host(x, width: int) -> np.ndarray:
    """bytes / list of bytes / ndarray -> (n, width) uint8."""
    if isinstance(x, np.ndarray):
        a = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, width)
    elif isinstance(x, (bytes, bytearray)):
        a = np.frombuffer(bytes(x), dtype=np.uint8).reshape(-1, width).copy()
    else:
        a = np.frombuffer(b"".join(x), dtype=np.uint8).reshape(-1, width).copy()
    return a


class Curve:
    """Batch versions of the reference's per-curve arithmetic surface.

    Host-memory methods take/return numpy uint8 arrays shaped (n, bytes) in the canonical
    big-endian wire format; the `*_device` methods take raw device pointers / torch tensors and
    are asynchronous on the context stream."""

    def __init__(self, ctx: Context, cid: int):
        self.ctx, self.id, self.nb = ctx, cid, FIELD_BYTES[cid]

    # --- FieldElement -----------------------------------------------------------------------
    def field_op(self, op: int, a, b=None) -> np.ndarray:
        a = _as_host(a, self.nb)
        bb = _as_host(b, self.nb) if b is not None else None
        out = _host_out(len(a), self.nb)
        pa, _ = _ptr(a); pb, _ = _ptr(bb); po, _ = _ptr(out)
        self.ctx.check(self.ctx.lib.ecgpu_field_op_batch(self.ctx.handle, self.id, op, pa, pb, po, len(a), HOST))
        return out

    # --- ProjectivePoint::{add, add_mixed, double}, BatchNormalize ------------------------------
    def add(self, p_xyz, q_xyz) -> np.ndarray:
        p, q = _as_host(p_xyz, 3 * self.nb), _as_host(q_xyz, 3 * self.nb)
        out = _host_out(len(p), 3 * self.nb)
        self.ctx.check(self.ctx.lib.ecgpu_point_add_batch(self.ctx.handle, self.id, _ptr(p)[0], _ptr(q)[0], _ptr(out)[0], len(p), HOST))
        return out

    def add_mixed(self, p_xyz, q_xy) -> np.ndarray:
        p, q = _as_host(p_xyz, 3 * self.nb), _as_host(q_xy, 2 * self.nb)
        out = _host_out(len(p), 3 * self.nb)
        self.ctx.check(self.ctx.lib.ecgpu_point_add_mixed_batch(self.ctx.handle, self.id, _ptr(p)[0], _ptr(q)[0], _ptr(out)[0], len(p), HOST))
        return out

    def double(self, p_xyz) -> np.ndarray:
        p = _as_host(p_xyz, 3 * self.nb)
        out = _host_out(len(p), 3 * self.nb)
        self.ctx.check(self.ctx.lib.ecgpu_point_double_batch(self.ctx.handle, self.id, _ptr(p)[0], _ptr(out)[0], len(p), HOST))
        return out

    def batch_normalize(self, p_xyz):
        p = _as_host(p_xyz, 3 * self.nb)
        out, inf = _host_out(len(p), 2 * self.nb), np.zeros(len(p), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_batch_normalize(self.ctx.handle, self.id, _ptr(p)[0], _ptr(out)[0], _ptr(inf)[0], len(p), HOST))
        return out, inf

    def add_device(self, d_p_xyz, d_q_xyz, d_out_xyz, n: int):
        """complete addition on device-resident projective points (torch tensors / device pointers)"""
        for t, nm in ((d_p_xyz, "d_p_xyz"), (d_q_xyz, "d_q_xyz"), (d_out_xyz, "d_out_xyz")):
            self._check_device(t, n * 3 * self.nb, nm)
        self.ctx.check(self.ctx.lib.ecgpu_point_add_batch(self.ctx.handle, self.id, _ptr(d_p_xyz)[0], _ptr(d_q_xyz)[0], _ptr(d_out_xyz)[0], n, DEVICE))

    def batch_normalize_device(self, d_p_xyz, d_out_xy, d_out_inf, n: int):
        self._check_device(d_p_xyz, n * 3 * self.nb, "d_p_xyz")
        self._check_device(d_out_xy, n * 2 * self.nb, "d_out_xy")
        self._check_device(d_out_inf, n, "d_out_inf")
        self.ctx.check(self.ctx.lib.ecgpu_batch_normalize(self.ctx.handle, self.id, _ptr(d_p_xyz)[0], _ptr(d_out_xy)[0], _ptr(d_out_inf)[0], n, DEVICE))

    def point_eq(self, p_xyz, q_xyz) -> np.ndarray:
        """ProjectivePoint == ProjectivePoint (ct_eq) per element -> uint8 flags"""
        p, q = _as_host(p_xyz, 3 * self.nb), _as_host(q_xyz, 3 * self.nb)
        if len(p) != len(q):
            raise ValueError("point batches differ in length")
        eq = np.zeros(len(p), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_point_eq_batch(self.ctx.handle, self.id, _ptr(p)[0], _ptr(q)[0], _ptr(eq)[0], len(p), HOST))
        return eq

    # --- Mul<Scalar>, MulByGenerator, LinearCombination -----------------------------------------
    def lincomb(self, scalars, points, terms: int = 1, point_format: int = AFFINE, out_format: int = AFFINE, flags: int = 0,
                out=None, out_inf=None, checked: bool = False):
        """out / out_inf: optional preallocated result arrays (e.g. Context.pinned_array) instead of fresh ones.
        checked=True also returns scalar_ok (Scalar::from_repr: 0 where a scalar of the element is >= n)."""
        s = _as_host(scalars, self.nb)
        if terms < 1 or len(s) % terms:
            raise ValueError("the number of scalars is not a multiple of `terms`")
        n = len(s) // terms
        pw = (3 if point_format == PROJECTIVE else 2) * self.nb
        ow = (3 if out_format == PROJECTIVE else 2) * self.nb
        p = _as_host(points, pw) if points is not None else None
        if p is not None and len(p) != n * terms:
            raise ValueError("scalars and points differ in count (%d scalars, %d points)" % (len(s), len(p)))
        out = _host_out(n, ow) if out is None else out
        inf = np.zeros(n, dtype=np.uint8) if out_inf is None else out_inf
        if (out.shape != (n, ow) or out.dtype != np.uint8 or not out.flags.c_contiguous
                or inf.shape != (n,) or inf.dtype != np.uint8 or not inf.flags.c_contiguous):
            raise ValueError("out / out_inf have the wrong shape, dtype or layout")
        if checked:
            ok = np.zeros(n, dtype=np.uint8)
            self.ctx.check(self.ctx.lib.ecgpu_lincomb_batch_checked(self.ctx.handle, self.id, _ptr(s)[0], _ptr(p)[0], point_format, terms,
                                                                    _ptr(out)[0], out_format, _ptr(inf)[0], _ptr(ok)[0], n, HOST, flags))
            return (out, inf, ok) if out_format == AFFINE else (out, ok)
        self.ctx.check(self.ctx.lib.ecgpu_lincomb_batch(self.ctx.handle, self.id, _ptr(s)[0], _ptr(p)[0], point_format, terms,
                                                        _ptr(out)[0], out_format, _ptr(inf)[0], n, HOST, flags))
        return (out, inf) if out_format == AFFINE else out

    def mul(self, scalars, points, point_format: int = AFFINE, out_format: int = AFFINE, flags: int = 0, out=None, out_inf=None):
        return self.lincomb(scalars, points, 1, point_format, out_format, flags, out, out_inf)

    def mul_by_generator(self, scalars, out_format: int = AFFINE, flags: int = 0):
        return self.lincomb(scalars, None, 1, AFFINE, out_format, flags)

    def diffie_hellman(self, secret_scalars, public_keys_xy) -> np.ndarray:
        """elliptic_curve::ecdh::diffie_hellman for a batch: SharedSecret = x((public * secret).to_affine())
        (k256/src/ecdh.rs:41-45).  Inputs are what the reference's types guarantee: non-zero scalars, valid keys.
        The scalars are secret: the multiplication runs on the constant-time variable-base kernels (csrc/varbase_ct.hpp on
        P-256 / P-384, csrc/varbase_ct_k256.hpp on secp256k1)."""
        shared, ok = self.ecdh(secret_scalars, public_keys_xy)
        if not ok.all():
            raise ValueError("diffie_hellman: element %d is not a NonZeroScalar / PublicKey pair" % int(np.argmin(ok)))
        return shared

    def ecdh(self, secret_scalars, public_keys_xy):
        """ecgpu_ecdh_batch -> (shared_x (n, NB), ok (n,)): ok = 0 and zeros for a zero / out-of-range secret or an invalid key"""
        d, q = _as_host(secret_scalars, self.nb), _as_host(public_keys_xy, 2 * self.nb)
        if len(d) != len(q):
            raise ValueError("secret and public-key batches differ in length")
        out, ok = _host_out(len(d), self.nb), np.zeros(len(d), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_ecdh_batch(self.ctx.handle, self.id, _ptr(d)[0], _ptr(q)[0], _ptr(out)[0], _ptr(ok)[0], len(d), HOST))
        return out, ok

    def ecdh_device(self, d_secret, d_pubkeys_xy, d_shared_x, d_ok, n: int):
        self._check_device(d_secret, n * self.nb, "d_secret")
        self._check_device(d_pubkeys_xy, n * 2 * self.nb, "d_pubkeys_xy")
        self._check_device(d_shared_x, n * self.nb, "d_shared_x")
        self._check_device(d_ok, n, "d_ok")
        self.ctx.check(self.ctx.lib.ecgpu_ecdh_batch(self.ctx.handle, self.id, _ptr(d_secret)[0], _ptr(d_pubkeys_xy)[0], _ptr(d_shared_x)[0], _ptr(d_ok)[0], n, DEVICE))

    def _check_device(self, t, need_bytes: int, what: str):
        """size check for torch tensors handed to the *_device methods (raw integer pointers cannot be checked)"""
        if t is not None and hasattr(t, "numel") and t.numel() * t.element_size() < need_bytes:
            raise ValueError("%s holds %d bytes, the call needs %d" % (what, t.numel() * t.element_size(), need_bytes))

    def mul_device(self, d_scalars, d_points, d_out, n: int, point_format: int = AFFINE, out_format: int = AFFINE,
                   d_out_inf=None, flags: int = 0):
        self._check_device(d_scalars, n * self.nb, "d_scalars")
        self._check_device(d_points, n * (3 if point_format == PROJECTIVE else 2) * self.nb, "d_points")
        self._check_device(d_out, n * (3 if out_format == PROJECTIVE else 2) * self.nb, "d_out")
        self._check_device(d_out_inf, n, "d_out_inf")
        self.ctx.check(self.ctx.lib.ecgpu_mul_batch(self.ctx.handle, self.id, _ptr(d_scalars)[0], _ptr(d_points)[0], point_format,
                                                    _ptr(d_out)[0], out_format, _ptr(d_out_inf)[0], n, DEVICE, flags))

    def msm(self, scalars, points, point_format: int = AFFINE, out_format: int = AFFINE) -> np.ndarray:
        s = _as_host(scalars, self.nb)
        pw = (3 if point_format == PROJECTIVE else 2) * self.nb
        p = _as_host(points, pw)
        if len(p) != len(s):
            raise ValueError("scalars and points differ in count (%d scalars, %d points)" % (len(s), len(p)))
        out = _host_out(1, (3 if out_format == PROJECTIVE else 2) * self.nb)
        self.ctx.check(self.ctx.lib.ecgpu_msm(self.ctx.handle, self.id, _ptr(s)[0], _ptr(p)[0], point_format, len(s), _ptr(out)[0], out_format, HOST))
        return out[0]

    def msm_device(self, d_scalars, d_points, n: int, d_out, point_format: int = AFFINE, out_format: int = AFFINE):
        self._check_device(d_scalars, n * self.nb, "d_scalars")
        self._check_device(d_points, n * (3 if point_format == PROJECTIVE else 2) * self.nb, "d_points")
        self._check_device(d_out, (3 if out_format == PROJECTIVE else 2) * self.nb, "d_out")
        self.ctx.check(self.ctx.lib.ecgpu_msm(self.ctx.handle, self.id, _ptr(d_scalars)[0], _ptr(d_points)[0], point_format, n,
                                              _ptr(d_out)[0], out_format, DEVICE))

    # --- decoding ---------------------------------------------------------------------------------
    def validate_scalars(self, scalars) -> np.ndarray:
        s = _as_host(scalars, self.nb)
        ok = np.zeros(len(s), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_validate_scalars(self.ctx.handle, self.id, _ptr(s)[0], _ptr(ok)[0], len(s), HOST))
        return ok

    def validate_points(self, points_xy) -> np.ndarray:
        p = _as_host(points_xy, 2 * self.nb)
        ok = np.zeros(len(p), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_validate_points(self.ctx.handle, self.id, _ptr(p)[0], _ptr(ok)[0], len(p), HOST))
        return ok

    def decompress(self, xs, y_is_odd):
        x = _as_host(xs, self.nb)
        odd = np.ascontiguousarray(y_is_odd, dtype=np.uint8)
        out, ok = _host_out(len(x), 2 * self.nb), np.zeros(len(x), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_decompress_batch(self.ctx.handle, self.id, _ptr(x)[0], _ptr(odd)[0], _ptr(out)[0], _ptr(ok)[0], len(x), HOST))
        return out, ok

    # --- GroupEncoding::{to_bytes, from_bytes} ------------------------------------------------------
    def to_bytes(self, points, point_format: int = AFFINE) -> np.ndarray:
        p = _as_host(points, (3 if point_format == PROJECTIVE else 2) * self.nb)
        out = _host_out(len(p), self.nb + 1)
        self.ctx.check(self.ctx.lib.ecgpu_to_bytes_batch(self.ctx.handle, self.id, _ptr(p)[0], point_format, _ptr(out)[0], len(p), HOST))
        return out

    def from_bytes(self, encoded):
        e = _as_host(encoded, self.nb + 1)
        out, ok = _host_out(len(e), 2 * self.nb), np.zeros(len(e), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_from_bytes_batch(self.ctx.handle, self.id, _ptr(e)[0], _ptr(out)[0], _ptr(ok)[0], len(e), HOST))
        return out, ok

    # --- ToEncodedPoint::to_encoded_point / FromEncodedPoint::from_encoded_point (fixed-width records) -------------
    def sec1_encode(self, points, compress: bool = False, point_format: int = AFFINE) -> np.ndarray:
        p = _as_host(points, (3 if point_format == PROJECTIVE else 2) * self.nb)
        out = _host_out(len(p), 1 + (1 if compress else 2) * self.nb)
        self.ctx.check(self.ctx.lib.ecgpu_sec1_encode_batch(self.ctx.handle, self.id, _ptr(p)[0], point_format, int(bool(compress)), _ptr(out)[0], len(p), HOST))
        return out

    def sec1_decode(self, encoded, record_bytes: Optional[int] = None):
        rb = record_bytes or (1 + 2 * self.nb)
        e = _as_host(encoded, rb)
        out, ok = _host_out(len(e), 2 * self.nb), np.zeros(len(e), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_sec1_decode_batch(self.ctx.handle, self.id, _ptr(e)[0], rb, _ptr(out)[0], _ptr(ok)[0], len(e), HOST))
        return out, ok

    # --- ECDSA: VerifyPrimitive::verify_prehashed / SignPrimitive::try_sign_prehashed ------------------
    def default_ecdsa_flags(self) -> int:
        """secp256k1 is used with low-s rules in the reference (k256/src/ecdsa.rs:182-207), the NIST curves are not."""
        return ECDSA_LOW_S if self.id == K256 else 0

    def ecdsa_verify(self, prehash, sig_rs, pubkeys_xy, flags: Optional[int] = None) -> np.ndarray:
        z, sg, q = _as_host(prehash, self.nb), _as_host(sig_rs, 2 * self.nb), _as_host(pubkeys_xy, 2 * self.nb)
        if not (len(z) == len(sg) == len(q)):
            raise ValueError("prehash, signature and public-key batches differ in length")
        ok = np.zeros(len(z), dtype=np.uint8)
        fl = self.default_ecdsa_flags() if flags is None else flags
        self.ctx.check(self.ctx.lib.ecgpu_ecdsa_verify_batch(self.ctx.handle, self.id, _ptr(z)[0], _ptr(sg)[0], _ptr(q)[0], _ptr(ok)[0], len(z), HOST, fl))
        return ok

    def ecdsa_verify_device(self, d_prehash, d_sig_rs, d_pubkeys_xy, d_ok, n: int, flags: Optional[int] = None):
        fl = self.default_ecdsa_flags() if flags is None else flags
        self.ctx.check(self.ctx.lib.ecgpu_ecdsa_verify_batch(self.ctx.handle, self.id, _ptr(d_prehash)[0], _ptr(d_sig_rs)[0], _ptr(d_pubkeys_xy)[0],
                                                             _ptr(d_ok)[0], n, DEVICE, fl))

    def map_to_curve(self, u, count: int = 1):
        """MapToCurve::map_to_curve (count = 1) or Q0 + Q1 of hash_from_bytes (count = 2) -> (points_xy, inf)"""
        uu = _as_host(u, self.nb)
        n = len(uu) // count
        out, inf = _host_out(n, 2 * self.nb), np.zeros(n, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_map_to_curve_batch(self.ctx.handle, self.id, _ptr(uu)[0], count, _ptr(out)[0], _ptr(inf)[0], n, HOST))
        return out, inf

    def ecdsa_recover(self, prehash, sig_rs, recovery_id, flags: Optional[int] = None):
        """VerifyingKey::recover_from_prehash for a batch -> (pubkeys_xy, ok)"""
        z, sg = _as_host(prehash, self.nb), _as_host(sig_rs, 2 * self.nb)
        rid = np.ascontiguousarray(recovery_id, dtype=np.uint8).reshape(-1)
        if not (len(z) == len(sg) == len(rid)):
            raise ValueError("prehash, signature and recovery-id batches differ in length")
        out, ok = _host_out(len(z), 2 * self.nb), np.zeros(len(z), dtype=np.uint8)
        fl = self.default_ecdsa_flags() if flags is None else flags
        self.ctx.check(self.ctx.lib.ecgpu_ecdsa_recover_batch(self.ctx.handle, self.id, _ptr(z)[0], _ptr(sg)[0], _ptr(rid)[0], _ptr(out)[0], _ptr(ok)[0],
                                                              len(z), HOST, fl))
        return out, ok

    def schnorr_verify(self, pubkeys_x, sig_rs, challenges) -> np.ndarray:
        """EC part of BIP340 verification; challenges = tagged challenge hashes (ecgpu.schnorr computes them)."""
        x, sg, e = _as_host(pubkeys_x, self.nb), _as_host(sig_rs, 2 * self.nb), _as_host(challenges, self.nb)
        if not (len(x) == len(sg) == len(e)):
            raise ValueError("key, signature and challenge batches differ in length")
        ok = np.zeros(len(x), dtype=np.uint8)
        self.ctx.check(self.ctx.lib.ecgpu_schnorr_verify_batch(self.ctx.handle, self.id, _ptr(x)[0], _ptr(sg)[0], _ptr(e)[0], _ptr(ok)[0], len(x), HOST))
        return ok

    def schnorr_verify_device(self, d_pubkeys_x, d_sig_rs, d_challenges, d_ok, n: int):
        self.ctx.check(self.ctx.lib.ecgpu_schnorr_verify_batch(self.ctx.handle, self.id, _ptr(d_pubkeys_x)[0], _ptr(d_sig_rs)[0], _ptr(d_challenges)[0],
                                                               _ptr(d_ok)[0], n, DEVICE))

    def ecdsa_sign(self, secret_d, nonce_k, prehash, flags: Optional[int] = None):
        """-> (sig_rs, recovery_id, ok)"""
        d, k, z = _as_host(secret_d, self.nb), _as_host(nonce_k, self.nb), _as_host(prehash, self.nb)
        if not (len(d) == len(k) == len(z)):
            raise ValueError("key, nonce and prehash batches differ in length")
        sig, rec, ok = _host_out(len(d), 2 * self.nb), np.zeros(len(d), dtype=np.uint8), np.zeros(len(d), dtype=np.uint8)
        fl = self.default_ecdsa_flags() if flags is None else flags
        self.ctx.check(self.ctx.lib.ecgpu_ecdsa_sign_batch(self.ctx.handle, self.id, _ptr(d)[0], _ptr(k)[0], _ptr(z)[0], _ptr(sig)[0], _ptr(rec)[0],
                                                           _ptr(ok)[0], len(d), HOST, fl))
        return sig, rec, ok

    def ecdsa_sign_device(self, d_secret, d_nonce, d_prehash, d_sig_rs, d_recid, d_ok, n: int, flags: Optional[int] = None):
        fl = self.default_ecdsa_flags() if flags is None else flags
        self.ctx.check(self.ctx.lib.ecgpu_ecdsa_sign_batch(self.ctx.handle, self.id, _ptr(d_secret)[0], _ptr(d_nonce)[0], _ptr(d_prehash)[0],
                                                           _ptr(d_sig_rs)[0], _ptr(d_recid)[0], _ptr(d_ok)[0], n, DEVICE, fl))

    # --- synthetic inputs (device buffers) ---------------------------------------------------------
    def synth_scalars_device(self, d_out, n: int, seed: int, first_index: int = 0):
        self.ctx.check(self.ctx.lib.ecgpu_synth_scalars(self.ctx.handle, self.id, seed, first_index, _ptr(d_out)[0], n))

    def synth_points_device(self, d_out, n: int, seed: int, first_index: int = 0):
        self.ctx.check(self.ctx.lib.ecgpu_synth_points(self.ctx.handle, self.id, seed, first_index, _ptr(d_out)[0], n))


GROUP_NO_RCCL = 1


class _BorrowedContext(Context):
    """a group member's context (owned by the group: never destroyed from here)"""

    def __init__(self, lib, handle, device):
        self.lib, self.handle, self.device, self._pinned = lib, ctypes.c_void_p(handle), device, []

    def close(self):
        for p in self._pinned:
            self.lib.ecgpu_host_free(self.handle, p)
        self._pinned = []
        self.handle = ctypes.c_void_p()


class Group:
    """Device group (include/ecgpu.h, "device groups"): the single-call entry points split over several GPUs - contiguous index
    ranges for independent batches (no collective), per-device bucket method + all-gather of one point per device + fold for one
    split sum.  `devices` may repeat a device (two contexts on one card)."""

    def __init__(self, devices: Sequence[int], flags: int = 0):
        self.lib = load_library()
        self.handle = ctypes.c_void_p()
        arr = (ctypes.c_int * len(devices))(*devices)
        rc = self.lib.ecgpu_group_create(ctypes.byref(self.handle), arr, len(devices), flags)
        if rc != 0:
            raise EcgpuError(f"ecgpu_group_create({list(devices)}) failed with {rc} (no usable gfx950 GPU? there is no CPU fallback)")
        self.devices = list(devices)
        self.size = len(devices)
        self.members = [_BorrowedContext(self.lib, self.lib.ecgpu_group_context(self.handle, i), d) for i, d in enumerate(devices)]

    def close(self):
        if self.handle:
            for m in self.members:
                m.close()
            self.lib.ecgpu_group_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != 0:
            raise EcgpuError(f"ecgpu group error {rc}: {self.lib.ecgpu_group_last_error(self.handle).decode()}")

    def context(self, index: int) -> Context:
        return self.members[index]

    def gather_path(self) -> str:
        return self.lib.ecgpu_group_gather_path(self.handle).decode()

    def synchronize(self):
        self.check(self.lib.ecgpu_group_synchronize(self.handle))

    def lincomb(self, curve, scalars, points, terms: int = 1, point_format: int = AFFINE, out_format: int = AFFINE, flags: int = 0, out=None, out_inf=None):
        cid = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        nb = FIELD_BYTES[cid]
        s = _as_host(scalars, nb)
        n = len(s) // terms
        pw, ow = (3 if point_format == PROJECTIVE else 2) * nb, (3 if out_format == PROJECTIVE else 2) * nb
        p = _as_host(points, pw) if points is not None else None
        if len(s) % terms or (p is not None and len(p) != n * terms):
            raise ValueError("scalars and points differ in count")
        out = _host_out(n, ow) if out is None else out
        inf = np.zeros(n, dtype=np.uint8) if out_inf is None else out_inf
        self.check(self.lib.ecgpu_group_lincomb_batch(self.handle, cid, _ptr(s)[0], _ptr(p)[0], point_format, terms, _ptr(out)[0], out_format, _ptr(inf)[0], n, flags))
        return (out, inf) if out_format == AFFINE else out

    def mul(self, curve, scalars, points, **kw):
        return self.lincomb(curve, scalars, points, 1, **kw)

    def msm(self, curve, scalars, points, point_format: int = AFFINE, out_format: int = AFFINE) -> np.ndarray:
        cid = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        nb = FIELD_BYTES[cid]
        s = _as_host(scalars, nb)
        p = _as_host(points, (3 if point_format == PROJECTIVE else 2) * nb)
        if len(p) != len(s):
            raise ValueError("scalars and points differ in count")
        out = _host_out(1, (3 if out_format == PROJECTIVE else 2) * nb)
        self.check(self.lib.ecgpu_group_msm(self.handle, cid, _ptr(s)[0], _ptr(p)[0], point_format, len(s), _ptr(out)[0], out_format))
        return out[0]

    @staticmethod
    def _ptr_array(xs):
        return (ctypes.c_void_p * len(xs))(*[(_ptr(x)[0].value if x is not None else None) for x in xs])

    def lincomb_sharded(self, curve, d_scalars, d_points, d_out, counts, terms: int = 1, point_format: int = AFFINE, out_format: int = AFFINE, d_out_inf=None,
                        flags: int = 0):
        """device-resident shards (lists of torch tensors / device pointers, one per member); asynchronous: Group.synchronize()"""
        cid = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        cnt = (ctypes.c_size_t * self.size)(*counts)
        self.check(self.lib.ecgpu_group_lincomb_sharded(self.handle, cid, self._ptr_array(d_scalars), self._ptr_array(d_points) if d_points is not None else None,
                                                        point_format, terms, self._ptr_array(d_out), out_format,
                                                        self._ptr_array(d_out_inf) if d_out_inf is not None else None, cnt, flags))

    def msm_sharded(self, curve, d_scalars, d_points, counts, point_format: int = AFFINE, out_format: int = AFFINE) -> np.ndarray:
        cid = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        nb = FIELD_BYTES[cid]
        cnt = (ctypes.c_size_t * self.size)(*counts)
        out = _host_out(1, (3 if out_format == PROJECTIVE else 2) * nb)
        self.check(self.lib.ecgpu_group_msm_sharded(self.handle, cid, self._ptr_array(d_scalars), self._ptr_array(d_points), point_format, cnt, _ptr(out)[0], out_format))
        return out[0]
