"""ctypes loader for oracle/libecoracle.so (the C restatement; TEST INFRASTRUCTURE, see ecoracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("ECGPU_ORACLE_LIB") or os.path.join(HERE, "libecoracle.so")      # override: a sanitizer build (tools/hosttwin_sanitize.sh)
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        L = ctypes.CDLL(path)
        vp, sz, i, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
        L.eco_lincomb_batch.argtypes = [i, vp, vp, i, vp, sz, i, i]
        L.eco_msm_naive.argtypes = [i, vp, vp, sz, vp]
        L.eco_synth_scalars.argtypes = [i, u64, u64, vp, sz]
        L.eco_synth_points.argtypes = [i, u64, u64, vp, sz]
        L.eco_point_op.argtypes = [i, i, vp, vp, vp, sz]
        L.eco_ecdsa_verify_batch.argtypes = [i, vp, vp, vp, vp, sz, i]
        L.eco_ecdsa_sign_batch.argtypes = [i, vp, vp, vp, vp, vp, vp, sz, i]
        L.eco_p384_invert.argtypes = [i, vp, vp, sz]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def nb(curve):
    return 48 if curve == 2 else 32


def lincomb_batch(curve, scalars, points, terms=1, out_proj=False, threads=1):
    """scalars (n*terms, nb) uint8, points (n*terms, 2nb) or None -> (n, 2nb+1) affine x||y||inf or (n, 3nb)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint8)
    p = None if points is None else np.ascontiguousarray(points, dtype=np.uint8)
    n = s.shape[0] // terms
    w = 3 * nb(curve) if out_proj else 2 * nb(curve) + 1
    out = np.zeros((n, w), dtype=np.uint8)
    assert lib().eco_lincomb_batch(curve, _p(s), _p(p), terms, _p(out), n, int(out_proj), threads) == 0
    return out


def msm_naive(curve, scalars, points):
    s = np.ascontiguousarray(scalars, dtype=np.uint8)
    p = np.ascontiguousarray(points, dtype=np.uint8)
    out = np.zeros(2 * nb(curve) + 1, dtype=np.uint8)
    assert lib().eco_msm_naive(curve, _p(s), _p(p), s.shape[0], _p(out)) == 0
    return out


def synth_scalars(curve, n, seed, first=0):
    out = np.zeros((n, nb(curve)), dtype=np.uint8)
    assert lib().eco_synth_scalars(curve, seed, first, _p(out), n) == 0
    return out


def synth_points(curve, n, seed, first=0):
    out = np.zeros((n, 2 * nb(curve)), dtype=np.uint8)
    assert lib().eco_synth_points(curve, seed, first, _p(out), n) == 0
    return out


def point_op(curve, op, p, q=None):
    """op 0: add (q projective), 1: double, 2: add_mixed (q affine x || y, zeros = identity) -> exact (X, Y, Z) bytes"""
    p = np.ascontiguousarray(p, dtype=np.uint8)
    q = None if q is None else np.ascontiguousarray(q, dtype=np.uint8)
    out = np.zeros_like(p)
    assert lib().eco_point_op(curve, op, _p(p), _p(q), _p(out), p.shape[0]) == 0
    return out


def p384_invert(values, fermat=False):
    """P-384 field inversion of (n, 48) canonical big-endian values: Bernstein-Yang divsteps as the reference
    (p384 field.rs:67-91), or the Fermat chain kept for cross-checking."""
    a = np.ascontiguousarray(values, dtype=np.uint8).reshape(-1, 48)
    out = np.zeros_like(a)
    assert lib().eco_p384_invert(1 if fermat else 0, _p(a), _p(out), a.shape[0]) == 0
    return out


def ecdsa_verify_batch(curve, prehash, sig_rs, pubkeys_xy, low_s=False):
    z = np.ascontiguousarray(prehash, dtype=np.uint8).reshape(-1, nb(curve))
    sg = np.ascontiguousarray(sig_rs, dtype=np.uint8).reshape(-1, 2 * nb(curve))
    q = np.ascontiguousarray(pubkeys_xy, dtype=np.uint8).reshape(-1, 2 * nb(curve))
    ok = np.zeros(z.shape[0], dtype=np.uint8)
    assert lib().eco_ecdsa_verify_batch(curve, _p(z), _p(sg), _p(q), _p(ok), z.shape[0], int(low_s)) == 0
    return ok


def ecdsa_sign_batch(curve, d, k, prehash, low_s=False):
    d = np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, nb(curve))
    k = np.ascontiguousarray(k, dtype=np.uint8).reshape(-1, nb(curve))
    z = np.ascontiguousarray(prehash, dtype=np.uint8).reshape(-1, nb(curve))
    sig = np.zeros((d.shape[0], 2 * nb(curve)), dtype=np.uint8)
    rec = np.zeros(d.shape[0], dtype=np.uint8)
    ok = np.zeros(d.shape[0], dtype=np.uint8)
    assert lib().eco_ecdsa_sign_batch(curve, _p(d), _p(k), _p(z), _p(sig), _p(rec), _p(ok), d.shape[0], int(low_s)) == 0
    return sig, rec, ok
