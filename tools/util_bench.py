"""Throughput of the secondary entry points (SURVEY.md 8f rows and the element-wise field / point ops) on device-resident
batches: python tools/util_bench.py [log2n]   -> one line per (curve, operation), M elements/s"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import torch
import ecgpu
from oracle import synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
ctx = ecgpu.Context(0)
lib, h = ctx.lib, ctx.handle
vp = ctypes.c_void_p


def P(t):
    return vp(t.data_ptr()) if t is not None else None


def timed(name, cn, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.timer_start()
        rc = fn()
        ms = ctx.timer_stop()
        assert rc in (0, None), (name, rc, lib.ecgpu_last_error(h))
        best = min(best, ms)
    print(f"{cn:5s} {name:34s} n=2^{lg}  {best:8.3f} ms  {n / best / 1e3:9.2f} M/s", flush=True)


for cn in ("k256", "p256", "p384"):
    cv = ctx.curve(cn)
    nb, cid = cv.nb, cv.id
    u8 = dict(dtype=torch.uint8, device="cuda")
    s = torch.empty((n, nb), **u8); s2 = torch.empty((n, nb), **u8)
    p = torch.empty((n, 2 * nb), **u8); o = torch.empty((n, 3 * nb), **u8); o2 = torch.empty((n, 3 * nb), **u8)
    f = torch.empty((n,), **u8); f2 = torch.empty((n,), **u8)
    cv.synth_scalars_device(s, n, synth.SEED, 0); cv.synth_scalars_device(s2, n, synth.SEED + 1, 0)
    cv.synth_points_device(p, n, synth.SEED, 0)
    ctx.synchronize()
    x = p[:, :nb].contiguous(); y = p[:, nb:].contiguous()
    fo = torch.empty((n, nb), **u8)
    for op, nm in ((0, "field mul"), (1, "field sqr"), (2, "field add"), (5, "field invert"), (6, "field sqrt")):
        timed(nm, cn, lambda op=op: lib.ecgpu_field_op_batch(h, cid, op, P(x), P(y), P(fo), n, 1))
    cv.mul_device(s, p, o, n, out_format=ecgpu.PROJECTIVE, flags=0)
    cv.mul_device(s2, p, o2, n, out_format=ecgpu.PROJECTIVE, flags=0)
    ctx.synchronize()
    q = torch.empty((n, 3 * nb), **u8)
    timed("point add (complete, X Y Z)", cn, lambda: lib.ecgpu_point_add_batch(h, cid, P(o), P(o2), P(q), n, 1))
    timed("point double", cn, lambda: lib.ecgpu_point_double_batch(h, cid, P(o), P(q), n, 1))
    timed("point eq", cn, lambda: lib.ecgpu_point_eq_batch(h, cid, P(o), P(o2), P(f), n, 1))
    xy = torch.empty((n, 2 * nb), **u8)
    timed("batch_normalize", cn, lambda: lib.ecgpu_batch_normalize(h, cid, P(q), P(xy), P(f), n, 1))
    timed("validate_points", cn, lambda: lib.ecgpu_validate_points(h, cid, P(p), P(f), n, 1))
    odd = (y[:, -1] & 1).contiguous()
    timed("decompress", cn, lambda: lib.ecgpu_decompress_batch(h, cid, P(x), P(odd), P(xy), P(f), n, 1))
    enc = torch.empty((n, nb + 1), **u8)
    timed("to_bytes (SEC1 compressed)", cn, lambda: lib.ecgpu_to_bytes_batch(h, cid, P(p), 0, P(enc), n, 1))
    timed("from_bytes", cn, lambda: lib.ecgpu_from_bytes_batch(h, cid, P(enc), P(xy), P(f), n, 1))
    unc = torch.empty((n, 2 * nb + 1), **u8)
    timed("sec1_encode (uncompressed)", cn, lambda: lib.ecgpu_sec1_encode_batch(h, cid, P(p), 0, 0, P(unc), n, 1))
    timed("sec1_decode (uncompressed)", cn, lambda: lib.ecgpu_sec1_decode_batch(h, cid, P(unc), 2 * nb + 1, P(xy), P(f), n, 1))
    timed("map_to_curve (2 elements, sum)", cn, lambda: lib.ecgpu_map_to_curve_batch(h, cid, P(torch.cat([x, y], 1)), 2, P(xy), P(f), n, 1))
    timed("mul_by_generator (throughput)", cn, lambda: cv.mul_device(s, None, xy, n))
    timed("mul_by_generator (reference, CT)", cn, lambda: cv.mul_device(s, None, xy, n, flags=ecgpu.EXACT_REFERENCE))
    timed("mul (throughput)", cn, lambda: cv.mul_device(s, p, xy, n))
    timed("mul (reference, CT)", cn, lambda: cv.mul_device(s, p, xy, n, flags=ecgpu.EXACT_REFERENCE), reps=2)
    timed("mul (SECRET_SCALARS: ECDH)", cn, lambda: cv.mul_device(s, p, xy, n, flags=ecgpu.SECRET_SCALARS), reps=2)
    keys = torch.empty((n, 2 * nb), **u8); sig = torch.empty((n, 2 * nb), **u8); rec = torch.empty((n,), **u8)
    cv.mul_device(s, None, keys, n)
    fl = cv.default_ecdsa_flags()
    timed("ecdsa sign (CT k G, default)", cn, lambda: cv.ecdsa_sign_device(s, s2, x, sig, rec, f, n, flags=fl), reps=2)
    timed("ecdsa sign (PUBLIC_SCALARS)", cn, lambda: cv.ecdsa_sign_device(s, s2, x, sig, rec, f, n, flags=fl | ecgpu.PUBLIC_SCALARS))
    timed("ecdsa verify", cn, lambda: cv.ecdsa_verify_device(x, sig, keys, f, n, flags=fl))
    ctx.synchronize()
    assert bool(f.all()), "verification of device-made signatures failed"
    timed("ecdsa recover", cn, lambda: lib.ecgpu_ecdsa_recover_batch(h, cid, P(x), P(sig), P(rec), P(xy), P(f2), n, 1, fl))
    ctx.synchronize()
    assert bool(f2.all()) and bool((xy == keys).all()), "recovery did not return the signing keys"
    del s, s2, p, o, o2, q, xy, keys, sig
    torch.cuda.empty_cache()
ctx.close()
