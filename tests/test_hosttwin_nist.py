"""Device templates for P-256 / P-384 (host build) against the oracle: field ops, exact (X, Y, Z)
of the RCB a = -3 formulas and of the primeorder 4-bit window multiplication."""
import random

import pytest

from oracle import ecmodel as M
from oracle import synth
from hosttwin_util import lib, buf, outbuf

CURVES = [("p256", 1), ("p384", 2)]


def fe_run(cid, c, op, xs, ys=None):
    nb = c.nbytes
    out = outbuf(nb * len(xs))
    a = b"".join(int(x).to_bytes(nb, "big") for x in xs)
    b = b"".join(int(y).to_bytes(nb, "big") for y in ys) if ys is not None else None
    assert lib().ht_nist_fe_op(cid, op, buf(a), buf(b) if b else None, out, len(xs)) == 0
    o = bytes(out)
    return [int.from_bytes(o[nb * i:nb * i + nb], "big") for i in range(len(xs))]


@pytest.mark.parametrize("cn,cid", CURVES)
def test_field_ops(cn, cid, ref_vectors):
    c = M.CURVES[cn]
    p = c.p
    rng = random.Random(41)
    edge = [0, 1, 2, 3, p - 1, p - 2, p - 3, 2**32 - 1, 2**32, 2**96 - 1, 2**96, (p - 1) // 2, 2**(8 * c.nbytes - 1) % p, p >> 1]
    xs = [a for a in edge for _ in edge] + [rng.randrange(p) for _ in range(1500)]
    ys = [b for _ in edge for b in edge] + [rng.randrange(p) for _ in range(1500)]
    assert fe_run(cid, c, 0, xs, ys) == [x * y % p for x, y in zip(xs, ys)]
    # the same edge values as INTERNAL (Montgomery-form) operands: x = e R^-1 is stored as e
    rinv = pow(1 << (8 * c.nbytes), -1, p)
    hi = [p - 1, p - 2, 2**(8 * c.nbytes - 32) - 1, (1 << (8 * c.nbytes - 1)) % p, p - 2**32, p - 2**96, p - 2**128]
    xm = [(a * rinv) % p for a in edge + hi for _ in edge + hi]
    ym = [(b * rinv) % p for _ in edge + hi for b in edge + hi]
    assert fe_run(cid, c, 0, xm, ym) == [x * y % p for x, y in zip(xm, ym)]
    xs2 = sorted(set(xm)) + [rng.randrange(p) for _ in range(1500)]
    assert fe_run(cid, c, 1, xs2) == [x * x % p for x in xs2]          # p384 squares on its own path
    assert fe_run(cid, c, 2, xs, ys) == [(x + y) % p for x, y in zip(xs, ys)]
    assert fe_run(cid, c, 3, xs, ys) == [(x - y) % p for x, y in zip(xs, ys)]
    assert fe_run(cid, c, 1, xs) == [x * x % p for x in xs]
    assert fe_run(cid, c, 4, xs) == [(-x) % p for x in xs]
    sm = edge + [rng.randrange(1, p) for _ in range(12)]
    for x, g in zip(sm, fe_run(cid, c, 5, sm)):
        assert (x == 0 and g == 0) or g * x % p == 1
    for x, g in zip(sm, fe_run(cid, c, 6, sm)):
        want = M.field_sqrt(c, x)
        assert g == ((1 << (8 * c.nbytes)) - 1 if want is None else want)
    if cn == "p256":
        dbl = [int(v, 16) for v in ref_vectors["p256"]["field_dbl"]]
        assert fe_run(cid, c, 2, dbl[:-1], dbl[:-1]) == dbl[1:]


def rand_proj(c, rng, n, start=0):
    out = []
    for i in range(n):
        x, y = synth.point(c, start + i, seed=42)
        z = rng.randrange(1, c.p)
        out.append((x * z % c.p, y * z % c.p, z))
    return out


@pytest.mark.parametrize("cn,cid", CURVES)
def test_point_ops_exact_xyz(cn, cid):
    c = M.CURVES[cn]
    rng = random.Random(43)
    ps = rand_proj(c, rng, 24) + [M.IDENTITY, c.G, M.point_neg(c, c.G), c.G, M.IDENTITY]
    qs = rand_proj(c, rng, 24, 100) + [c.G, M.IDENTITY, c.G, c.G, M.IDENTITY]
    w = 3 * c.nbytes
    out = outbuf(w * len(ps))
    pa = b"".join(M.proj_bytes(c, p) for p in ps)
    assert lib().ht_nist_pt_op(cid, 0, buf(pa), buf(b"".join(M.proj_bytes(c, q) for q in qs)), out, len(ps)) == 0
    assert bytes(out) == b"".join(M.proj_bytes(c, M.am3_add(c, p, q)) for p, q in zip(ps, qs))
    out = outbuf(w * len(ps))
    assert lib().ht_nist_pt_op(cid, 2, buf(pa), None, out, len(ps)) == 0
    assert bytes(out) == b"".join(M.proj_bytes(c, M.am3_double(c, p)) for p in ps)
    aff = [M.to_affine(c, q) for q in qs]
    out = outbuf(w * len(ps))
    assert lib().ht_nist_pt_op(cid, 1, buf(pa), buf(b"".join(M.affine_bytes(c, a) for a in aff)), out, len(ps)) == 0
    assert bytes(out) == b"".join(M.proj_bytes(c, M.am3_add_mixed(c, p, a)) for p, a in zip(ps, aff))


@pytest.mark.parametrize("cn,cid", CURVES)
def test_mul_ref_exact_xyz_and_vectors(cn, cid, ref_vectors):
    c = M.CURVES[cn]
    rng = random.Random(44)
    ks = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2] + [rng.randrange(c.n) for _ in range(6)]
    ps = [c.G, c.G, M.point_neg(c, c.G), M.IDENTITY] + rand_proj(c, rng, len(ks) - 4, 200)
    w = 3 * c.nbytes
    out = outbuf(w * len(ks))
    assert lib().ht_nist_mul_ref(cid, buf(b"".join(M.proj_bytes(c, p) for p in ps)), buf(b"".join(M.i2b(c, k) for k in ks)), out, len(ks), 0) == 0
    o = bytes(out)
    for i, (p, k) in enumerate(zip(ps, ks)):
        assert o[w * i:w * i + w] == M.proj_bytes(c, M.primeorder_mul_ref(c, p, k)), i
    vec = ref_vectors[cn]["group"]["mul"][:10]
    out = outbuf(w * len(vec))
    assert lib().ht_nist_mul_ref(cid, None, buf(b"".join(bytes.fromhex(k) for k, _, _ in vec)), out, len(vec), 1) == 0
    o = bytes(out)
    for i, (k, x, y) in enumerate(vec):
        X, Y, Z = (int.from_bytes(o[w * i + c.nbytes * t:w * i + c.nbytes * (t + 1)], "big") for t in range(3))
        assert M.to_affine_opt(c, (X, Y, Z)) == (int(x, 16), int(y, 16))


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_hash_to_curve_map(cn, cid, ref_vectors):
    """h2c_map.hpp on the host: the RFC 9380 vectors of <curve>/src/arithmetic/hash2curve.rs (u -> Q0, Q1 and Q0 + Q1 = P),
    then edge and random field elements against the model - the map keeps x as a fraction, evaluates the secp256k1
    isogeny in homogeneous form and inverts once per output."""
    import ctypes
    c = M.CURVES[cn]
    nb = c.nbytes
    L = lib()
    L.ht_h2c_map.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]

    def run(us, count):
        n = len(us) // count
        out, inf = outbuf(2 * nb * n), outbuf(n)
        assert L.ht_h2c_map(cid, buf(b"".join(M.i2b(c, u) for u in us)), count, out, inf, n) == 0
        o = bytes(out)
        return [o[2 * nb * i:2 * nb * (i + 1)] for i in range(n)], bytes(inf)

    vs = ref_vectors[cn]["hash2curve"]
    us = [int(v[k], 16) for v in vs for k in ("u_0", "u_1")]
    q, inf = run(us, 1)
    assert not any(inf)
    for i, v in enumerate(vs):
        assert q[2 * i].hex() == v["q0_x"] + v["q0_y"] and q[2 * i + 1].hex() == v["q1_x"] + v["q1_y"]
    pts, inf = run(us, 2)
    assert [p.hex() for p in pts] == [v["p_x"] + v["p_y"] for v in vs] and not any(inf)
    rng = random.Random(9380)
    us = [0, 1, 2, c.p - 1, c.p - 2] + [rng.randrange(c.p) for _ in range(60)]
    got, _ = run(us, 1)
    for x, g in zip(us, got):
        Q = M.map_to_curve(c, x)
        assert g == M.i2b(c, Q[0]) + M.i2b(c, Q[1]), x
    pairs = [(us[i], us[i + 1]) for i in range(0, 40, 2)] + [(7, 7), (9, c.p - 9)]      # equal points; u and -u map to P and -P
    got, inf = run([u for ab in pairs for u in ab], 2)
    for (a, b), g, f in zip(pairs, got, inf):
        S = M.affine_add(c, M.map_to_curve(c, a), M.map_to_curve(c, b))
        if S is None:
            assert f == 1 and not g.strip(b"\0")
        else:
            assert f == 0 and g == M.i2b(c, S[0]) + M.i2b(c, S[1])
    assert inf[-1] == 1
