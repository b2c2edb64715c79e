"""Generic Jacobian group law (host build of jacobian.hpp) for all three curves, incl. the exceptional
cases of the incomplete formulas, against the independent affine model."""
import random

import pytest

from oracle import ecmodel as M
from oracle import synth
from hosttwin_util import lib, buf, outbuf

CURVES = [("k256", 0), ("p256", 1), ("p384", 2)]


@pytest.mark.parametrize("cn,cid", CURVES)
def test_jacobian_ops(cn, cid):
    c = M.CURVES[cn]
    p, nb = c.p, c.nbytes
    rng = random.Random(71)

    def fe(v):
        return int(v % p).to_bytes(nb, "big")

    def jacp(A):
        if A is None:
            return (rng.randrange(p), rng.randrange(p), 0)
        z = rng.randrange(1, p)
        return (A[0] * z * z % p, A[1] * z * z * z % p, z)

    def back(o, i):
        X, Y, Z = (int.from_bytes(o[3 * nb * i + nb * t:3 * nb * i + nb * (t + 1)], "big") for t in range(3))
        if Z == 0:
            return None
        zi = pow(Z, -1, p)
        return (X * zi * zi % p, Y * zi * zi * zi % p)

    pts = [synth.point(c, i, seed=71) for i in range(24)]
    singles = pts[:10] + [None]
    jin = b"".join(b"".join(fe(v) for v in jacp(a)) for a in singles)
    out = outbuf(3 * nb * len(singles))
    assert lib().ht_jac_op(cid, 0, buf(jin), None, out, len(singles)) == 0
    o = bytes(out)
    for i, a in enumerate(singles):
        assert back(o, i) == M.affine_add(c, a, a), ("dbl", i)
    pairs = [(pts[i], pts[i + 1]) for i in range(0, 20, 2)] + [(pts[0], pts[0]), (pts[1], M.affine_neg(c, pts[1])), (None, pts[2])]
    jin = b"".join(b"".join(fe(v) for v in jacp(a)) for a, _ in pairs)
    qin = b"".join(fe(b[0]) + fe(b[1]) for _, b in pairs)
    out = outbuf(3 * nb * len(pairs))
    assert lib().ht_jac_op(cid, 1, buf(jin), buf(qin), out, len(pairs)) == 0
    o = bytes(out)
    for i, (a, b) in enumerate(pairs):
        assert back(o, i) == M.affine_add(c, a, b), ("add_mixed", i)
    # add_affine: the accumulator is (x, y, 1) - the fixed-base kernel's second entry; same point and opposite points included
    pairs_a = [(a, b) for a, b in pairs if a is not None]
    jin = b"".join(fe(a[0]) + fe(a[1]) + fe(1) for a, _ in pairs_a)
    qin = b"".join(fe(b[0]) + fe(b[1]) for _, b in pairs_a)
    out = outbuf(3 * nb * len(pairs_a))
    assert lib().ht_jac_op(cid, 3, buf(jin), buf(qin), out, len(pairs_a)) == 0
    o = bytes(out)
    for i, (a, b) in enumerate(pairs_a):
        assert back(o, i) == M.affine_add(c, a, b), ("add_affine", i)
    pairs2 = pairs + [(pts[3], None), (None, None)]
    jin = b"".join(b"".join(fe(v) for v in jacp(a)) for a, _ in pairs2)
    qin = b"".join(b"".join(fe(v) for v in jacp(b)) for _, b in pairs2)
    out = outbuf(3 * nb * len(pairs2))
    assert lib().ht_jac_op(cid, 2, buf(jin), buf(qin), out, len(pairs2)) == 0
    o = bytes(out)
    for i, (a, b) in enumerate(pairs2):
        assert back(o, i) == M.affine_add(c, a, b), ("add", i)


@pytest.mark.parametrize("cn,cid", CURVES)
def test_xyzz_bucket_accumulator(cn, cid):
    """XYZZ mixed addition of the MSM bucket sums (msm.hpp, all curves), the general XYZZ addition that folds bucket pieces
    (every fourth entry) and the conversions to and from Jacobian, with the exceptional cases: accumulator at infinity,
    the same point (doubling), opposite points."""
    c = M.CURVES[cn]
    p, nb = c.p, c.nbytes
    rng = random.Random(72)

    def fe(v):
        return int(v % p).to_bytes(nb, "big")

    def xyzz(A):
        if A is None:
            return (rng.randrange(p), rng.randrange(p), 0, 0)
        z = rng.randrange(1, p)
        return (A[0] * z * z % p, A[1] * z * z * z % p, z * z % p, z * z * z % p)

    pts = [synth.point(c, i, seed=72) for i in range(40)]
    pairs = [(pts[i], pts[i + 1]) for i in range(0, 36, 2)] + [(pts[0], pts[0]), (pts[1], M.affine_neg(c, pts[1])), (None, pts[2]), (pts[5], pts[5])]
    # every fourth entry (index = 2 mod 4) is summed by the general XYZZ addition: its exceptional cases at such indices too
    pairs += [(pts[1], M.affine_neg(c, pts[1])), (pts[6], pts[7]), (pts[8], pts[9]), (pts[10], pts[11]), (None, pts[2]), (pts[12], pts[13]),
              (pts[14], pts[15]), (pts[16], pts[17]), (pts[8], pts[8])]
    assert [i for i, (a, b) in enumerate(pairs) if i % 4 == 2 and (a is None or a == b or a == M.affine_neg(c, b))] == [18, 22, 26, 30]
    pin = b"".join(b"".join(fe(v) for v in xyzz(a)) for a, _ in pairs)
    qin = b"".join(fe(b[0]) + fe(b[1]) for _, b in pairs)
    out = outbuf(3 * nb * len(pairs))
    assert lib().ht_xyzz_add_mixed(cid, buf(pin), buf(qin), out, len(pairs)) == 0
    o = bytes(out)
    for i, (a, b) in enumerate(pairs):
        X, Y, Z = (int.from_bytes(o[3 * nb * i + nb * t:3 * nb * i + nb * (t + 1)], "big") for t in range(3))
        got = None if Z == 0 else (X * pow(Z, -2, p) % p, Y * pow(Z, -3, p) % p)
        assert got == M.affine_add(c, a, b), i
