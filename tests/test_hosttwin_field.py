"""Checks the device field templates (compiled for the host, tests/hosttwin) against big integers.
Mirrors the reference's fuzzy_* proptests (k256/src/arithmetic/field.rs:792-872)."""
import random

import pytest

from oracle import ecmodel as M
from hosttwin_util import lib, buf, outbuf

P = M.K256.p
C = 2**256 - P
EDGE = [0, 1, 2, 977, C - 1, C, C + 1, P - 2, P - 1, P, P + 1, 2**256 - 2, 2**256 - 1, 2**255, 2**128 - 1,
        2**256 - C - 1, 2**256 - C // 2, 2**32 - 1, 2**32, 2**64 - 1, (1 << 256) - (1 << 32)]


def run(op, xs, ys=None, raw=0):
    n = len(xs)
    a = b"".join(x.to_bytes(32, "big") for x in xs)
    b = b"".join(y.to_bytes(32, "big") for y in ys) if ys is not None else None
    out = outbuf(32 * n)
    rc = lib().ht_k256_fe_op(op, buf(a), buf(b) if b else None, out, n, raw)
    assert rc == 0
    o = bytes(out)
    return [int.from_bytes(o[32 * i:32 * i + 32], "big") for i in range(n)]


def pairs():
    rng = random.Random(11)
    xs, ys = [], []
    for x in EDGE:
        for y in EDGE:
            xs.append(x); ys.append(y)
    for _ in range(3000):
        xs.append(rng.randrange(2**256)); ys.append(rng.randrange(2**256))
    return xs, ys


@pytest.mark.parametrize("op,fn", [(0, lambda x, y: x * y), (2, lambda x, y: x + y), (3, lambda x, y: x - y)])
def test_binary_ops_on_raw_inputs(op, fn):
    xs, ys = pairs()
    got = run(op, xs, ys)
    for x, y, g in zip(xs, ys, got):
        assert g == fn(x, y) % P, (op, hex(x), hex(y))
    # weakly reduced outputs stay below 2^256 and congruent
    raw = run(op, xs, ys, raw=1)
    for x, y, g in zip(xs, ys, raw):
        assert g % P == fn(x, y) % P


def test_fused_sum_of_two_products():
    """k256::mul_add2 (a b + e f with one reduction; the sum reaches 2^513, word 16 folds through C^2)."""
    xs, ys = pairs()
    M = 2**256 - 1
    for op, fn in ((12, lambda x, y: 2 * x * y), (13, lambda x, y: x * y + (M - x) * (M - y))):
        got = run(op, xs, ys)
        for x, y, g in zip(xs, ys, got):
            assert g == fn(x, y) % P, (op, hex(x), hex(y))
        raw = run(op, xs, ys, raw=1)
        for x, y, g in zip(xs, ys, raw):
            assert g < 2**256 and g % P == fn(x, y) % P


def test_unary_ops():
    rng = random.Random(5)
    xs = EDGE + [rng.randrange(2**256) for _ in range(1500)]
    for x, g in zip(xs, run(1, xs)):
        assert g == x * x % P
    for x, g in zip(xs, run(4, xs)):
        assert g == (-x) % P
    for x, g in zip(xs, run(8, xs)):
        assert g == x % P
    # funnel-shift multiples by 2, 4, 8 on raw inputs (everything below 2^256 is a valid weak representative),
    # including the values whose fold carries out of word 1 and past 2^256
    hard = [2**256 - 1, 2**255, 2**255 + 2**64 - 1, 2**256 - 2**32, (2**256 - 1) ^ (2**64 - 1) | (2**64 - 978), 2**254 + 2**64 - 1,
            2**253 + 2**64 - 5, P, P - 1, P + 1, 2**256 - 978]
    for op, k in ((9, 2), (10, 4), (11, 8)):
        vals = xs + hard
        for x, g in zip(vals, run(op, vals)):
            assert g == x * k % P, (op, hex(x))
        for x, g in zip(vals, run(op, vals, raw=1)):
            assert g < 2**256 and g % P == x * k % P
    ks = [0, 1, 2, 3, 7, 8, 21, 24, 168, 65535] * 200
    xk = [rng.randrange(2**256) for _ in ks]
    got = run(7, xk, ks)
    for x, k, g in zip(xk, ks, got):
        assert g == x * k % P


def test_inv_sqrt():
    rng = random.Random(6)
    xs = [1, 2, P - 1, P + 1, C] + [rng.randrange(1, P) for _ in range(60)]
    for x, g in zip(xs, run(5, xs)):
        assert g * x % P == 1
    assert run(5, [0, P]) == [0, 0]
    for x, g in zip(xs, run(6, xs)):
        want = M.field_sqrt(M.K256, x)
        if want is None:
            assert g == 2**256 - 1
        else:
            assert g == want


def test_reference_field_kats(ref_vectors):
    k = {n: int(v, 16) for n, v in ref_vectors["k256"]["field_kat"].items()}
    a, b = k["a"], k["b"]
    assert run(2, [a], [b]) == [k["add"]]
    assert run(0, [a], [b]) == [k["mul"]]
    assert run(1, [a]) == [k["square_a"]]
    assert run(4, [a]) == [k["negate_a"]]
    na, nb = run(4, [a])[0], run(4, [b])[0]
    assert run(2, [na], [nb]) == [k["add_negated"]]
    v = 1
    for want in ref_vectors["k256"]["field_dbl"]:
        assert v == int(want, 16)
        v = run(2, [v], [v])[0]
