"""Throughput schedule of the k256 scalar multiplication (Jacobian, common-Z table, batched
inversion), host build of the device templates, against the independent affine model."""
import random

from oracle import ecmodel as M
from oracle import synth
from hosttwin_util import lib, buf, outbuf

C = M.K256
N, P = C.n, C.p


def jac_to_affine(X, Y, Z):
    if Z % P == 0:
        return None
    zi = pow(Z, -1, P)
    return (X * zi * zi % P, Y * zi * zi * zi % P)


def fe(v):
    return int(v % P).to_bytes(32, "big")


def test_jacobian_primitives_including_exceptional_cases():
    rng = random.Random(31)
    cases = []
    for i in range(40):
        A = synth.point(C, i, seed=31)
        B = synth.point(C, 100 + i, seed=31)
        z = rng.randrange(1, P)
        cases.append(((A[0] * z * z % P, A[1] * z * z * z % P, z), B, M.affine_add(C, A, B)))
    A = synth.point(C, 7, seed=31)
    z = rng.randrange(1, P)
    J = (A[0] * z * z % P, A[1] * z * z * z % P, z)
    cases.append((J, A, M.affine_add(C, A, A)))                       # same point: doubling branch
    cases.append((J, M.affine_neg(C, A), None))                       # opposite: infinity
    cases.append(((0, 0, 0), A, A))                                   # accumulator at infinity
    cases.append(((5, 9, 0), A, A))                                   # infinity with junk X, Y
    cases.append(((A[0], A[1], 1), A, M.affine_add(C, A, A)))         # Z = 1 doubling
    pin = b"".join(fe(j[0]) + fe(j[1]) + fe(j[2]) for j, _, _ in cases)
    qin = b"".join(fe(q[0]) + fe(q[1]) for _, q, _ in cases)
    out = outbuf(96 * len(cases))
    assert lib().ht_k256_jac_add_mixed(buf(pin), buf(qin), out, len(cases)) == 0
    o = bytes(out)
    for i, (_, _, want) in enumerate(cases):
        X, Y, Z = (int.from_bytes(o[96 * i + 32 * t:96 * i + 32 * t + 32], "big") for t in range(3))
        assert jac_to_affine(X, Y, Z) == want, i
    out = outbuf(96 * len(cases))
    assert lib().ht_k256_jac_double(buf(pin), out, len(cases)) == 0
    o = bytes(out)
    for i, (j, _, _) in enumerate(cases):
        X, Y, Z = (int.from_bytes(o[96 * i + 32 * t:96 * i + 32 * t + 32], "big") for t in range(3))
        a = jac_to_affine(*j)
        assert jac_to_affine(X, Y, Z) == (None if a is None else M.affine_add(C, a, a)), i


import pytest

WB = 0          # window width the walks below use: 0 = the product's default (K256_WB), 4 | 5 explicitly


def run_fast(pts_bytes, proj, ks, batch):
    n = len(ks)
    out = outbuf(65 * n)
    sb = buf(b"".join(M.i2b(C, k) for k in ks))
    if WB:
        assert lib().ht_k256_mul_fast_w(WB, buf(pts_bytes), proj, sb, out, n, batch) == 0
    else:
        assert lib().ht_k256_mul_fast(buf(pts_bytes), proj, sb, out, n, batch) == 0
    o = bytes(out)
    return [o[65 * i:65 * i + 65] for i in range(n)]


@pytest.fixture(params=[4, 5], autouse=True)
def window_width(request):
    """every test of this file runs on both window widths of the throughput schedule (mulfast_k256.hpp: 4 bits - the two-term kernel - and
    5 bits - the single-term kernel since round 4)"""
    global WB
    WB = request.param
    yield
    WB = 0


def test_mul_fast_affine_inputs_edges_and_random():
    rng = random.Random(32)
    lam = M.K256_LAMBDA
    ks = [0, 1, 2, 3, 7, 8, 9, 15, 16, 17, N - 1, N - 2, N - 8, N - 9, (N - 1) // 2, (N + 1) // 2, 2**128 - 1, 2**128, 2**128 + 1,
          lam, lam + 1, N - lam, 2 * lam % N, 16 * lam % N]
    # the 5-bit recoding: digits -16 and 15, fields across the word boundaries of a half (bits 30-34, 60-64, 95-99, 125-129), in either half
    ks += [31, 32, 33, 16 * 32**6, 31 * 32**6 + 15, 16 * 32**12, 31 * 32**19, 16 * 32**25 % N, (2**127 - 1), 31 * lam % N, 32 * lam % N, (16 * 32**6 * lam) % N,
           (31 * 32**12 + (16 * 32**19) * lam) % N, (2**127 - 1) * lam % N]
    ks += [rng.randrange(N) for _ in range(70)] + [rng.randrange(2**32) for _ in range(10)]
    pts = [synth.point(C, i, seed=32) for i in range(len(ks))]
    pts[3] = None
    pts[5] = (C.gx, C.gy)
    pb = b"".join(bytes(64) if p is None else M.i2b(C, p[0]) + M.i2b(C, p[1]) for p in pts)
    for batch in (1, 5, 16):
        got = run_fast(pb, 0, ks, batch)
        for i, (k, p) in enumerate(zip(ks, pts)):
            w = M.affine_mul(C, k, p)
            assert got[i] == M.affine_bytes(C, (0, 0, 1) if w is None else (w[0], w[1], 0)), (batch, i)


def test_mul_fast_projective_inputs():
    rng = random.Random(33)
    ks = [rng.randrange(N) for _ in range(20)] + [5, 0]
    pts = []
    for i in range(len(ks)):
        x, y = synth.point(C, i, seed=33)
        z = rng.randrange(1, P)
        pts.append((x * z % P, y * z % P, z))
    pts[4] = M.IDENTITY
    got = run_fast(b"".join(M.proj_bytes(C, p) for p in pts), 1, ks, 8)
    for i, (k, p) in enumerate(zip(ks, pts)):
        w = M.affine_mul(C, k, M.to_affine_opt(C, p))
        assert got[i] == M.affine_bytes(C, (0, 0, 1) if w is None else (w[0], w[1], 0)), i


def test_mul_fast_group_vectors_and_config1(ref_vectors):
    from conftest import load_config1
    vec = ref_vectors["k256"]["group"]["mul"]
    g = M.i2b(C, C.gx) + M.i2b(C, C.gy)
    got = run_fast(g * len(vec), 0, [int(k, 16) for k, _, _ in vec], 16)
    for i, (_, x, y) in enumerate(vec):
        assert got[i].hex() == (x + y).lower() + "00"
    rows = load_config1("k256")["rows"][:200]
    got = run_fast(b"".join(bytes.fromhex(r[1] + r[2]) for r in rows), 0, [int(r[0], 16) for r in rows], 16)
    for i, r in enumerate(rows):
        assert got[i].hex() == r[3]


def test_general_jacobian_add_with_exceptional_cases():
    rng = random.Random(35)

    def jac(A):
        if A is None:
            return (rng.randrange(P), rng.randrange(P), 0)
        z = rng.randrange(1, P)
        return (A[0] * z * z % P, A[1] * z * z * z % P, z)

    pts = [synth.point(C, i, seed=35) for i in range(30)]
    cases = [(pts[i], pts[i + 1]) for i in range(0, 28, 2)]
    cases += [(pts[0], pts[0]), (pts[1], M.affine_neg(C, pts[1])), (None, pts[2]), (pts[3], None), (None, None)]
    pin = b"".join(b"".join(fe(v) for v in jac(a)) for a, _ in cases)
    qin = b"".join(b"".join(fe(v) for v in jac(b)) for _, b in cases)
    out = outbuf(96 * len(cases))
    assert lib().ht_k256_jac_add(buf(pin), buf(qin), out, len(cases)) == 0
    o = bytes(out)
    for i, (a, b) in enumerate(cases):
        X, Y, Z = (int.from_bytes(o[96 * i + 32 * t:96 * i + 32 * t + 32], "big") for t in range(3))
        assert jac_to_affine(X, Y, Z) == M.affine_add(C, a, b), i
