"""Device k256 point / scalar-mul templates (host build) against the oracle: exact (X, Y, Z)."""
import random

from oracle import ecmodel as M
from hosttwin_util import lib, buf, outbuf

C = M.K256
N = C.n


def rand_points(rng, n, projective=True):
    pts = []
    for _ in range(n):
        A = M.affine_mul(C, rng.randrange(1, N), (C.gx, C.gy))
        if projective:
            z = rng.randrange(1, C.p)
            pts.append((A[0] * z % C.p, A[1] * z % C.p, z))
        else:
            pts.append(A)
    return pts


def call_pt(op, ps, qs=None):
    n = len(ps)
    out = outbuf(96 * n)
    a = b"".join(M.proj_bytes(C, p) for p in ps)
    q = b"".join(qs) if qs is not None else None
    assert lib().ht_k256_pt_op(op, buf(a), buf(q) if q else None, out, n) == 0
    o = bytes(out)
    return [o[96 * i:96 * i + 96] for i in range(n)]


def test_point_ops_exact_xyz():
    rng = random.Random(21)
    ps = rand_points(rng, 40) + [M.IDENTITY, C.G, M.point_neg(C, C.G), C.G, M.IDENTITY]
    qs = rand_points(rng, 40) + [C.G, M.IDENTITY, C.G, C.G, M.IDENTITY]
    got = call_pt(0, ps, [M.proj_bytes(C, q) for q in qs])
    for p, q, g in zip(ps, qs, got):
        assert g == M.proj_bytes(C, M.k256_add(p, q))
    got = call_pt(2, ps)
    for p, g in zip(ps, got):
        assert g == M.proj_bytes(C, M.k256_double(p))
    got = call_pt(3, ps)
    for p, g in zip(ps, got):
        assert g == M.proj_bytes(C, M.k256_neg(p))
    got = call_pt(4, ps)
    for p, g in zip(ps, got):
        assert g == M.proj_bytes(C, M.k256_endomorphism(p))
    aff = [M.to_affine(C, q) for q in qs]
    got = call_pt(1, ps, [M.affine_bytes(C, a) for a in aff])
    for p, a, g in zip(ps, aff, got):
        assert g == M.proj_bytes(C, M.k256_add_mixed(p, a))


def test_glv_split_matches_reference():
    rng = random.Random(22)
    ks = [0, 1, 2, N - 1, N - 2, (N - 1) // 2, 2**128 - 1, 2**128 + 1, N // 2, N // 2 + 1] + [rng.randrange(N) for _ in range(3000)]
    out = outbuf(34 * len(ks))
    assert lib().ht_k256_glv(buf(b"".join(M.i2b(C, k) for k in ks)), out, len(ks)) == 0
    o = bytes(out)
    for i, k in enumerate(ks):
        r1, r2 = M.k256_decompose_scalar(k)
        s1, s2 = M.k256_is_high(r1), M.k256_is_high(r2)
        r1c = (N - r1) % N if s1 else r1
        r2c = (N - r2) % N if s2 else r2
        e = o[34 * i:34 * i + 34]
        assert int.from_bytes(e[:16], "big") == r1c and int.from_bytes(e[16:32], "big") == r2c
        # the sign of a zero half is irrelevant (digits are all zero)
        assert (e[32] == int(s1) or r1c == 0) and (e[33] == int(s2) or r2c == 0)


def test_mul_ref_exact_xyz_and_vectors(ref_vectors):
    rng = random.Random(23)
    ks = [0, 1, 2, N - 1, N - 2, (N - 1) // 2, 2**128 - 1, 2**128 + 1] + [rng.randrange(N) for _ in range(24)]
    ps = [C.G] * 4 + [M.point_neg(C, C.G), M.IDENTITY] + rand_points(rng, len(ks) - 6)
    out = outbuf(96 * len(ks))
    assert lib().ht_k256_mul_ref(buf(b"".join(M.proj_bytes(C, p) for p in ps)), buf(b"".join(M.i2b(C, k) for k in ks)), out, len(ks)) == 0
    o = bytes(out)
    for i, (p, k) in enumerate(zip(ps, ks)):
        assert o[96 * i:96 * i + 96] == M.proj_bytes(C, M.k256_mul_ref(p, k)), i
    # MUL_TEST_VECTORS through mul_by_generator (exact XYZ + affine against the vectors)
    vec = ref_vectors["k256"]["group"]["mul"]
    ks = [int(k, 16) for k, _, _ in vec]
    out = outbuf(96 * len(ks))
    assert lib().ht_k256_mul_gen_ref(buf(b"".join(M.i2b(C, k) for k in ks)), out, len(ks)) == 0
    o = bytes(out)
    aff = outbuf(65 * len(ks))
    assert lib().ht_k256_to_affine(buf(o), aff, len(ks)) == 0
    a = bytes(aff)
    for i, (k, x, y) in enumerate(vec):
        assert o[96 * i:96 * i + 96] == M.proj_bytes(C, M.k256_mul_by_generator_ref(ks[i]))
        assert a[65 * i:65 * i + 65].hex() == (x + y).lower() + "00"


def test_lincomb2_and_to_affine():
    rng = random.Random(24)
    n = 6
    ps = rand_points(rng, 2 * n)
    ks = [rng.randrange(N) for _ in range(2 * n)]
    out = outbuf(96 * n)
    assert lib().ht_k256_lincomb2_ref(buf(b"".join(M.proj_bytes(C, p) for p in ps)), buf(b"".join(M.i2b(C, k) for k in ks)), out, n) == 0
    o = bytes(out)
    for i in range(n):
        want = M.k256_lincomb_ref([(ps[2 * i], ks[2 * i]), (ps[2 * i + 1], ks[2 * i + 1])])
        assert o[96 * i:96 * i + 96] == M.proj_bytes(C, want)
    pts = ps[:4] + [M.IDENTITY]
    aff = outbuf(65 * len(pts))
    assert lib().ht_k256_to_affine(buf(b"".join(M.proj_bytes(C, p) for p in pts)), aff, len(pts)) == 0
    a = bytes(aff)
    for i, p in enumerate(pts):
        assert a[65 * i:65 * i + 65] == M.affine_bytes(C, M.to_affine(C, p))


def test_config1_sample():
    from conftest import load_config1
    fx = load_config1("k256")
    rows = fx["rows"][:48]
    pts = b"".join(bytes.fromhex(r[1]) + bytes.fromhex(r[2]) + (1).to_bytes(32, "big") for r in rows)
    ks = b"".join(bytes.fromhex(r[0]) for r in rows)
    out = outbuf(96 * len(rows))
    assert lib().ht_k256_mul_ref(buf(pts), buf(ks), out, len(rows)) == 0
    aff = outbuf(65 * len(rows))
    assert lib().ht_k256_to_affine(buf(bytes(out)), aff, len(rows)) == 0
    a = bytes(aff)
    for i, r in enumerate(rows):
        assert a[65 * i:65 * i + 65].hex() == r[3]
