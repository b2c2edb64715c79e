"""Pins oracle/ecmodel.py against every known-answer vector the reference holds for the path
(SURVEY.md 8c).  CPU only."""
import random

import pytest

from oracle import ecmodel as M
from oracle import synth
from conftest import load_config1

CURVES = ["k256", "p256", "p384"]


def h2i(s):
    return int(s, 16)


@pytest.mark.parametrize("cn", CURVES)
def test_constants(cn):
    c = M.CURVES[cn]
    assert M.on_curve(c, (c.gx, c.gy))
    assert M.affine_mul(c, c.n - 1, (c.gx, c.gy)) == (c.gx, (-c.gy) % c.p)
    assert M.affine_add(c, M.affine_mul(c, c.n - 1, (c.gx, c.gy)), (c.gx, c.gy)) is None


@pytest.mark.parametrize("cn", CURVES)
def test_group_add_vectors(cn, ref_vectors):
    """<curve>/src/test_vectors/group.rs ADD_TEST_VECTORS: k*G for k = 1..20, via repeated complete
    addition, repeated mixed addition and doubling (k256 projective.rs:858-967, primeorder dev.rs:66-155)."""
    c = M.CURVES[cn]
    vec = [(h2i(x), h2i(y)) for x, y in ref_vectors[cn]["group"]["add"]]
    acc = M.IDENTITY
    accm = M.IDENTITY
    ga = (c.gx, c.gy, 0)
    for i, want in enumerate(vec):
        acc = M.point_add(c, acc, c.G)
        accm = M.point_add_mixed(c, accm, ga)
        assert M.to_affine_opt(c, acc) == want
        assert M.to_affine_opt(c, accm) == want
        assert M.affine_mul(c, i + 1, (c.gx, c.gy)) == want
    # doubling: 2^j G for j = 0..4 hits vector indices 0, 1, 3, 7, 15
    d = c.G
    for j in range(5):
        assert M.to_affine_opt(c, d) == vec[(1 << j) - 1]
        d2 = M.point_double(c, d)
        assert M.to_affine(c, d2) == M.to_affine(c, M.point_add(c, d, d))   # add-vs-double
        d = d2
    # add / sub round trip
    assert M.to_affine_opt(c, M.point_add(c, acc, M.point_neg(c, c.G))) == vec[18]


@pytest.mark.parametrize("cn", CURVES)
def test_group_mul_vectors(cn, ref_vectors):
    """MUL_TEST_VECTORS (k, x, y) incl. k near n, against the independent model, the faithful
    `P * k` restatement and the faithful mul_by_generator."""
    c = M.CURVES[cn]
    for k, x, y in ref_vectors[cn]["group"]["mul"]:
        k, want = h2i(k), (h2i(x), h2i(y))
        assert M.affine_mul(c, k, (c.gx, c.gy)) == want
        assert M.to_affine_opt(c, M.mul_ref(c, c.G, k)) == want
        assert M.to_affine_opt(c, M.mul_by_generator_ref(c, k)) == want


@pytest.mark.parametrize("cn", CURVES)
def test_ecdsa_fixed_base_vectors(cn, ref_vectors):
    """d -> Q = d*G and k -> r = (k*G).x mod n from <curve>/src/test_vectors/ecdsa.rs."""
    c = M.CURVES[cn]
    for v in ref_vectors[cn]["ecdsa"]:
        Q = M.to_affine(c, M.mul_by_generator_ref(c, h2i(v["d"])))
        assert (Q[0], Q[1]) == (h2i(v["q_x"]), h2i(v["q_y"]))
        R = M.to_affine(c, M.mul_by_generator_ref(c, h2i(v["k"])))
        assert R[0] % c.n == h2i(v["r"])


@pytest.mark.parametrize("cn", CURVES)
def test_hash2curve_add_triples(cn, ref_vectors):
    """Q0 + Q1 = P (arbitrary-point addition KATs, <curve>/src/arithmetic/hash2curve.rs)."""
    c = M.CURVES[cn]
    assert len(ref_vectors[cn]["hash2curve"]) == 5
    for v in ref_vectors[cn]["hash2curve"]:
        q0 = (h2i(v["q0_x"]), h2i(v["q0_y"]))
        q1 = (h2i(v["q1_x"]), h2i(v["q1_y"]))
        want = (h2i(v["p_x"]), h2i(v["p_y"]))
        assert M.on_curve(c, q0) and M.on_curve(c, q1)
        assert M.affine_add(c, q0, q1) == want
        assert M.to_affine_opt(c, M.point_add(c, q0 + (1,), q1 + (1,))) == want
        assert M.to_affine_opt(c, M.point_add_mixed(c, q0 + (1,), q1 + (0,))) == want


@pytest.mark.parametrize("cn", ["k256", "p256"])
def test_field_doubling_vectors(cn, ref_vectors):
    c = M.CURVES[cn]
    v = 1
    for want in ref_vectors[cn]["field_dbl"]:
        assert v == h2i(want)
        v = v * 2 % c.p
    # repeated_mul (k256 field.rs:677-689): powers of two by multiplication
    assert pow(2, 249, c.p) == h2i(ref_vectors[cn]["field_dbl"][249])


def test_k256_field_kats(ref_vectors):
    """field_8x32_risc0.rs:225-303 (mathematical results: valid for any backend), field_5x52.rs:510-544."""
    p = M.K256.p
    k = {n: h2i(v) for n, v in ref_vectors["k256"]["field_kat"].items()}
    a, b = k["a"], k["b"]
    assert (a + b) % p == k["add"]
    assert (-a - b) % p == k["add_negated"]
    assert (-a) % p == k["negate_a"]
    assert a * b % p == k["mul"]
    assert a * a % p == k["square_a"]
    assert (1 << 256) % p == h2i(ref_vectors["k256"]["field_2pow256"])


def test_k256_glv_and_radix16():
    """mul.rs:129-152 constants, :260-268 split, :274-305 recoding; lambda*G = (beta*Gx, Gy)."""
    c = M.K256
    lam = M.K256_LAMBDA
    assert M.affine_mul(c, lam, (c.gx, c.gy)) == (c.gx * M.K256_BETA % c.p, c.gy)
    rng = random.Random(1)
    ks = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, 2**128 - 1, 2**128 + 1] + [rng.randrange(c.n) for _ in range(2000)]
    for k in ks:
        r1, r2 = M.k256_decompose_scalar(k)
        assert (r1 + r2 * lam) % c.n == k
        for r in (r1, r2):
            rc = c.n - r if M.k256_is_high(r) else r
            assert rc < 2**128
            d = M.radix16_decomposition(rc, 33)
            assert all(-8 <= x <= 7 for x in d) and d[32] >= 0
            assert sum(x << (4 * i) for i, x in enumerate(d)) == rc
    d = M.radix16_decomposition(rng.randrange(c.n), 65)
    assert all(-8 <= x <= 7 for x in d[:64]) and 0 <= d[64] <= 1



# the reduced basis of the GLV lattice {(x, y): x + y lambda = 0 mod n} (k256/src/arithmetic/mul.rs:129-152: a1, b1, a2, b2 = a1)
K256_A1 = 0x3086D221A7D46BCDE86C90E49284EB15
K256_A2 = 0x114CA50F7A8E2F3F657C1108D9D44CFD8


def k256_glv_corner_scalars():
    """Scalars whose split lands next to the corners of the fundamental cell (|alpha|, |beta| -> 1/2), where |k1|, |k2| peak."""
    n, lam = M.K256.n, M.K256_LAMBDA
    b1, b2 = -M.K256_MINUS_B1, K256_A1
    out = []
    for s1 in (-1, 1):
        for s2 in (-1, 1):
            x, y = (s1 * K256_A1 + s2 * K256_A2) // 2, (s1 * b1 + s2 * b2) // 2
            for d in range(-40, 41):
                out.append((x + d + y * lam) % n)
                out.append((x + (y + d) * lam) % n)
    return out


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_fixed_base_ct_top_window(cn):
    """The numeric facts csrc/fixedbase_ct.hpp's exception-freeness argument uses: signed 5-bit digits, S_j = the value of the digits
    below window j, |S_j| <= 16 (32^j - 1) / 31.  Below the top window 17 * 32^j < n; in the top window the one candidate D = -n
    needs |S_j| = n mod 32^j, which is larger than the bound on |S_j|; and |D| < 2n there."""
    c = M.CURVES[cn]
    nwin = (8 * c.nbytes + 4) // 5
    assert (8 * c.nbytes) % 5 != 0
    top = 32 ** (nwin - 1)
    bound = 16 * (top - 1) // 31
    assert 17 * 32 ** (nwin - 2) < c.n
    assert top < c.n and c.n % top > bound
    dmax = ((c.n - 1) >> (5 * (nwin - 1))) + 1            # the top window keeps its value: the scalar's top bits plus a carry
    assert dmax <= 16 and bound + dmax * top < 2 * c.n


def test_k256_glv_bounds():
    """The facts csrc/varbase_ct_k256.hpp's exception-freeness argument stands on: the GLV lattice's shortest non-zero vector in the
    maximum norm is |b1| = 2^127.835, and decompose_scalar stays below (a1 + a2 + 1) / 2 and (b2 - b1) / 2 + 1 (libsecp256k1's bounds),
    i.e. more than 2^126 below it - also next to the corners of the fundamental cell."""
    n, lam = M.K256.n, M.K256_LAMBDA
    a1, a2, b1, b2 = K256_A1, K256_A2, -M.K256_MINUS_B1, K256_A1
    assert (a1 + b1 * lam) % n == 0 and (a2 + b2 * lam) % n == 0 and a1 * b2 - a2 * b1 == n          # a basis of the lattice
    mu = min(max(abs(i * a1 + j * a2), abs(i * b1 + j * b2)) for i in range(-8, 9) for j in range(-8, 9) if (i, j) != (0, 0))
    assert mu == M.K256_MINUS_B1 == 0xE4437ED6010E88286F547FA90ABFE4C3
    bound1, bound2 = (a1 + a2 + 1) // 2, (b2 - b1) // 2 + 1
    assert max(bound1, bound2) + 16 < mu - (1 << 126)
    rng = random.Random(20261004)
    ks = k256_glv_corner_scalars() + [rng.randrange(n) for _ in range(20000)] + list(range(64)) + [n - d for d in range(1, 64)]
    top1 = top2 = 0
    for k in ks:
        r1, r2 = M.k256_decompose_scalar(k)
        assert (r1 + r2 * lam - k) % n == 0
        k1 = r1 if not M.k256_is_high(r1) else n - r1
        k2 = r2 if not M.k256_is_high(r2) else n - r2
        top1, top2 = max(top1, k1), max(top2, k2)
    assert top1 < bound1 and top2 < bound2
    assert top1 > bound1 - 64 and top2 > bound2 - 64          # the corner scalars do reach the bounds

@pytest.mark.parametrize("cn", CURVES)
def test_faithful_equals_independent_random(cn):
    """test_lincomb / test_mul_by_generator / test_lincomb_slice (mul.rs:493-526) restated."""
    c = M.CURVES[cn]
    rng = random.Random(7)
    for _ in range(6):
        k, l = rng.randrange(c.n), rng.randrange(c.n)
        P = M.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy))
        Q = M.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy))
        want = M.affine_add(c, M.affine_mul(c, k, P), M.affine_mul(c, l, Q))
        got = M.lincomb_ref(c, [(P + (1,), k), (Q + (1,), l)])
        assert M.to_affine_opt(c, got) == want
        assert M.to_affine_opt(c, M.mul_by_generator_ref(c, k)) == M.affine_mul(c, k, (c.gx, c.gy))
    # edge scalars and identity inputs
    for k in (0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, 2**128 - 1, 2**128 + 1):
        for P in (c.G, M.point_neg(c, c.G), M.IDENTITY):
            got = M.to_affine_opt(c, M.mul_ref(c, P, k))
            assert got == M.affine_mul(c, k, M.to_affine_opt(c, P))


@pytest.mark.parametrize("cn", CURVES)
def test_config1_fixture_matches_model(cn):
    """BASELINE.json configs[0]: seeded (scalar, point) pairs -> affine k*P; spot-check the committed
    fixture against both models and the synthetic-input spec."""
    c = M.CURVES[cn]
    fx = load_config1(cn)
    assert fx["n"] == len(fx["rows"]) and fx["seed"] == synth.SEED
    step = max(1, fx["n"] // 24)
    for i in range(0, fx["n"], step):
        k, px, py, out = fx["rows"][i]
        assert h2i(k) == synth.scalar(c, i) and (h2i(px), h2i(py)) == synth.point(c, i)
        want = M.affine_mul(c, h2i(k), (h2i(px), h2i(py)))
        assert bytes.fromhex(out) == M.affine_bytes(c, (want[0], want[1], 0))
        assert M.affine_bytes(c, M.to_affine(c, M.mul_ref(c, (h2i(px), h2i(py), 1), h2i(k)))) == bytes.fromhex(out)


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_sec1_encoding_vectors(cn, ref_vectors):
    """Base-point encodings of the reference's affine tests, and the all-zero identity (identity_encoding)."""
    c = M.CURVES[cn]
    e = ref_vectors[cn]["encoding"]
    G = (c.gx, c.gy)
    unc = bytes.fromhex(e["uncompressed_basepoint"])
    assert unc == b"\x04" + M.i2b(c, c.gx) + M.i2b(c, c.gy)
    comp = bytes.fromhex(e["compressed_basepoint"])
    assert M.group_to_bytes(c, G) == comp
    assert M.group_from_bytes(c, comp) == (True, G)
    assert M.group_to_bytes(c, None) == bytes(c.nbytes + 1)
    assert M.group_from_bytes(c, bytes(c.nbytes + 1)) == (True, None)
    if "compact_basepoint" in e:
        ok, P = M.group_from_bytes(c, bytes.fromhex(e["compact_basepoint"]))
        assert ok and b"\x04" + M.i2b(c, P[0]) + M.i2b(c, P[1]) == bytes.fromhex(e["uncompact_basepoint"])
    assert M.group_from_bytes(c, b"\x04" + bytes(c.nbytes))[0] is False
    assert M.group_from_bytes(c, b"\x00" + b"\x01" * c.nbytes)[0] is False


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_hash_to_curve_vectors_pin_the_model(cn, ref_vectors):
    """RFC 9380 vectors held by <curve>/src/arithmetic/hash2curve.rs: msg -> u -> Q0, Q1 -> P."""
    c = M.CURVES[cn]
    vs = ref_vectors[cn]["hash2curve"]
    assert len(vs) == 5
    for v in vs:
        u0, u1 = M.hash_to_field(c, v["msg"].encode(), v["dst"].encode())
        assert (u0, u1) == (int(v["u_0"], 16), int(v["u_1"], 16))
        assert M.map_to_curve(c, u0) == (int(v["q0_x"], 16), int(v["q0_y"], 16))
        assert M.map_to_curve(c, u1) == (int(v["q1_x"], 16), int(v["q1_y"], 16))
        assert M.hash_to_curve(c, v["msg"].encode(), v["dst"].encode()) == (int(v["p_x"], 16), int(v["p_y"], 16))
