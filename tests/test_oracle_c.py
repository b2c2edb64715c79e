"""Pins the C restatement (oracle/ecoracle.c) against the big-integer model and the golden
fixtures: exact (X, Y, Z) of the reference algorithms, affine outputs, synthetic streams."""
import random

import numpy as np
import pytest

from oracle import coracle as CO
from oracle import ecmodel as M
from oracle import synth
from conftest import load_config1

CURVES = [("k256", 0), ("p256", 1), ("p384", 2)]


def arr(rows, w):
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(-1, w).copy()


@pytest.mark.parametrize("cn,cid", CURVES)
def test_synth_streams(cn, cid):
    c = M.CURVES[cn]
    n, first = 48, 1000
    s = CO.synth_scalars(cid, n, synth.SEED, first)
    p = CO.synth_points(cid, n, synth.SEED, first)
    for i in range(n):
        assert bytes(s[i]) == M.i2b(c, synth.scalar(c, first + i))
        x, y = synth.point(c, first + i)
        assert bytes(p[i]) == M.i2b(c, x) + M.i2b(c, y)


@pytest.mark.parametrize("cn,cid", CURVES)
def test_mul_exact_xyz_and_edges(cn, cid):
    c = M.CURVES[cn]
    rng = random.Random(9)
    ks = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, 2**128 - 1, 2**128 + 1] + [rng.randrange(c.n) for _ in range(12)]
    pts = [(c.gx, c.gy)] * 4 + [(c.gx, (-c.gy) % c.p), None] + [synth.point(c, i, seed=3) for i in range(len(ks) - 6)]
    sb = arr([M.i2b(c, k) for k in ks], c.nbytes)
    pb = arr([bytes(2 * c.nbytes) if P is None else M.i2b(c, P[0]) + M.i2b(c, P[1]) for P in pts], 2 * c.nbytes)
    xyz = CO.lincomb_batch(cid, sb, pb, out_proj=True)
    aff = CO.lincomb_batch(cid, sb, pb, threads=3)
    for i, (k, P) in enumerate(zip(ks, pts)):
        Pp = M.IDENTITY if P is None else (P[0], P[1], 1)
        want = M.mul_ref(c, Pp, k)
        assert bytes(xyz[i]) == M.proj_bytes(c, want), (cn, i)
        assert bytes(aff[i]) == M.affine_bytes(c, M.to_affine(c, want))
    g = CO.lincomb_batch(cid, sb, None, out_proj=True)
    for i, k in enumerate(ks):
        assert bytes(g[i]) == M.proj_bytes(c, M.mul_by_generator_ref(c, k)), (cn, i)


@pytest.mark.parametrize("cn,cid", CURVES)
def test_group_vectors_and_config1(cn, cid, ref_vectors):
    c = M.CURVES[cn]
    vec = ref_vectors[cn]["group"]["mul"]
    out = CO.lincomb_batch(cid, arr([bytes.fromhex(k) for k, _, _ in vec], c.nbytes), None)
    for i, (_, x, y) in enumerate(vec):
        assert bytes(out[i]).hex() == (x + y).lower() + "00"
    fx = load_config1(cn)
    rows = fx["rows"][:96]
    out = CO.lincomb_batch(cid, arr([bytes.fromhex(r[0]) for r in rows], c.nbytes),
                           arr([bytes.fromhex(r[1] + r[2]) for r in rows], 2 * c.nbytes), threads=4)
    for i, r in enumerate(rows):
        assert bytes(out[i]).hex() == r[3]


@pytest.mark.parametrize("cn,cid", CURVES)
def test_point_ops_lincomb2_msm(cn, cid):
    c = M.CURVES[cn]
    rng = random.Random(10)
    ps, qs = [], []
    for i in range(24):
        for lst, off in ((ps, 0), (qs, 100)):
            x, y = synth.point(c, i + off, seed=4)
            z = rng.randrange(1, c.p)
            lst.append((x * z % c.p, y * z % c.p, z))
    ps += [M.IDENTITY, c.G]; qs += [c.G, c.G]
    pa, qa = arr([M.proj_bytes(c, p) for p in ps], 3 * c.nbytes), arr([M.proj_bytes(c, q) for q in qs], 3 * c.nbytes)
    got = CO.point_op(cid, 0, pa, qa)
    assert bytes(got) == b"".join(M.proj_bytes(c, M.point_add(c, p, q)) for p, q in zip(ps, qs))
    got = CO.point_op(cid, 1, pa)
    assert bytes(got) == b"".join(M.proj_bytes(c, M.point_double(c, p)) for p in ps)
    # add_mixed (k256 projective.rs:164-221, primeorder point_arithmetic.rs:247-277); an identity among the affine operands
    aff = [M.to_affine(c, q) for q in qs]
    aff[3] = M.to_affine(c, M.IDENTITY)
    qx = arr([M.i2b(c, a[0]) + M.i2b(c, a[1]) for a in aff], 2 * c.nbytes)
    got = CO.point_op(cid, 2, pa, qx)
    assert bytes(got) == b"".join(M.proj_bytes(c, M.point_add_mixed(c, p, a)) for p, a in zip(ps, aff))
    n = 6
    ks = [rng.randrange(c.n) for _ in range(2 * n)]
    pts = [synth.point(c, i, seed=8) for i in range(2 * n)]
    sb = arr([M.i2b(c, k) for k in ks], c.nbytes)
    pb = arr([M.i2b(c, x) + M.i2b(c, y) for x, y in pts], 2 * c.nbytes)
    out = CO.lincomb_batch(cid, sb, pb, terms=2, out_proj=True)
    for i in range(n):
        want = M.lincomb_ref(c, [((pts[2 * i][0], pts[2 * i][1], 1), ks[2 * i]), ((pts[2 * i + 1][0], pts[2 * i + 1][1], 1), ks[2 * i + 1])])
        assert bytes(out[i]) == M.proj_bytes(c, want)
    tot = None
    for k, P in zip(ks, pts):
        tot = M.affine_add(c, tot, M.affine_mul(c, k, P))
    assert bytes(CO.msm_naive(cid, sb, pb)) == M.affine_bytes(c, (tot[0], tot[1], 0))


def test_p384_bernstein_yang_inversion_matches_fermat_and_model():
    """oracle/ecoracle.c inverts in the P-384 base field the way the reference does (Bernstein-Yang divsteps,
    p384/src/arithmetic/field.rs:67-91, primeorder/src/field.rs:505-559): the same field element as a^(p-2) by the Fermat
    chain and as Python's pow(a, -1, p), on edge values and random ones."""
    c = M.P384
    rng = random.Random(384)
    vals = [0, 1, 2, 3, c.p - 1, c.p - 2, (c.p - 1) // 2, (c.p + 1) // 2, 1 << 383, (1 << 383) - 1, 1 << 192] + [rng.randrange(c.p) for _ in range(300)]
    a = arr([v.to_bytes(48, "big") for v in vals], 48)
    by = CO.p384_invert(a)
    fe = CO.p384_invert(a, fermat=True)
    assert bytes(by) == bytes(fe)
    for v, row in zip(vals, by):
        assert int.from_bytes(bytes(row), "big") == (pow(v, -1, c.p) if v else 0), hex(v)
