"""GPU parity tests for k256 through the C ABI (libecgpu.so), against the oracle and the
committed golden fixtures.  Bit-exact: integer arithmetic, canonical bytes."""
import random

import numpy as np
import pytest

from oracle import ecmodel as M
from oracle import synth
from conftest import load_config1

pytestmark = pytest.mark.gpu

C = M.K256
N, P = C.n, C.p


@pytest.fixture(scope="module")
def curve():
    import ecgpu
    ctx = ecgpu.Context(0)
    yield ctx.curve("k256")
    ctx.close()


def fe_bytes(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "big") for v in vals), dtype=np.uint8).reshape(-1, 32).copy()


def to_ints(arr):
    return [int.from_bytes(bytes(r), "big") for r in arr]


def proj_arr(pts):
    return np.frombuffer(b"".join(M.proj_bytes(C, p) for p in pts), dtype=np.uint8).reshape(-1, 96).copy()


def rand_proj(rng, n, start=0):
    out = []
    for i in range(n):
        x, y = synth.point(C, start + i, seed=77)
        z = rng.randrange(1, P)
        out.append((x * z % P, y * z % P, z))
    return out


EDGE = [0, 1, 2, 977, 2**32 + 976, P - 2, P - 1, 2**255, 2**128 - 1, 2**32 - 1, 2**64]


def test_field_ops(curve, ref_vectors):
    import ecgpu
    rng = random.Random(1)
    xs = [a for a in EDGE for _ in EDGE] + [rng.randrange(P) for _ in range(4096)]
    ys = [b for _ in EDGE for b in EDGE] + [rng.randrange(P) for _ in range(4096)]
    a, b = fe_bytes(xs), fe_bytes(ys)
    for op, fn in ((ecgpu.FE_MUL, lambda x, y: x * y), (ecgpu.FE_ADD, lambda x, y: x + y), (ecgpu.FE_SUB, lambda x, y: x - y)):
        got = to_ints(curve.field_op(op, a, b))
        assert got == [fn(x, y) % P for x, y in zip(xs, ys)]
    assert to_ints(curve.field_op(ecgpu.FE_SQR, a)) == [x * x % P for x in xs]
    assert to_ints(curve.field_op(ecgpu.FE_NEG, a)) == [(-x) % P for x in xs]
    inv = to_ints(curve.field_op(ecgpu.FE_INV, a[:600]))
    assert all((x == 0 and g == 0) or g * x % P == 1 for x, g in zip(xs, inv))
    sq = to_ints(curve.field_op(ecgpu.FE_SQRT, a[:600]))
    for x, g in zip(xs, sq):
        want = M.field_sqrt(C, x)
        assert g == (2**256 - 1 if want is None else want)
    # reference KATs (field_8x32_risc0.rs:225-303, test_vectors/field.rs)
    k = {n: int(v, 16) for n, v in ref_vectors["k256"]["field_kat"].items()}
    assert to_ints(curve.field_op(ecgpu.FE_MUL, fe_bytes([k["a"]]), fe_bytes([k["b"]]))) == [k["mul"]]
    assert to_ints(curve.field_op(ecgpu.FE_ADD, fe_bytes([k["a"]]), fe_bytes([k["b"]]))) == [k["add"]]
    assert to_ints(curve.field_op(ecgpu.FE_SQR, fe_bytes([k["a"]]))) == [k["square_a"]]
    assert to_ints(curve.field_op(ecgpu.FE_NEG, fe_bytes([k["a"]]))) == [k["negate_a"]]
    dbl = [int(v, 16) for v in ref_vectors["k256"]["field_dbl"]]
    assert to_ints(curve.field_op(ecgpu.FE_ADD, fe_bytes(dbl[:-1]), fe_bytes(dbl[:-1]))) == dbl[1:]


def test_point_ops_exact_xyz(curve):
    rng = random.Random(2)
    ps = rand_proj(rng, 300) + [M.IDENTITY, C.G, M.point_neg(C, C.G), C.G, M.IDENTITY]
    qs = rand_proj(rng, 300, start=1000) + [C.G, M.IDENTITY, C.G, C.G, M.IDENTITY]
    got = curve.add(proj_arr(ps), proj_arr(qs))
    assert bytes(got) == b"".join(M.proj_bytes(C, M.k256_add(p, q)) for p, q in zip(ps, qs))
    got = curve.double(proj_arr(ps))
    assert bytes(got) == b"".join(M.proj_bytes(C, M.k256_double(p)) for p in ps)
    aff = [M.to_affine(C, q) for q in qs]
    qa = np.frombuffer(b"".join(M.i2b(C, a[0]) + M.i2b(C, a[1]) for a in aff), dtype=np.uint8).reshape(-1, 64).copy()
    got = curve.add_mixed(proj_arr(ps), qa)
    assert bytes(got) == b"".join(M.proj_bytes(C, M.k256_add_mixed(p, a)) for p, a in zip(ps, aff))
    xy, inf = curve.batch_normalize(proj_arr(ps))
    for i, p in enumerate(ps):
        assert bytes(xy[i]) + bytes([inf[i]]) == M.affine_bytes(C, M.to_affine(C, p))


def test_group_vectors(curve, ref_vectors):
    """ADD / MUL vectors of k256/src/test_vectors/group.rs through the GPU."""
    import ecgpu
    vec = ref_vectors["k256"]["group"]["mul"]
    ks = fe_bytes([int(k, 16) for k, _, _ in vec])
    want = b"".join(bytes.fromhex(x + y) for _, x, y in vec)
    for flags in (0, ecgpu.EXACT_REFERENCE):
        xy, inf = curve.mul_by_generator(ks, flags=flags)
        assert bytes(xy) == want and not inf.any()
        g = np.frombuffer(M.i2b(C, C.gx) + M.i2b(C, C.gy), dtype=np.uint8).reshape(1, 64).repeat(len(vec), 0).copy()
        xy, inf = curve.mul(ks, g, flags=flags)
        assert bytes(xy) == want and not inf.any()
    # repeated addition of G (ADD_TEST_VECTORS) by mixed and full additions
    add = ref_vectors["k256"]["group"]["add"]
    acc = proj_arr([M.IDENTITY])
    gp = proj_arr([C.G])
    ga = np.frombuffer(M.i2b(C, C.gx) + M.i2b(C, C.gy), dtype=np.uint8).reshape(1, 64).copy()
    accm = acc.copy()
    for x, y in add:
        acc = curve.add(acc, gp)
        accm = curve.add_mixed(accm, ga)
        for a in (acc, accm):
            xy, inf = curve.batch_normalize(a)
            assert bytes(xy[0]).hex() == (x + y).lower() and inf[0] == 0
    # ECDSA d -> Q, k -> r
    for v in ref_vectors["k256"]["ecdsa"]:
        xy, _ = curve.mul_by_generator(fe_bytes([int(v["d"], 16), int(v["k"], 16)]))
        assert bytes(xy[0]).hex() == v["q_x"] + v["q_y"]
        assert int.from_bytes(bytes(xy[1][:32]), "big") % N == int(v["r"], 16)


def test_hash2curve_add_triples(curve, ref_vectors):
    for v in ref_vectors["k256"]["hash2curve"]:
        q0 = (int(v["q0_x"], 16), int(v["q0_y"], 16), 1)
        q1 = (int(v["q1_x"], 16), int(v["q1_y"], 16), 1)
        xy, inf = curve.batch_normalize(curve.add(proj_arr([q0]), proj_arr([q1])))
        assert bytes(xy[0]).hex() == v["p_x"] + v["p_y"]


def test_config1_fixture_bit_exact(curve):
    """BASELINE.json configs[0]: 1024 seeded (scalar, point) pairs, affine outputs byte for byte."""
    import ecgpu
    fx = load_config1("k256")
    ks = np.frombuffer(b"".join(bytes.fromhex(r[0]) for r in fx["rows"]), dtype=np.uint8).reshape(-1, 32).copy()
    pts = np.frombuffer(b"".join(bytes.fromhex(r[1] + r[2]) for r in fx["rows"]), dtype=np.uint8).reshape(-1, 64).copy()
    want = b"".join(bytes.fromhex(r[3]) for r in fx["rows"])
    for flags in (0, ecgpu.EXACT_REFERENCE):
        xy, inf = curve.mul(ks, pts, flags=flags)
        got = b"".join(bytes(xy[i]) + bytes([inf[i]]) for i in range(len(xy)))
        assert got == want


def test_mul_exact_reference_xyz_and_edges(curve):
    import ecgpu
    rng = random.Random(3)
    ks = [0, 1, 2, N - 1, N - 2, (N - 1) // 2, 2**128 - 1, 2**128 + 1] * 3 + [rng.randrange(N) for _ in range(40)]
    ps = [C.G] * 8 + [M.point_neg(C, C.G)] * 8 + [M.IDENTITY] * 8 + rand_proj(rng, 40, start=5000)
    out = curve.mul(fe_bytes(ks), proj_arr(ps), point_format=ecgpu.PROJECTIVE, out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    for i, (p, k) in enumerate(zip(ps, ks)):
        assert bytes(out[i]) == M.proj_bytes(C, M.k256_mul_ref(p, k)), i
    # default schedule: the group element (affine) must match on the same edge inputs
    xy, inf = curve.mul(fe_bytes(ks), proj_arr(ps), point_format=ecgpu.PROJECTIVE)
    for i, (p, k) in enumerate(zip(ps, ks)):
        want = M.affine_mul(C, k, M.to_affine_opt(C, p))
        assert bytes(xy[i]) + bytes([inf[i]]) == M.affine_bytes(C, (0, 0, 1) if want is None else (want[0], want[1], 0)), i
    # mul_by_generator exact XYZ
    out = curve.mul_by_generator(fe_bytes(ks), out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    for i, k in enumerate(ks):
        assert bytes(out[i]) == M.proj_bytes(C, M.k256_mul_by_generator_ref(k)), i


def test_lincomb_two_terms(curve):
    import ecgpu
    rng = random.Random(4)
    n = 64
    ps = rand_proj(rng, 2 * n, start=9000)
    ks = [rng.randrange(N) for _ in range(2 * n)]
    out = curve.lincomb(fe_bytes(ks), proj_arr(ps), terms=2, point_format=ecgpu.PROJECTIVE, out_format=ecgpu.PROJECTIVE,
                        flags=ecgpu.EXACT_REFERENCE)
    for i in range(n):
        want = M.k256_lincomb_ref([(ps[2 * i], ks[2 * i]), (ps[2 * i + 1], ks[2 * i + 1])])
        assert bytes(out[i]) == M.proj_bytes(C, want)
    xy, inf = curve.lincomb(fe_bytes(ks), proj_arr(ps), terms=2, point_format=ecgpu.PROJECTIVE)
    for i in range(n):
        a = M.affine_add(C, M.affine_mul(C, ks[2 * i], M.to_affine_opt(C, ps[2 * i])),
                         M.affine_mul(C, ks[2 * i + 1], M.to_affine_opt(C, ps[2 * i + 1])))
        assert bytes(xy[i]) == M.i2b(C, a[0]) + M.i2b(C, a[1])
    # throughput schedule on the awkward cases: equal points (the doubling branch of the mixed addition),
    # opposite points, identity inputs, zero scalars
    G = (C.gx, C.gy)
    Q = synth.point(C, 1, seed=99)
    cases = [(5, G, 5, G), (7, G, 7, M.affine_neg(C, G)), (3, Q, N - 3, Q), (11, None, 13, Q), (17, Q, 19, None), (0, G, 0, Q),
             (1, G, 1, Q), (N - 1, G, 1, G), (2**128, Q, 2**128 + 1, G), (rng.randrange(N), Q, rng.randrange(N), Q)]
    kk, pp = [], []
    for k0, p0, k1, p1 in cases:
        kk += [k0, k1]
        pp += [bytes(64) if p0 is None else M.i2b(C, p0[0]) + M.i2b(C, p0[1]), bytes(64) if p1 is None else M.i2b(C, p1[0]) + M.i2b(C, p1[1])]
    xy, inf = curve.lincomb(fe_bytes(kk), np.frombuffer(b"".join(pp), dtype=np.uint8).reshape(-1, 64).copy(), terms=2)
    for i, (k0, p0, k1, p1) in enumerate(cases):
        a = M.affine_add(C, M.affine_mul(C, k0, p0), M.affine_mul(C, k1, p1))
        assert bytes(xy[i]) + bytes([inf[i]]) == M.affine_bytes(C, (0, 0, 1) if a is None else (a[0], a[1], 0)), i


def test_validate_and_decompress(curve):
    rng = random.Random(5)
    sc = [0, 1, N - 1, N, N + 1, 2**256 - 1] + [rng.randrange(2**256) for _ in range(50)]
    ok = curve.validate_scalars(fe_bytes(sc))
    assert list(ok) == [1 if s < N else 0 for s in sc]
    pts = [synth.point(C, i, seed=5) for i in range(20)]
    bad = [(x, (y + 1) % P) for x, y in pts[:5]] + [(P, 1), (1, P + 5), (0, 0)]
    allp = pts + bad
    arr = np.frombuffer(b"".join(M.i2b(C, x) + M.i2b(C, y) for x, y in allp), dtype=np.uint8).reshape(-1, 64).copy()
    assert list(curve.validate_points(arr)) == [1] * 20 + [0] * 7 + [1]
    xs = [p[0] for p in pts] + [rng.randrange(P) for _ in range(40)] + [P, P + 1]
    odd = [p[1] & 1 for p in pts] + [rng.randrange(2) for _ in range(42)]
    out, okd = curve.decompress(fe_bytes(xs), np.array(odd, dtype=np.uint8))
    for i, (x, o) in enumerate(zip(xs, odd)):
        want = M.decompress(C, x, o)
        if want is None:
            assert okd[i] == 0 and not out[i].any()
        else:
            assert okd[i] == 1 and bytes(out[i]) == M.i2b(C, want[0]) + M.i2b(C, want[1])


def test_synth_streams_match_spec_and_device_pointers(curve):
    """Device generators reproduce oracle/synth.py; the ABI accepts device pointers (torch tensors)."""
    import torch
    n = 512
    first = 12345
    d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    curve.synth_scalars_device(d_s, n, synth.SEED, first)
    curve.synth_points_device(d_p, n, synth.SEED, first)
    curve.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i)
    curve.ctx.synchronize()
    s, p, o = d_s.cpu().numpy(), d_p.cpu().numpy(), d_o.cpu().numpy()
    for i in range(0, n, 7):
        k = synth.scalar(C, first + i)
        pt = synth.point(C, first + i)
        assert bytes(s[i]) == M.i2b(C, k)
        assert bytes(p[i]) == M.i2b(C, pt[0]) + M.i2b(C, pt[1])
        w = M.affine_mul(C, k, pt)
        assert bytes(o[i]) == M.i2b(C, w[0]) + M.i2b(C, w[1])
    assert not d_i.cpu().numpy().any()


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_lincomb_many_terms(cn):
    """lincomb_ext over arrays longer than two (k256 mul.rs:325-340): 3, 5 and 16 terms per combination against the
    model, including zero scalars, identity points and cancelling terms."""
    import random
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cn)
    c = M.CURVES[cn]
    nb = c.nbytes
    rng = random.Random(77)
    for terms in (3, 5, 16):
        n = 40
        ks = [[rng.randrange(c.n) for _ in range(terms)] for _ in range(n)]
        ps = [[M.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy)) for _ in range(terms)] for _ in range(n)]
        ks[0][1] = 0
        ps[1][2] = None
        ps[2][1], ks[2][1] = ps[2][0], (c.n - ks[2][0]) % c.n            # terms 0 and 1 cancel
        if terms == 3:
            ks[2][2] = 0                                                  # whole combination is the identity
        sb = b"".join(M.i2b(c, k) for row in ks for k in row)
        pb = b"".join(bytes(2 * nb) if P is None else M.i2b(c, P[0]) + M.i2b(c, P[1]) for row in ps for P in row)
        out, inf = cv.lincomb(sb, pb, terms=terms)
        for i in range(n):
            acc = None
            for k, P in zip(ks[i], ps[i]):
                acc = M.affine_add(c, acc, M.affine_mul(c, k, P) if P is not None else None)
            if acc is None:
                assert inf[i] == 1 and not bytes(out[i]).strip(b"\0")
            else:
                assert inf[i] == 0 and bytes(out[i]) == M.i2b(c, acc[0]) + M.i2b(c, acc[1]), (terms, i)
        # the term-by-term form of rounds 1-3 (one reference multiplication per term) is an independent path to the same bytes
        ctx.set_option(ecgpu.OPT_LINCOMB_TERM_BY_TERM, 1)
        out2, inf2 = cv.lincomb(sb, pb, terms=terms)
        ctx.set_option(ecgpu.OPT_LINCOMB_TERM_BY_TERM, 0)
        assert bytes(out2) == bytes(out) and bytes(inf2) == bytes(inf)
    if cn == "k256":
        xyz = cv.lincomb(sb, pb, terms=16, out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
        for i in (0, 1, 2, 7):
            want = M.k256_lincomb_ref([((P[0], P[1], 1) if P is not None else M.IDENTITY, k) for k, P in zip(ks[i], ps[i])])
            assert bytes(xyz[i]) == M.proj_bytes(c, want), i
    else:
        with pytest.raises(ecgpu.EcgpuError):
            cv.lincomb(sb, pb, terms=16, flags=ecgpu.EXACT_REFERENCE)
    ctx.close()


@pytest.mark.parametrize("terms", [3, 5, 16])
def test_k256_lincomb_ext_exact_xyz(terms):
    """LinearCombinationExt over a slice (k256 mul.rs:325-340) with ECGPU_EXACT_REFERENCE: the reference's interleaved schedule for a
    run-time number of terms returns the very (X, Y, Z) of the model's restatement of mul.rs:342-393 - projective inputs (Z != 1),
    identities, zero scalars, cancelling terms; 300 combinations so that several lanes and the grid stride take part."""
    import random
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve("k256")
    c = M.K256
    rng = random.Random(900 + terms)
    n = 300
    base = [M.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy)) for _ in range(9)]
    ks, pts = [], []
    for i in range(n * terms):
        ks.append(rng.randrange(c.n))
        P = base[rng.randrange(9)]
        z = rng.randrange(1, c.p)
        pts.append((P[0] * z % c.p, P[1] * z % c.p, z))
    ks[0] = 0
    pts[1] = M.IDENTITY
    pts[terms + 1], ks[terms + 1] = pts[terms], (c.n - ks[terms]) % c.n
    sb = b"".join(k.to_bytes(32, "big") for k in ks)
    pb = b"".join(M.proj_bytes(c, P) for P in pts)
    xyz = cv.lincomb(sb, pb, terms=terms, point_format=ecgpu.PROJECTIVE, out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    for i in list(range(6)) + [n // 2, n - 1]:
        want = M.k256_lincomb_ref(list(zip(pts[i * terms:(i + 1) * terms], ks[i * terms:(i + 1) * terms])))
        assert bytes(xyz[i]) == M.proj_bytes(c, want), (terms, i)
    # the affine form of the same call equals the throughput schedule's bytes
    a1, i1 = cv.lincomb(sb, pb, terms=terms, point_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    a2, i2 = cv.lincomb(sb, pb, terms=terms, point_format=ecgpu.PROJECTIVE)
    assert bytes(a1) == bytes(a2) and bytes(i1) == bytes(i2)
    ctx.close()


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
@pytest.mark.parametrize("terms,n", [(1024, 3), (100, 40), (17, 5000), (7, 70001)])
def test_lincomb_many_terms_shapes(cn, terms, n):
    """Shapes of the shared-doubling schedule (csrc/straus.hpp): 1024 terms (the maximum: groups of 4 per work item on an idle chip),
    100 and 17 terms (several balanced groups per combination, the second stage adds them), 7 terms x 70 001 (two combinations per
    pass, ragged) - against the term-by-term path on all combinations and the C oracle on a sample."""
    import ecgpu
    from oracle import coracle as CO
    from oracle import synth
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cn)
    cid = cv.id
    nb = cv.nb
    if cid == 2 and terms * n > 200000:
        n = 200000 // terms
    s = CO.synth_scalars(cid, n * terms, synth.SEED, 61_000_000)
    p = CO.synth_points(cid, n * terms, synth.SEED, 61_000_000)
    s[3] = 0
    p[terms + 2] = 0
    out, inf = cv.lincomb(s, p, terms=terms)
    ctx.set_option(ecgpu.OPT_LINCOMB_TERM_BY_TERM, 1)
    out2, inf2 = cv.lincomb(s, p, terms=terms)
    ctx.set_option(ecgpu.OPT_LINCOMB_TERM_BY_TERM, 0)
    assert bytes(out) == bytes(out2) and bytes(inf) == bytes(inf2)
    for i in (0, 1, n - 1):                       # the oracle: term-by-term products summed with the model's affine addition
        c = M.CURVES[cn]
        prod = CO.lincomb_batch(cid, s[i * terms:(i + 1) * terms], p[i * terms:(i + 1) * terms], threads=8)
        acc = None
        for r in prod:
            if r[-1]:
                continue
            acc = M.affine_add(c, acc, (int.from_bytes(bytes(r[:nb]), "big"), int.from_bytes(bytes(r[nb:2 * nb]), "big")))
        assert bytes(out[i]) == M.i2b(c, acc[0]) + M.i2b(c, acc[1]) and inf[i] == 0, (terms, i)
    ctx.close()


def test_diffie_hellman_agreement(curve):
    """Both sides of an ECDH exchange derive the same shared secret; value checked against the model."""
    rng = random.Random(25519)
    n = 64
    a = [rng.randrange(1, M.K256.n) for _ in range(n)]
    b = [rng.randrange(1, M.K256.n) for _ in range(n)]
    tob = lambda v: v.to_bytes(32, "big")
    pa, _ = curve.mul_by_generator(b"".join(map(tob, a)))
    pb, _ = curve.mul_by_generator(b"".join(map(tob, b)))
    s1 = curve.diffie_hellman(b"".join(map(tob, a)), pb)
    s2 = curve.diffie_hellman(b"".join(map(tob, b)), pa)
    assert bytes(s1) == bytes(s2)
    want = M.affine_mul(M.K256, a[0] * b[0] % M.K256.n, (M.K256.gx, M.K256.gy))[0]
    assert bytes(s1[0]) == tob(want)
