"""GPU parity tests for P-256 / P-384 through the C ABI against the oracle and the golden fixtures
(the reference's own test macros: primeorder/src/dev.rs:66-155 via p256|p384/tests/projective.rs)."""
import random

import numpy as np
import pytest

from oracle import coracle as CO
from oracle import ecmodel as M
from oracle import synth
from conftest import load_config1

pytestmark = pytest.mark.gpu
CURVES = ["p256", "p384"]


@pytest.fixture(scope="module")
def ctx():
    import ecgpu
    c = ecgpu.Context(0)
    yield c
    c.close()


def arr(rows, w):
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(-1, w).copy()


def fe_arr(c, vals):
    return arr([int(v).to_bytes(c.nbytes, "big") for v in vals], c.nbytes)


def ints(a):
    return [int.from_bytes(bytes(r), "big") for r in a]


def rand_proj(c, rng, n, start=0):
    out = []
    for i in range(n):
        x, y = synth.point(c, start + i, seed=91)
        z = rng.randrange(1, c.p)
        out.append((x * z % c.p, y * z % c.p, z))
    return out


@pytest.mark.parametrize("cn", CURVES)
def test_field_ops(ctx, cn, ref_vectors):
    import ecgpu
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    p = c.p
    rng = random.Random(51)
    edge = [0, 1, 2, 3, p - 1, p - 2, 2**32 - 1, 2**32, 2**96, (p - 1) // 2, p >> 1]
    xs = [a for a in edge for _ in edge] + [rng.randrange(p) for _ in range(2048)]
    ys = [b for _ in edge for b in edge] + [rng.randrange(p) for _ in range(2048)]
    a, b = fe_arr(c, xs), fe_arr(c, ys)
    assert ints(cv.field_op(ecgpu.FE_MUL, a, b)) == [x * y % p for x, y in zip(xs, ys)]
    assert ints(cv.field_op(ecgpu.FE_ADD, a, b)) == [(x + y) % p for x, y in zip(xs, ys)]
    assert ints(cv.field_op(ecgpu.FE_SUB, a, b)) == [(x - y) % p for x, y in zip(xs, ys)]
    assert ints(cv.field_op(ecgpu.FE_SQR, a)) == [x * x % p for x in xs]
    assert ints(cv.field_op(ecgpu.FE_NEG, a)) == [(-x) % p for x in xs]
    inv = ints(cv.field_op(ecgpu.FE_INV, a[:300]))
    assert all((x == 0 and g == 0) or g * x % p == 1 for x, g in zip(xs, inv))
    sq = ints(cv.field_op(ecgpu.FE_SQRT, a[:300]))
    for x, g in zip(xs, sq):
        want = M.field_sqrt(c, x)
        assert g == ((1 << (8 * c.nbytes)) - 1 if want is None else want)
    if cn == "p256":
        dbl = [int(v, 16) for v in ref_vectors["p256"]["field_dbl"]]
        assert ints(cv.field_op(ecgpu.FE_ADD, fe_arr(c, dbl[:-1]), fe_arr(c, dbl[:-1]))) == dbl[1:]


@pytest.mark.parametrize("cn", CURVES)
def test_point_ops_exact_xyz(ctx, cn):
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    rng = random.Random(52)
    ps = rand_proj(c, rng, 200) + [M.IDENTITY, c.G, M.point_neg(c, c.G), c.G, M.IDENTITY]
    qs = rand_proj(c, rng, 200, 1000) + [c.G, M.IDENTITY, c.G, c.G, M.IDENTITY]
    w = 3 * c.nbytes
    pa, qa = arr([M.proj_bytes(c, p) for p in ps], w), arr([M.proj_bytes(c, q) for q in qs], w)
    assert bytes(cv.add(pa, qa)) == b"".join(M.proj_bytes(c, M.am3_add(c, p, q)) for p, q in zip(ps, qs))
    assert bytes(cv.double(pa)) == b"".join(M.proj_bytes(c, M.am3_double(c, p)) for p in ps)
    aff = [M.to_affine(c, q) for q in qs]
    qaff = arr([M.i2b(c, a[0]) + M.i2b(c, a[1]) for a in aff], 2 * c.nbytes)
    assert bytes(cv.add_mixed(pa, qaff)) == b"".join(M.proj_bytes(c, M.am3_add_mixed(c, p, a)) for p, a in zip(ps, aff))
    xy, inf = cv.batch_normalize(pa)
    for i, p in enumerate(ps):
        assert bytes(xy[i]) + bytes([inf[i]]) == M.affine_bytes(c, M.to_affine(c, p))


@pytest.mark.parametrize("cn", CURVES)
def test_reference_vectors(ctx, cn, ref_vectors):
    """group.rs ADD/MUL vectors, ecdsa.rs d->Q / k->r, hash2curve Q0+Q1=P."""
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    nb = c.nbytes
    vec = ref_vectors[cn]["group"]["mul"]
    ks = arr([bytes.fromhex(k) for k, _, _ in vec], nb)
    xy, inf = cv.mul_by_generator(ks)
    assert bytes(xy) == b"".join(bytes.fromhex(x + y) for _, x, y in vec) and not inf.any()
    acc = arr([M.proj_bytes(c, M.IDENTITY)], 3 * nb)
    gp = arr([M.proj_bytes(c, c.G)], 3 * nb)
    ga = arr([M.i2b(c, c.gx) + M.i2b(c, c.gy)], 2 * nb)
    accm = acc.copy()
    for x, y in ref_vectors[cn]["group"]["add"]:
        acc = cv.add(acc, gp)
        accm = cv.add_mixed(accm, ga)
        for a in (acc, accm):
            o, i = cv.batch_normalize(a)
            assert bytes(o[0]).hex() == (x + y).lower() and i[0] == 0
    ev = ref_vectors[cn]["ecdsa"]
    xy, _ = cv.mul_by_generator(arr([bytes.fromhex(v["d"]) for v in ev] + [bytes.fromhex(v["k"]) for v in ev], nb))
    for i, v in enumerate(ev):
        assert bytes(xy[i]).hex() == v["q_x"] + v["q_y"]
        assert int.from_bytes(bytes(xy[len(ev) + i][:nb]), "big") % c.n == int(v["r"], 16)
    for v in ref_vectors[cn]["hash2curve"]:
        q0 = (int(v["q0_x"], 16), int(v["q0_y"], 16), 1)
        q1 = (int(v["q1_x"], 16), int(v["q1_y"], 16), 1)
        o, _ = cv.batch_normalize(cv.add(arr([M.proj_bytes(c, q0)], 3 * nb), arr([M.proj_bytes(c, q1)], 3 * nb)))
        assert bytes(o[0]).hex() == v["p_x"] + v["p_y"]


@pytest.mark.parametrize("cn", CURVES)
def test_config1_fixture_and_exact_xyz(ctx, cn):
    import ecgpu
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    nb = c.nbytes
    fx = load_config1(cn)
    ks = arr([bytes.fromhex(r[0]) for r in fx["rows"]], nb)
    pts = arr([bytes.fromhex(r[1] + r[2]) for r in fx["rows"]], 2 * nb)
    xy, inf = cv.mul(ks, pts)
    assert b"".join(bytes(xy[i]) + bytes([inf[i]]) for i in range(len(xy))) == b"".join(bytes.fromhex(r[3]) for r in fx["rows"])
    rng = random.Random(53)
    sc = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2] * 2 + [rng.randrange(c.n) for _ in range(20)]
    ps = [c.G] * 6 + [M.IDENTITY] * 3 + [M.point_neg(c, c.G)] * 3 + rand_proj(c, rng, 20, 3000)
    out = cv.mul(fe_arr(c, sc), arr([M.proj_bytes(c, p) for p in ps], 3 * nb), point_format=ecgpu.PROJECTIVE,
                 out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    for i, (p, k) in enumerate(zip(ps, sc)):
        assert bytes(out[i]) == M.proj_bytes(c, M.primeorder_mul_ref(c, p, k)), i
    out = cv.mul_by_generator(fe_arr(c, sc), out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    for i, k in enumerate(sc):
        assert bytes(out[i]) == M.proj_bytes(c, M.primeorder_mul_ref(c, c.G, k)), i
    # lincomb(x, k, y, l) = x*k + y*l  (primeorder/src/projective.rs:415-420)
    n = 8
    lp = rand_proj(c, rng, 2 * n, 4000)
    lk = [rng.randrange(c.n) for _ in range(2 * n)]
    out = cv.lincomb(fe_arr(c, lk), arr([M.proj_bytes(c, p) for p in lp], 3 * nb), terms=2, point_format=ecgpu.PROJECTIVE,
                     out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    for i in range(n):
        want = M.lincomb_ref(c, [(lp[2 * i], lk[2 * i]), (lp[2 * i + 1], lk[2 * i + 1])])
        assert bytes(out[i]) == M.proj_bytes(c, want)


@pytest.mark.parametrize("cn,cid", [("p256", 1), ("p384", 2)])
def test_larger_batch_against_c_oracle_and_synth(ctx, cn, cid):
    """2^12 seeded units generated on the device, multiplied on the device, checked against the C oracle."""
    import torch
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    nb = c.nbytes
    n, first = 4096, 777
    d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, first)
    cv.synth_points_device(d_p, n, synth.SEED, first)
    cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i)
    ctx.synchronize()
    s, p = d_s.cpu().numpy(), d_p.cpu().numpy()
    assert bytes(s) == bytes(CO.synth_scalars(cid, n, synth.SEED, first))
    assert bytes(p) == bytes(CO.synth_points(cid, n, synth.SEED, first))
    want = CO.lincomb_batch(cid, s, p, threads=4)
    got = np.concatenate([d_o.cpu().numpy(), d_i.cpu().numpy()[:, None]], axis=1)
    assert bytes(got) == bytes(want)


@pytest.mark.parametrize("cn", CURVES)
def test_validate_and_decompress(ctx, cn):
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    nb = c.nbytes
    rng = random.Random(54)
    top = 1 << (8 * nb)
    sc = [0, 1, c.n - 1, c.n, c.n + 1, top - 1] + [rng.randrange(top) for _ in range(40)]
    assert list(cv.validate_scalars(fe_arr(c, sc))) == [1 if s < c.n else 0 for s in sc]
    pts = [synth.point(c, i, seed=6) for i in range(16)]
    bad = [(x, (y + 1) % c.p) for x, y in pts[:4]] + [(c.p, 1), (0, 0)]
    allp = pts + bad
    ok = cv.validate_points(arr([int(x).to_bytes(nb, "big") + int(y).to_bytes(nb, "big") for x, y in allp], 2 * nb))
    assert list(ok) == [1] * 16 + [0] * 5 + [1]
    xs = [p[0] for p in pts] + [rng.randrange(c.p) for _ in range(30)] + [c.p]
    odd = [p[1] & 1 for p in pts] + [rng.randrange(2) for _ in range(31)]
    out, okd = cv.decompress(fe_arr(c, xs), np.array(odd, dtype=np.uint8))
    for i, (x, o) in enumerate(zip(xs, odd)):
        want = M.decompress(c, x, o)
        if want is None:
            assert okd[i] == 0 and not out[i].any()
        else:
            assert okd[i] == 1 and bytes(out[i]) == M.i2b(c, want[0]) + M.i2b(c, want[1])


@pytest.mark.parametrize("lg", [18, 21])
@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_fixed_base_wide_table_path(ctx, cn, cid, lg):
    """Batches of 2^18 and more take the 16-bit-window table, 2^21 and more the 20-bit-window table (fixedbase.hpp):
    edge scalars up front (digit boundaries of both recodings), a seeded batch behind them, sampled against the C
    oracle (reference mul_by_generator)."""
    import torch
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    nb = c.nbytes
    n = 1 << lg
    bits = 8 * nb
    edge = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, (c.n + 1) // 2, 0x8000, 0x7FFF, 0xFFFF, 0x10000, 0x80008000, (1 << (8 * nb - 1)) % c.n,
            (0x7FFF << (8 * nb - 16)) | ((1 << (8 * nb - 16)) - 1), c.n - 0x8000, c.n - 0x7FFF, int("8000" * (nb // 2), 16) % c.n,
            int("7FFF" * (nb // 2), 16), int("FFFF" * (nb // 2), 16) % c.n,
            # 20-bit windows: digits 2^19 - 1, 2^19, 2^20 - 1 in every window, carries rippling through all windows
            0x7FFFF, 0x80000, 0xFFFFF, 0x100000, sum(0x7FFFF << (20 * j) for j in range(bits // 20)) % c.n,
            sum(0x80000 << (20 * j) for j in range(bits // 20)) % c.n, sum(0xFFFFF << (20 * j) for j in range(bits // 20)) % c.n,
            ((c.n - 1) // 2) - 0x80000, (1 << (bits - 2)) - 1, (1 << (bits - 2))]
    d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, 424242)
    ctx.synchronize()
    s = d_s.cpu().numpy()
    for i, k in enumerate(edge):
        s[i] = np.frombuffer(int(k).to_bytes(nb, "big"), dtype=np.uint8)
    d_s.copy_(torch.from_numpy(s))
    d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.mul_device(d_s, None, d_o, n, d_out_inf=d_i)
    ctx.synchronize()
    o, inf = d_o.cpu().numpy(), d_i.cpu().numpy()
    idx = list(range(len(edge))) + list(range(len(edge), n, n // 300))
    want = CO.lincomb_batch(cid, s[idx].copy(), None, threads=4)
    got = np.concatenate([o[idx], inf[idx][:, None]], axis=1)
    assert bytes(got) == bytes(want)


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_group_encoding_to_from_bytes(ctx, cn, ref_vectors):
    """GroupEncoding::{to_bytes, from_bytes} against the model: reference base-point vectors, random points (affine and
    projective input), the identity, compact tag, bad tags, x >= p, x without a root."""
    import random
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    nb = c.nbytes
    rng = random.Random(99)
    pts = [None, (c.gx, c.gy)] + [M.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy)) for _ in range(40)]
    aff = b"".join(bytes(2 * nb) if P is None else M.i2b(c, P[0]) + M.i2b(c, P[1]) for P in pts)
    want = [M.group_to_bytes(c, P) for P in pts]
    got = cv.to_bytes(aff)
    assert [bytes(g) for g in got] == want
    assert want[1].hex() == ref_vectors[cn]["encoding"]["compressed_basepoint"]
    proj = []
    for P in pts:
        if P is None:
            proj.append(M.proj_bytes(c, M.IDENTITY))
        else:
            z = rng.randrange(1, c.p)
            proj.append(M.proj_bytes(c, (P[0] * z % c.p, P[1] * z % c.p, z)))
    got = cv.to_bytes(b"".join(proj), point_format=1)
    assert [bytes(g) for g in got] == want
    enc = list(want)
    enc += [bytes([5]) + w[1:] for w in want[1:6]]                       # compact tag
    enc += [bytes([4]) + want[1][1:], bytes([1]) + want[1][1:], bytes([0]) + want[1][1:], bytes([6]) + bytes(nb)]
    enc += [bytes([2]) + M.i2b(c, c.p), bytes([3]) + (c.p + 5).to_bytes(nb, "big"), bytes([2]) + b"\xff" * nb]
    x = 1
    while M.decompress(c, x, 0) is not None:
        x += 1
    enc += [bytes([2]) + M.i2b(c, x)]                                    # no square root
    if "compact_basepoint" in ref_vectors[cn]["encoding"]:
        enc += [bytes.fromhex(ref_vectors[cn]["encoding"]["compact_basepoint"])]
    out, ok = cv.from_bytes(b"".join(enc))
    for i, e in enumerate(enc):
        w_ok, P = M.group_from_bytes(c, e)
        assert bool(ok[i]) == w_ok, (i, e.hex())
        exp = bytes(2 * nb) if (P is None) else M.i2b(c, P[0]) + M.i2b(c, P[1])
        assert bytes(out[i]) == exp, (i, e.hex())


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_hash_to_curve(ctx, cn, ref_vectors):
    """map_to_curve and hash_from_bytes: the RFC 9380 vectors of <curve>/src/arithmetic/hash2curve.rs (u -> Q0, Q1,
    msg -> P), then random and edge field elements (0, 1, p - 1, values making the first candidate a non-square)
    against the model."""
    import random
    from ecgpu import hash2curve
    c = M.CURVES[cn]
    cv = ctx.curve(cn)
    vs = ref_vectors[cn]["hash2curve"]
    dst = vs[0]["dst"].encode()
    u = b"".join(bytes.fromhex(v["u_0"]) + bytes.fromhex(v["u_1"]) for v in vs)
    q, inf = cv.map_to_curve(u, count=1)
    assert not inf.any()
    for i, v in enumerate(vs):
        assert bytes(q[2 * i]).hex() == v["q0_x"] + v["q0_y"] and bytes(q[2 * i + 1]).hex() == v["q1_x"] + v["q1_y"]
    pts, inf = hash2curve.hash_from_bytes(cv, [v["msg"].encode() for v in vs], dst)
    assert [bytes(p).hex() for p in pts] == [v["p_x"] + v["p_y"] for v in vs] and not inf.any()
    assert hash2curve.hash_to_field(cv, b"abc", dst).hex() == vs[1]["u_0"] + vs[1]["u_1"]
    rng = random.Random(9380)
    us = [0, 1, 2, c.p - 1, c.p - 2] + [rng.randrange(c.p) for _ in range(200)]
    got, _ = cv.map_to_curve(b"".join(M.i2b(c, x) for x in us), count=1)
    for x, g in zip(us, got):
        Q = M.map_to_curve(c, x)
        assert M.on_curve(c, Q) and bytes(g) == M.i2b(c, Q[0]) + M.i2b(c, Q[1]), x
    pairs = [(us[i], us[i + 1]) for i in range(0, 100, 2)] + [(7, 7), (9, c.p - 9)]      # equal points; u and -u map to P and -P
    got, inf = cv.map_to_curve(b"".join(M.i2b(c, a) + M.i2b(c, b) for a, b in pairs), count=2)
    for (a, b), g, f in zip(pairs, got, inf):
        S = M.affine_add(c, M.map_to_curve(c, a), M.map_to_curve(c, b))
        if S is None:
            assert f == 1 and not bytes(g).strip(b"\0")
        else:
            assert f == 0 and bytes(g) == M.i2b(c, S[0]) + M.i2b(c, S[1])
    assert inf[-1] == 1


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_fixed_base_wide_tables_ragged_sizes(ctx, cn, cid):
    """The wide fixed-base kernels keep up to 64 results per lane for one shared inversion: batch sizes that are not
    multiples of anything (2^18 + 333 -> 16-bit windows, 2^21 + 777 -> 20-bit windows), head and tail against the C oracle."""
    import torch
    import ecgpu
    cv = ctx.curve(cn)
    nb = cv.nb
    for n in ((1 << 18) + 333, (1 << 21) + 777):
        d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
        d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
        d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
        cv.synth_scalars_device(d_s, n, synth.SEED, 99 + n)
        idx = np.concatenate([np.arange(0, 200), np.arange(n - 200, n), np.arange(1000, n, n // 97)])
        # identity results in every slot of a lane's shared inversion, not only the first (a lane's results are strided by the
        # grid size): zero scalars and n itself at sampled positions across the whole batch
        ctx.synchronize()
        s_all = d_s.cpu().numpy()
        c = M.CURVES[cn]
        for q, i in enumerate(idx[5::7]):
            s_all[i] = 0 if q % 2 == 0 else np.frombuffer(int(c.n).to_bytes(nb, "big"), dtype=np.uint8)
        d_s.copy_(torch.from_numpy(s_all))
        cv.mul_device(d_s, None, d_o, n, d_out_inf=d_i)
        ctx.synchronize()
        s = s_all[idx].copy()
        want = CO.lincomb_batch(cid, s, None, threads=4)
        got = np.concatenate([d_o.cpu().numpy()[idx], d_i.cpu().numpy()[idx][:, None]], axis=1)
        assert bytes(got) == bytes(want), n
        assert int(d_i.cpu().numpy()[idx[5::7]].min()) == 1            # the planted scalars gave the identity
        # the same batch with projective output: (x : y : 1), identity (0 : 1 : 0)
        d_p = torch.empty((n, 3 * nb), dtype=torch.uint8, device="cuda")
        cv.mul_device(d_s, None, d_p, n, out_format=ecgpu.PROJECTIVE)
        ctx.synchronize()
        pr = d_p.cpu().numpy()[idx]
        one = np.zeros(nb, dtype=np.uint8); one[-1] = 1
        for j in range(len(idx)):
            if want[j, 2 * nb]:
                assert not pr[j, :nb].any() and bytes(pr[j, nb:2 * nb]) == bytes(one) and not pr[j, 2 * nb:].any()
            else:
                assert bytes(pr[j, :2 * nb]) == bytes(want[j, :2 * nb]) and bytes(pr[j, 2 * nb:]) == bytes(one)


@pytest.mark.parametrize("cn,cid", [("p256", 1), ("p384", 2)])
def test_two_term_lincomb_throughput_schedule(ctx, cn, cid):
    """LinearCombination::lincomb (k P + l Q, primeorder/src/projective.rs:415-420) on the throughput schedule: the two
    terms share the doublings (varbase_lane.hpp, NT = 2).  4 097 units - several slots per lane only above 2^18, covered by
    the host twin - with edge cases, against the C oracle's two reference multiplications and complete addition."""
    import ecgpu
    c = M.CURVES[cn]
    nb = c.nbytes
    cv = ctx.curve(cn)
    n = 4097
    s = CO.synth_scalars(cid, 2 * n, synth.SEED, 321)
    p = CO.synth_points(cid, 2 * n, synth.SEED, 321)
    s[4] = 0                      # a zero scalar
    p[7] = 0                      # an identity point
    p[11] = p[10]; s[11] = s[10]  # k P + k P: the doubling branch of the addition
    p[13] = p[12]; s[13] = np.frombuffer(M.i2b(c, (c.n - int.from_bytes(bytes(s[12]), "big")) % c.n), dtype=np.uint8)   # k P - k P = identity
    p[14] = 0; p[15] = 0          # both terms identity
    out, inf = cv.lincomb(s, p, terms=2)
    want = CO.lincomb_batch(cid, s, p, terms=2, threads=8)
    assert bytes(np.concatenate([out, inf[:, None]], axis=1)) == bytes(want)
    assert inf[6] == 1 and inf[7] == 1 and inf.sum() == 2
    # and the exact-reference flag still gives the reference's own (X, Y, Z)
    ref = cv.lincomb(s[:64], p[:64], terms=2, out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    xyz = cv.lincomb(s[:64], p[:64], terms=2, out_format=ecgpu.PROJECTIVE)
    assert cv.point_eq(ref, xyz).all()
