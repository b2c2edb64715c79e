"""Host build of the device templates (tests/hosttwin), two properties that need no GPU:

* the per-lane body of the P-256 / P-384 variable-base kernel (csrc/varbase_lane.hpp) walked with a handful of
  lanes, so that every slot count 1..BATCH, several passes, the shared table inversion and the batched output are
  compared with the big-integer model (the kernel itself only ever reaches those paths at >= 2^19 units);
* the reference-schedule multiplications read their tables by the reference's constant-time scan
  (k256/src/arithmetic/mul.rs:92-127, primeorder/src/projective.rs:132-137): the sequence of table entries touched
  is the same for every scalar, while the throughput schedule's is not.
"""
import ctypes
import random

import pytest

from oracle import ecmodel as M
from oracle import synth
from hosttwin_util import lib, buf, outbuf

CURVES = [("p256", 1), ("p384", 2)]


def _vb(cid, c, ks, ps, lanes, out_fmt=0, proj_in=False):
    nb = c.nbytes
    n = len(ks)
    sb = b"".join(int(k).to_bytes(nb, "big") for k in ks)
    if proj_in:
        pb = b"".join(M.proj_bytes(c, p) for p in ps)
    else:
        pb = b"".join(M.i2b(c, p[0]) + M.i2b(c, p[1]) if p is not None else bytes(2 * nb) for p in ps)
    out = outbuf((3 if out_fmt else 2) * nb * n)
    inf = outbuf(n)
    L = lib()
    L.ht_vb_mul.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                            ctypes.c_size_t, ctypes.c_size_t]
    assert L.ht_vb_mul(cid, buf(sb), buf(pb), 1 if proj_in else 0, out, out_fmt, inf, n, lanes) == 0
    return bytes(out), bytes(inf)


@pytest.mark.parametrize("cn,cid", CURVES)
@pytest.mark.parametrize("lanes,n", [(3, 53), (1, 11), (5, 40), (2, 1)])
def test_vb_lane_walk_matches_model(cn, cid, lanes, n):
    """lanes * 8 units per pass: (3, 53) is two full passes and a ragged third (slot counts 2 and 1), (1, 11) one lane
    with 8 then 3 slots, (5, 40) exactly one pass, (2, 1) a single unit."""
    c = M.CURVES[cn]
    nb = c.nbytes
    rng = random.Random(1000 * cid + lanes)
    ks = [synth.scalar(c, 700 + i) for i in range(n)]
    ps = [synth.point(c, 700 + i) for i in range(n)]
    # edge cases in slots b > 0 and in the later passes
    edge = {lanes * 2: 0, lanes * 3 + (1 % lanes): 1, lanes * 5: c.n - 1, lanes * 8 + 2 * lanes: (c.n - 1) // 2, lanes * 9: (c.n + 1) // 2,
            lanes * 7: c.n, lanes * 6 + (lanes - 1): c.n + 5}          # >= n: reduced once (Reduce<U256>::reduce)
    for i, k in edge.items():
        if i < n:
            ks[i] = k
    ident = [lanes * 4, lanes * 8 + lanes + (2 % lanes)]
    for i in ident:
        if i < n:
            ps[i] = None
    out, inf = _vb(cid, c, ks, ps, lanes)
    for i in range(n):
        want = None if ps[i] is None else M.affine_mul(c, ks[i] % c.n, ps[i])
        got = out[2 * nb * i:2 * nb * (i + 1)]
        if want is None:
            assert got == bytes(2 * nb) and inf[i] == 1, i
        else:
            assert got == M.i2b(c, want[0]) + M.i2b(c, want[1]) and inf[i] == 0, i
    # projective in / projective out: (x : y : 1), identity (0 : 1 : 0)
    zs = [rng.randrange(1, c.p) for _ in range(n)]
    pp = [M.IDENTITY if p is None else (p[0] * z % c.p, p[1] * z % c.p, z) for p, z in zip(ps, zs)]
    outp, _ = _vb(cid, c, ks, pp, lanes, out_fmt=1, proj_in=True)
    for i in range(n):
        want = None if ps[i] is None else M.affine_mul(c, ks[i] % c.n, ps[i])
        got = outp[3 * nb * i:3 * nb * (i + 1)]
        if want is None:
            assert got == M.proj_bytes(c, M.IDENTITY), i
        else:
            assert got == M.i2b(c, want[0]) + M.i2b(c, want[1]) + M.i2b(c, 1), i


def _trace(fn):
    L = lib()
    L.ht_trace_stop.restype = ctypes.c_size_t
    L.ht_trace_stop.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.ht_trace_start()
    fn()
    cap = 1 << 16
    arr = (ctypes.c_int * cap)()
    cnt = L.ht_trace_stop(arr, cap)
    assert cnt <= cap
    return list(arr[:cnt])


def test_k256_reference_schedule_scans_every_table_entry():
    c = M.K256
    L = lib()
    G = M.proj_bytes(c, (c.G[0], c.G[1], 1))
    scalars = [0, 1, c.n - 1, int("f" * 64, 16) % c.n, int("8" * 64, 16) % c.n, synth.scalar(c, 1), synth.scalar(c, 2)]
    traces = []
    for k in scalars:
        out = outbuf(96)
        traces.append(_trace(lambda: L.ht_k256_mul_ref(buf(G), buf(M.i2b(c, k)), out, 1)))
        assert bytes(out) == M.proj_bytes(c, M.k256_mul_ref((c.G[0], c.G[1], 1), k))      # values unchanged by the scan
    # 66 selects (2 x 33 digits) of 8 entries each, in the same order whatever the scalar
    assert all(t == traces[0] for t in traces) and len(traces[0]) == 66 * 8
    assert traces[0][:8] == list(range(8))
    gen = []
    for k in scalars:
        out = outbuf(96)
        gen.append(_trace(lambda: L.ht_k256_mul_gen_ref(buf(M.i2b(c, k)), out, 1)))
        assert bytes(out) == M.proj_bytes(c, M.k256_mul_by_generator_ref(k))
    assert all(t == gen[0] for t in gen) and len(gen[0]) == 65 * 8
    # negative control: the throughput schedule indexes its table by the digits
    fast = []
    for k in scalars[3:]:
        out = outbuf(65)
        fast.append(_trace(lambda: L.ht_k256_mul_fast(buf(M.i2b(c, c.G[0]) + M.i2b(c, c.G[1])), 0, buf(M.i2b(c, k)), out, 1, 1)))
    assert len({tuple(t) for t in fast}) == len(fast)


@pytest.mark.parametrize("cn,cid", CURVES)
def test_primeorder_reference_schedule_scans_every_table_entry(cn, cid):
    c = M.CURVES[cn]
    L = lib()
    nb = c.nbytes
    G = M.proj_bytes(c, (c.G[0], c.G[1], 1))
    traces = []
    for k in [0, 1, c.n - 1, int("f" * (2 * nb), 16) % c.n, synth.scalar(c, 3)]:
        out = outbuf(3 * nb)
        traces.append(_trace(lambda: L.ht_nist_mul_ref(cid, buf(G), buf(M.i2b(c, k)), out, 1, 0)))
        assert bytes(out) == M.proj_bytes(c, M.primeorder_mul_ref(c, (c.G[0], c.G[1], 1), k))
    assert all(t == traces[0] for t in traces) and len(traces[0]) == 2 * nb * 15      # one 15-way scan per 4-bit window
    assert traces[0][:15] == list(range(1, 16))


@pytest.mark.parametrize("cn,cid", CURVES)
@pytest.mark.parametrize("lanes,n", [(3, 29), (1, 5)])
def test_vb_two_term_lincomb_walk(cn, cid, lanes, n):
    """LinearCombination::lincomb (k P + l Q) on the throughput schedule: the two terms of a unit share the doublings;
    4 units per lane and pass.  Edge cases: a zero scalar, an identity point in either slot, P = Q, P = -Q (the sum of
    the two products passes through the doubling / infinity branches of the Jacobian addition), scalars >= n."""
    c = M.CURVES[cn]
    nb = c.nbytes
    ks = [synth.scalar(c, 900 + i) for i in range(2 * n)]
    ps = [synth.point(c, 900 + i) for i in range(2 * n)]
    ks[2] = 0
    ps[5] = None
    if n > 8:
        ps[8] = None
        ps[13] = ps[12]
        ks[13] = ks[12]                       # k P + k P
        ps[15] = (ps[14][0], (-ps[14][1]) % c.p)
        ks[15] = ks[14]                       # k P + k (-P) = identity
        ks[16] = c.n + 7
        ks[19] = c.n - 1
        ks[18] = 1
        ps[19] = ps[18]                       # P + (n - 1) P = identity
        ps[20] = None
        ps[21] = None                         # both terms identity
    sb = b"".join(int(k).to_bytes(nb, "big") for k in ks)
    pb = b"".join(M.i2b(c, p[0]) + M.i2b(c, p[1]) if p is not None else bytes(2 * nb) for p in ps)
    out, inf = outbuf(2 * nb * n), outbuf(n)
    L = lib()
    L.ht_vb_lincomb.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                ctypes.c_size_t, ctypes.c_size_t]
    assert L.ht_vb_lincomb(cid, buf(sb), buf(pb), 0, 2, out, 0, inf, n, lanes) == 0
    o = bytes(out)
    for i in range(n):
        a = None if ps[2 * i] is None else M.affine_mul(c, ks[2 * i] % c.n, ps[2 * i])
        b = None if ps[2 * i + 1] is None else M.affine_mul(c, ks[2 * i + 1] % c.n, ps[2 * i + 1])
        want = M.affine_add(c, a, b)
        got = o[2 * nb * i:2 * nb * (i + 1)]
        if want is None:
            assert got == bytes(2 * nb) and bytes(inf)[i] == 1, i
        else:
            assert got == M.i2b(c, want[0]) + M.i2b(c, want[1]) and bytes(inf)[i] == 0, i


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_constant_time_fixed_base_reads_the_whole_table(cn, cid):
    """fb::mul_ct_one (the body of the signing kernel fb::mul_ct_kernel): k G for edge and random scalars against the
    big-integer model, and the sequence of table entries read - every entry of every window, in order - is the same for
    every scalar."""
    c = M.CURVES[cn]
    nb = c.nbytes
    nwin = (8 * nb + 4) // 5
    G = (c.G[0], c.G[1])
    rows = []
    for j in range(nwin):
        base = M.affine_mul(c, pow(2, 5 * j, c.n), G)
        acc = None
        for d in range(1, 17):
            acc = M.affine_add(c, acc, base)
            rows.append(M.i2b(c, acc[0]) + M.i2b(c, acc[1]))
    table = b"".join(rows)
    scalars = [0, 1, 2, 15, 16, 17, 31, 32, 33, c.n - 1, c.n - 2, c.n - 16, c.n - 17, c.n // 2, c.n // 2 + 1, int("f" * (2 * nb - 1), 16),
               int("8" * (2 * nb), 16) % c.n, (1 << (8 * nb - 1)), c.n, c.n + 5] + [synth.scalar(c, 900 + i) for i in range(6)]
    # the operands closest to an exceptional case of the Jacobian addition (csrc/fixedbase_ct.hpp): the largest top digits with the most
    # negative lower part, every digit at +-16, the scalar n - 2 (n mod 32^(nwin-1)) that would make the last addition a doubling if the
    # bound on the lower digits did not exclude it, single digits d 32^j and d 32^j +- 1
    top = 32 ** (nwin - 1)
    r = c.n % top
    scalars += [(c.n - 2 * r) % c.n, (c.n - 2 * r + 1) % c.n, (c.n // top) * top, (c.n // top) * top - 1, top - 1, top, top + 1, 2 * top - 1,
                sum(16 * 32 ** i for i in range(nwin - 1)) % c.n, (top - sum(16 * 32 ** i for i in range(nwin - 1))) % c.n,
                sum(15 * 32 ** i for i in range(0, nwin - 1, 2)) % c.n]
    scalars += [d * 32 ** j + e for j in (1, 7, nwin - 2) for d in (1, 16, 17, 31) for e in (-1, 0, 1)]
    L = lib()
    L.ht_mul_ct.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    traces = []
    for k in scalars:
        out = outbuf(3 * nb)
        traces.append(_trace(lambda: L.ht_mul_ct(cid, buf(table), buf(int(k).to_bytes(nb, "big")), out, 1)))
        X, Y, Z = (int.from_bytes(bytes(out)[nb * t:nb * (t + 1)], "big") for t in range(3))
        want = M.affine_mul(c, k % c.n, G) if k % c.n else None
        got = None if Z == 0 else (X * pow(Z, -1, c.p) % c.p, Y * pow(Z, -1, c.p) % c.p)
        assert got == want, hex(k)
    assert all(t == traces[0] for t in traces) and traces[0] == list(range(nwin * 16))
