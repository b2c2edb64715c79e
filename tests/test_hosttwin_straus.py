"""Host build of the many-term linear combination (csrc/straus.hpp: groups of up to 16 terms share the doublings of one window loop
over per-term affine tables; a second stage adds the groups' partial sums) and of the exact-(X, Y, Z) run-time-N schedule for
secp256k1 (csrc/mul_k256.hpp: lincomb_ref_term / lincomb_ref_run, k256/src/arithmetic/mul.rs:342-393), walked with a few lanes:
the device arithmetic and schedule checked on the CPU against the big-integer model."""
import ctypes
import random

import pytest

from oracle import ecmodel as M
from hosttwin_util import lib, buf, outbuf

CURVES = [("k256", 0), ("p256", 1), ("p384", 2)]


def _straus(cid, c, ks, ps, terms, lanes, plan_lanes, g_force=0, proj_in=False, out_fmt=0):
    nb = c.nbytes
    n = len(ks) // terms
    sb = b"".join(int(k).to_bytes(nb, "big") for k in ks)
    if proj_in:
        pb = b"".join(M.proj_bytes(c, p) for p in ps)
    else:
        pb = b"".join(M.i2b(c, p[0]) + M.i2b(c, p[1]) if p is not None else bytes(2 * nb) for p in ps)
    out = outbuf((3 if out_fmt else 2) * nb * n)
    inf = outbuf(n)
    used = (ctypes.c_int * 2)()
    f = lib().ht_straus
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                  ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    assert f(cid, buf(sb), buf(pb), 1 if proj_in else 0, terms, out, out_fmt, inf, n, lanes, plan_lanes, g_force, used) == 0
    return bytes(out), bytes(inf), (used[0], used[1])


def _want(c, ks, ps, terms, i):
    acc = None
    for k, P in zip(ks[i * terms:(i + 1) * terms], ps[i * terms:(i + 1) * terms]):
        acc = M.affine_add(c, acc, M.affine_mul(c, k % c.n, P) if P is not None else None)
    return acc


def _inputs(c, n, terms, seed):
    rng = random.Random(seed)
    base = [M.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy)) for _ in range(12)]
    ks = [rng.randrange(c.n) for _ in range(n * terms)]
    ps = [base[rng.randrange(len(base))] for _ in range(n * terms)]
    return rng, ks, ps


@pytest.mark.parametrize("cn,cid", CURVES)
@pytest.mark.parametrize("terms,n,lanes,plan_lanes,g_force", [
    (3, 13, 2, 1, 0),        # g = 3: five combinations per pass share one table inversion; ragged last pass
    (5, 7, 3, 1, 0),         # g = 5: three per pass
    (16, 3, 1, 1, 0),        # one full group per combination
    (17, 3, 2, 1, 0),        # two balanced groups (9 + 8) per combination: the fold adds them
    (40, 2, 3, 1, 0),        # three groups (14 + 14 + 12)
    (6, 5, 2, 1 << 20, 0),   # few combinations on a large machine: the plan falls back to one term per work item (g = 1)
    (7, 4, 2, 1, 2),         # odd group size forced: groups 2 + 2 + 2 + 1
])
def test_straus_against_the_model(cn, cid, terms, n, lanes, plan_lanes, g_force):
    c = M.CURVES[cn]
    nb = c.nbytes
    rng, ks, ps = _inputs(c, n, terms, 1000 * terms + n + cid)
    ks[1] = 0                                              # a zero scalar
    ps[2] = None                                           # an identity point
    ps[terms + 1], ks[terms + 1] = ps[terms], (c.n - ks[terms]) % c.n          # two terms of combination 1 cancel
    if n > 2:                                              # combination 2: every term cancels pairwise / is zero -> the identity
        for t in range(0, terms - 1, 2):
            ps[2 * terms + t + 1], ks[2 * terms + t + 1] = ps[2 * terms + t], (c.n - ks[2 * terms + t]) % c.n
        if terms % 2:
            ks[3 * terms - 1] = 0
    ks[-1] = c.n - 1
    ps[-2], ks[-2] = ps[-1], ks[-1]                        # the same term twice: the accumulator meets P + P inside a window
    out, inf, (g, gpc) = _straus(cid, c, ks, ps, terms, lanes, plan_lanes, g_force)
    assert 1 <= g <= 16 and gpc == -(-terms // g)
    if plan_lanes == 1 and not g_force:
        assert gpc == -(-terms // 16)
    for i in range(n):
        w = _want(c, ks, ps, terms, i)
        got = out[2 * nb * i:2 * nb * (i + 1)]
        if w is None:
            assert got == bytes(2 * nb) and inf[i] == 1, i
        else:
            assert got == M.i2b(c, w[0]) + M.i2b(c, w[1]) and inf[i] == 0, (i, g, gpc)


@pytest.mark.parametrize("cn,cid", CURVES)
def test_straus_projective_in_and_out(cn, cid):
    c = M.CURVES[cn]
    nb = c.nbytes
    terms, n = 4, 6
    rng, ks, ps = _inputs(c, n, terms, 77 + cid)
    ps[5] = None
    pp = [M.IDENTITY if p is None else (lambda z: (p[0] * z % c.p, p[1] * z % c.p, z))(rng.randrange(1, c.p)) for p in ps]
    out, _, _ = _straus(cid, c, ks, pp, terms, 2, 1, proj_in=True, out_fmt=1)
    for i in range(n):
        w = _want(c, ks, ps, terms, i)
        assert out[3 * nb * i:3 * nb * (i + 1)] == M.i2b(c, w[0]) + M.i2b(c, w[1]) + M.i2b(c, 1), i


def test_plan_fills_the_machine_first():
    """straus::plan: the largest group size that still gives every resident lane a work item, balanced over a combination's groups;
    read back from the library through a walk over zero lanes (nothing is computed)."""
    c = M.K256

    def lib_plan(n, terms, lanes):
        nb = c.nbytes
        used = (ctypes.c_int * 2)()
        f = lib().ht_straus
        f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                      ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        assert f(0, None, None, 0, terms, None, 0, None, n, 0, lanes, 0, used) == 0
        return used[0], used[1]

    for n, terms, lanes, want in [(1 << 20, 16, 1 << 18, (16, 1)), (1 << 20, 3, 1 << 18, (3, 1)), (1 << 18, 64, 1 << 18, (16, 4)), (1 << 12, 16, 1 << 18, (1, 16)),
                                  (1 << 16, 16, 1 << 18, (4, 4)), (1 << 20, 17, 1 << 18, (9, 2)), (1 << 10, 1024, 1 << 18, (4, 256))]:
        assert lib_plan(n, terms, lanes) == want, (n, terms, lanes)


@pytest.mark.parametrize("terms", [1, 2, 3, 5, 16])
def test_k256_lincomb_ext_exact_xyz_for_any_length(terms):
    """lincomb_ext over a slice (mul.rs:325-340): the run-time-N schedule returns the very (X, Y, Z) of the model's restatement of
    mul.rs:342-393, projective inputs with Z != 1 and identities included."""
    c = M.K256
    rng = random.Random(500 + terms)
    n = 4
    ks, pts = [], []
    for i in range(n * terms):
        ks.append(rng.randrange(c.n))
        P = M.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy))
        z = rng.randrange(1, c.p)
        pts.append((P[0] * z % c.p, P[1] * z % c.p, z))
    ks[0] = 0
    if terms > 1:
        pts[1] = M.IDENTITY
    sb = b"".join(k.to_bytes(32, "big") for k in ks)
    pb = b"".join(M.proj_bytes(c, P) for P in pts)
    out = outbuf(96 * n)
    f = lib().ht_k256_lincomb_ref_n
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    assert f(buf(sb), buf(pb), terms, out, n) == 0
    for i in range(n):
        want = M.k256_lincomb_ref(list(zip(pts[i * terms:(i + 1) * terms], ks[i * terms:(i + 1) * terms])))
        assert bytes(out)[96 * i:96 * (i + 1)] == M.proj_bytes(c, want), (terms, i)
