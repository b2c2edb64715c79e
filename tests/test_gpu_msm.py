"""k256 multi-scalar multiplication (Pippenger) through the C ABI against the oracle."""
import random

import numpy as np
import pytest

from oracle import coracle as CO
from oracle import ecmodel as M
from oracle import synth

pytestmark = pytest.mark.gpu
C = M.K256
N = C.n


def _set_path(ctx, path, slab=0):
    """"auto": the library's own choice (below 5 * 2^14 terms: n scalar multiplications and a tree sum; 16-bit windows
    below 2^21 terms, 19-bit windows from there on); "buckets16" / "buckets19": the bucket method with that window width
    forced for all sizes (per-context options ECGPU_OPT_MSM_SMALL_PATH = 0 and ECGPU_OPT_MSM_WINDOW_BITS); slab: terms per
    slab of the bucket method (0: the window's maximum)."""
    import ecgpu
    ctx.set_option(ecgpu.OPT_MSM_SMALL_PATH, 1 if path == "auto" else 0)
    ctx.set_option(ecgpu.OPT_MSM_WINDOW_BITS, 0 if path == "auto" else int(path[-2:]))
    ctx.set_option(ecgpu.OPT_MSM_SLAB_TERMS, slab)


@pytest.fixture(scope="module", params=["auto", "buckets16", "buckets19"])
def curve(request):
    """Every test runs three times: see _set_path."""
    import ecgpu
    ctx = ecgpu.Context(0)
    _set_path(ctx, request.param)
    yield ctx.curve("k256")
    ctx.close()


def arr(rows, w):
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(-1, w).copy()


def oracle_msm(ks, pts):
    tot = None
    for k, p in zip(ks, pts):
        tot = M.affine_add(C, tot, M.affine_mul(C, k, p))
    return tot


def test_small_and_edge_cases(curve):
    rng = random.Random(61)
    G = (C.gx, C.gy)
    cases = []
    cases.append(([5], [G]))
    cases.append(([0], [G]))
    cases.append(([N - 1, 1], [G, G]))                          # sums to the identity
    cases.append(([3, 3, 3], [G, G, G]))                        # duplicates: the doubling branch of the bucket sum
    cases.append(([7, 9], [G, None]))                           # identity point among the inputs
    cases.append(([2**16, 2**15, 2**15 - 1, 2**32 - 1, 2**255 % N], [synth.point(C, i, seed=61) for i in range(5)]))
    edge = [(N - 1) // 2, (N + 1) // 2, N - 1, N - 2, (N - 2**255) % N, 2**255 - 1, 2**255 + 1, N - 2**16, N - 2**15, N - 32767,
            0x7FFF8000 << 224, (0x7FFF << 240) | ((1 << 240) - 1),
            # digit boundaries of the 18- / 19-bit windows: runs of ones that carry through every window, single bits at window edges
            2**17, 2**18 - 1, 2**18, 2**18 + 1, 2**36 - 1, 2**72 - 1, 2**72, 2**110, 2**128 - 1, 2**128, 2**128 + 2**17, (1 << 200) - (1 << 17)]
    cases.append((edge, [synth.point(C, 50 + i, seed=61) for i in range(len(edge))]))
    ks = [rng.randrange(N) for _ in range(300)]
    cases.append((ks, [synth.point(C, i, seed=62) for i in range(300)]))
    for ks, pts in cases:
        s = arr([M.i2b(C, k) for k in ks], 32)
        p = arr([bytes(64) if q is None else M.i2b(C, q[0]) + M.i2b(C, q[1]) for q in pts], 64)
        got = curve.msm(s, p)
        want = oracle_msm(ks, pts)
        assert bytes(got) == (bytes(64) if want is None else M.i2b(C, want[0]) + M.i2b(C, want[1])), ks[:3]


def test_projective_io(curve):
    import ecgpu
    rng = random.Random(63)
    n = 50
    ks = [rng.randrange(N) for _ in range(n)]
    pts = [synth.point(C, i, seed=63) for i in range(n)]
    proj = []
    for x, y in pts:
        z = rng.randrange(1, C.p)
        proj.append((x * z % C.p, y * z % C.p, z))
    got = curve.msm(arr([M.i2b(C, k) for k in ks], 32), arr([M.proj_bytes(C, q) for q in proj], 96),
                    point_format=ecgpu.PROJECTIVE, out_format=ecgpu.PROJECTIVE)
    want = oracle_msm(ks, pts)
    assert bytes(got) == M.i2b(C, want[0]) + M.i2b(C, want[1]) + M.i2b(C, 1)


@pytest.mark.parametrize("lg", [10, 16])
def test_unstructured_against_c_oracle(curve, lg):
    """exact comparison against the CPU MSM at 2^10 and 2^16 unstructured terms."""
    n = 1 << lg
    s = CO.synth_scalars(0, n, synth.SEED, 5)
    p = CO.synth_points(0, n, synth.SEED, 5)
    want = CO.msm_naive(0, s, p)
    got = curve.msm(s, p)
    assert bytes(got) == bytes(want[:64]) and want[64] == 0


def test_structured_large(curve):
    """2^20 terms with P_i = (a0 + i d) G, so the answer is (sum k_i (a0 + i d) mod n) G."""
    import torch
    n = 1 << 20
    a0, d = 0x1234567890ABCDEF1234567890ABCDEF % N, 0xFEDCBA0987654321 % N
    ctx = curve.ctx
    ks = CO.synth_scalars(0, n, synth.SEED, 0)
    # point scalars a0 + i*d as big-endian bytes
    vals = [(a0 + i * d) % N for i in range(n)]
    ps = np.frombuffer(b"".join(v.to_bytes(32, "big") for v in vals), dtype=np.uint8).reshape(n, 32).copy()
    d_ps = torch.from_numpy(ps).cuda()
    d_pts = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    curve.mul_device(d_ps, None, d_pts, n)                       # P_i = (a0 + i d) G on the device
    d_ks = torch.from_numpy(ks).cuda()
    d_out = torch.empty((64,), dtype=torch.uint8, device="cuda")
    curve.msm_device(d_ks, d_pts, n, d_out)
    ctx.synchronize()
    tot = 0
    kb = ks.tobytes()
    for i in range(n):
        tot = (tot + int.from_bytes(kb[32 * i:32 * i + 32], "big") * vals[i]) % N
    want = M.affine_mul(C, tot, (C.gx, C.gy))
    assert bytes(d_out.cpu().numpy()) == M.i2b(C, want[0]) + M.i2b(C, want[1])


def test_heavy_buckets_equal_scalars(curve):
    """Equal scalars put every term of a window into one bucket: such a bucket is left in pieces by more than SPAN_MAX
    runs of the bucket sums and folded by whole workgroups (msm_kernels.hpp step 3).  n = 20000, against the oracle's
    term-by-term sum."""
    cv = curve
    n = 20000
    pts = CO.synth_points(0, n, synth.SEED, 4242)
    ones = np.zeros((n, 32), dtype=np.uint8)
    ones[:, 31] = 1
    want = CO.msm_naive(0, ones, pts)
    got = cv.msm(ones, pts)
    assert bytes(got) == bytes(want[:64])
    # one repeated 256-bit scalar (every window heavy) mixed with random ones, and some identity points
    sc = CO.synth_scalars(0, n, synth.SEED, 4242)
    sc[: n // 2] = sc[0]
    pts[7] = 0
    pts[n // 2 + 3] = 0
    want = CO.msm_naive(0, sc, pts)
    got = cv.msm(sc, pts)
    assert bytes(got) == bytes(want[:64])


def test_oversized_sort_bins(curve):
    """2^19 terms whose scalars take one, or three, values: every window's entries fall into one (three) buckets, so the
    sort's bins exceed what one workgroup takes and the many-workgroup path (big_count / big_scan / big_place) runs; the
    buckets are then folded by the workgroup path of the bucket sums.  Structured points P_i = (a0 + i d) G give the
    closed form (sum k_i (a0 + i d) mod n) G."""
    import torch
    n = 1 << 19
    a0, d = 0x1234567890ABCDEF1234567890ABCDEF % N, 0xFEDCBA0987654321 % N
    ctx = curve.ctx
    vals = [(a0 + i * d) % N for i in range(n)]
    ps = np.frombuffer(b"".join(v.to_bytes(32, "big") for v in vals), dtype=np.uint8).reshape(n, 32).copy()
    d_pts = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    curve.mul_device(torch.from_numpy(ps).cuda(), None, d_pts, n)
    ctx.synchronize()
    for kset in ([synth.scalar(C, 11)], [1], [synth.scalar(C, 12), N - 5, 2**200 + 12345]):
        ks = [kset[i % len(kset)] for i in range(n)]
        kb = np.frombuffer(b"".join(M.i2b(C, k) for k in kset), dtype=np.uint8).reshape(len(kset), 32)
        sc = np.tile(kb, (n // len(kset) + 1, 1))[:n].copy()
        d_out = torch.empty((64,), dtype=torch.uint8, device="cuda")
        curve.msm_device(torch.from_numpy(sc).cuda(), d_pts, n, d_out)
        ctx.synchronize()
        tot = sum(k * v for k, v in zip(ks, vals)) % N
        want = M.affine_mul(C, tot, (C.gx, C.gy))
        assert bytes(d_out.cpu().numpy()) == M.i2b(C, want[0]) + M.i2b(C, want[1]), kset[:1]


def test_slabs_of_a_large_sum():
    """Sums above 2^24 terms run in slabs whose window sums are added (a sorted entry keeps the term index in 24 bits).
    ECGPU_OPT_MSM_SLAB_TERMS shrinks the slab so that the loop - three slabs, the last one ragged - runs on 2^16 + 777 terms."""
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve("k256")
    n = (1 << 16) + 777
    s = CO.synth_scalars(0, n, synth.SEED, 123)
    p = CO.synth_points(0, n, synth.SEED, 123)
    p[40000] = 0
    s[50000] = 0
    want = CO.msm_naive(0, s, p)
    for path in ("buckets16", "buckets19"):
        _set_path(ctx, path, slab=30000)
        assert bytes(cv.msm(s, p)) == bytes(want[:64]) and want[64] == 0, path
    ctx.close()


def _oracle_sum(cid, c, s, p, threads=16):
    """sum_i s_i P_i with the C oracle: one reference multiplication per term (threaded), complete additions in a tree."""
    nb = c.nbytes
    parts = CO.lincomb_batch(cid, s, p, out_proj=True, threads=threads)
    ident = np.frombuffer(M.proj_bytes(c, M.IDENTITY), dtype=np.uint8)[None, :]
    while parts.shape[0] > 1:
        if parts.shape[0] % 2:
            parts = np.concatenate([parts, ident])
        parts = CO.point_op(cid, 0, parts[0::2].copy(), parts[1::2].copy())
    X, Y, Z = (int.from_bytes(bytes(parts[0][nb * t:nb * (t + 1)]), "big") for t in range(3))
    return M.to_affine(c, (X, Y, Z))


@pytest.mark.parametrize("cn,cid", [("p256", 1), ("p384", 2)])
@pytest.mark.parametrize("path", ["auto", "buckets16", "buckets19"])
def test_nist_msm(cn, cid, path):
    """The bucket method on the curves without an endomorphism (one half-term per term, 16 / 24 windows of 16 bits and a
    carry window, or 14 / 21 windows of 19 bits; sign fold k > n/2 -> n - k): edge cases against the big-integer model, 2^13 unstructured terms against the C oracle, and
    2^18 structured terms P_i = (a0 + i d) G against the closed form (sum k_i (a0 + i d) mod n) G."""
    import torch
    import ecgpu
    c = M.CURVES[cn]
    nb, n_ord = c.nbytes, c.n
    ctx = ecgpu.Context(0)
    try:
        _set_path(ctx, path)
        cv = ctx.curve(cn)
        G = (c.gx, c.gy)
        rng = random.Random(64 + cid)
        cases = [([5], [G]), ([0], [G]), ([n_ord - 1, 1], [G, G]), ([3, 3, 3], [G, G, G]), ([7, 9], [G, None]),
                 ([(n_ord - 1) // 2, (n_ord + 1) // 2, n_ord - 2, 2**15, 2**16 - 1, 2**(8 * nb - 1) % n_ord, n_ord - 32768,
                   2**17, 2**18 - 1, 2**18, 2**36 - 1, 2**180 - 1, 2**181, (n_ord - 1) // 2 - 2**17, n_ord - 2**18, 2**(8 * nb - 2) - 1, 2**(8 * nb - 2)],
                  [synth.point(c, i, seed=64) for i in range(17)]),
                 ([rng.randrange(n_ord) for _ in range(200)], [synth.point(c, i, seed=65) for i in range(200)])]
        for ks, pts in cases:
            s = arr([M.i2b(c, k) for k in ks], nb)
            p = arr([bytes(2 * nb) if q is None else M.i2b(c, q[0]) + M.i2b(c, q[1]) for q in pts], 2 * nb)
            tot = None
            for k, q in zip(ks, pts):
                tot = M.affine_add(c, tot, None if q is None else M.affine_mul(c, k, q))
            assert bytes(cv.msm(s, p)) == (bytes(2 * nb) if tot is None else M.i2b(c, tot[0]) + M.i2b(c, tot[1])), ks[:3]
        n = 1 << 13
        s = CO.synth_scalars(cid, n, synth.SEED, 77)
        p = CO.synth_points(cid, n, synth.SEED, 77)
        p[100] = 0
        s[200] = 0
        want = _oracle_sum(cid, c, s, p)
        assert bytes(cv.msm(s, p)) == M.i2b(c, want[0]) + M.i2b(c, want[1])
        got = cv.msm(s, p, out_format=ecgpu.PROJECTIVE)
        assert bytes(got) == M.i2b(c, want[0]) + M.i2b(c, want[1]) + M.i2b(c, 1)
        if path != "auto":
            # structured: 2^18 terms through the full-size pipeline
            n = 1 << 18
            a0, d = 0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321
            vals = [a0 + i * d for i in range(n)]
            d_v = torch.from_numpy(np.frombuffer(b"".join(v.to_bytes(nb, "big") for v in vals), dtype=np.uint8).reshape(n, nb).copy()).cuda()
            d_pts = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
            cv.mul_device(d_v, None, d_pts, n)
            ks = CO.synth_scalars(cid, n, synth.SEED, 0)
            d_ks = torch.from_numpy(ks).cuda()
            d_out = torch.empty((2 * nb,), dtype=torch.uint8, device="cuda")
            ctx.synchronize()
            cv.msm_device(d_ks, d_pts, n, d_out)
            ctx.synchronize()
            kb = ks.tobytes()
            tot = sum(int.from_bytes(kb[nb * i:nb * i + nb], "big") * vals[i] for i in range(n)) % n_ord
            w = M.affine_mul(c, tot, G)
            assert bytes(d_out.cpu().numpy()) == M.i2b(c, w[0]) + M.i2b(c, w[1])
    finally:
        ctx.close()


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1)])
def test_bucket_path_agrees_with_term_by_term_path_on_awkward_sizes(cn, cid):
    """Two independent routes to the same sum - the bucket method (digits, two-level sort, bucket parts, reduction tree)
    and n scalar multiplications folded by a tree - on sizes around the wave, workgroup, chunk-alignment (multiples of
    four) and slab boundaries, with duplicate points, identity points, zero scalars and repeated scalars mixed in."""
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cn)
    nmax = 131075
    s_all = CO.synth_scalars(cid, nmax, synth.SEED, 4040)
    p_all = CO.synth_points(cid, nmax, synth.SEED, 4040)
    p_all[3] = 0
    p_all[70] = p_all[69]
    s_all[70] = s_all[69]
    s_all[100] = 0
    s_all[1000:1400] = s_all[999]               # a run of equal scalars: one bucket per window gets 400 extra entries
    p_all[5000] = 0
    try:
        for n in (1, 2, 3, 5, 63, 64, 65, 255, 257, 1023, 1025, 4093, 4095, 4097, 32767, 65537, 131071, 131075):
            s, p = s_all[:n], p_all[:n]
            _set_path(ctx, "auto")
            a = bytes(cv.msm(s, p))
            for path in ("buckets16", "buckets19"):
                _set_path(ctx, path)
                assert bytes(cv.msm(s, p)) == a, (path, n)
                if n > 5000:
                    _set_path(ctx, path, slab=4099)            # ragged slabs (not a multiple of four)
                    assert bytes(cv.msm(s, p)) == a, (path, "slabs", n)
    finally:
        ctx.close()
