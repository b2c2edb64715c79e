"""Behaviour of the C ABI itself (include/ecgpu.h "Ownership / errors / threading" of SURVEY.md section 8b): status
codes, last_error, empty batches, two contexts used from two host threads, ordering on a caller-supplied stream."""
import ctypes
import threading

import numpy as np
import pytest

from oracle import coracle as CO
from oracle import synth

pytestmark = pytest.mark.gpu


def test_status_codes_and_last_error():
    import ecgpu
    ctx = ecgpu.Context(0)
    lib, h = ctx.lib, ctx.handle
    buf = (ctypes.c_uint8 * 256)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.ecgpu_mul_batch(h, 7, p, p, 0, p, 0, None, 1, ecgpu.HOST, 0) == -4         # ECGPU_ERR_UNSUPPORTED: curve
    assert b"curve" in lib.ecgpu_last_error(h)
    assert lib.ecgpu_mul_batch(h, 0, None, p, 0, p, 0, None, 1, ecgpu.HOST, 0) == -1      # ECGPU_ERR_ARG: null scalars
    assert lib.ecgpu_mul_batch(h, 0, p, p, 5, p, 0, None, 1, ecgpu.HOST, 0) == -1         # bad point format
    assert lib.ecgpu_lincomb_batch(h, 0, p, p, 0, 0, p, 0, None, 1, ecgpu.HOST, 0) == -1  # zero terms
    assert lib.ecgpu_msm(h, 9, p, p, 0, 1, p, 0, ecgpu.HOST) == -4                        # unknown curve
    assert lib.ecgpu_schnorr_verify_batch(h, 1, p, p, p, p, 1, ecgpu.HOST) == -4          # BIP340 is k256 only
    assert lib.ecgpu_mul_batch(h, 0, p, p, 0, p, 0, None, 0, ecgpu.HOST, 0) == 0          # empty batch: no-op
    assert lib.ecgpu_field_op_batch(h, 0, 99, p, p, p, 1, ecgpu.HOST) == -1               # unknown field op
    bad = ctypes.c_void_p()
    assert lib.ecgpu_create(ctypes.byref(bad), 99) != 0 and not bad.value                 # no such device
    ctx.close()


def test_two_contexts_from_two_threads():
    """Contexts are independent (own stream, staging buffers, tables): two host threads drive one each and the
    results are those of the oracle (ctypes drops the GIL during the calls, so they really overlap)."""
    import ecgpu
    n = 1 << 16
    jobs = []
    for t, cid in enumerate((0, 1)):
        s = CO.synth_scalars(cid, n, synth.SEED, 1000 * t)
        p = CO.synth_points(cid, n, synth.SEED, 1000 * t)
        jobs.append((cid, s, p))
    results = [None, None]
    errors = []

    def work(t):
        try:
            ctx = ecgpu.Context(0)
            cv = ctx.curve(jobs[t][0])
            for _ in range(3):
                out, inf = cv.mul(jobs[t][1], jobs[t][2])
            g, _ = cv.mul_by_generator(jobs[t][1])
            results[t] = (out, inf, g)
            ctx.close()
        except Exception as e:          # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors
    for t, (cid, s, p) in enumerate(jobs):
        m = 2048
        want = CO.lincomb_batch(cid, s[:m], p[:m], threads=4)
        got = np.concatenate([results[t][0][:m], results[t][1][:m, None]], axis=1)
        assert bytes(got) == bytes(want)
        want_g = CO.lincomb_batch(cid, s[:m], None, threads=4)
        assert bytes(results[t][2][:m]) == bytes(want_g[:, :-1])


def test_caller_stream_ordering():
    """Launches go to the stream set with ecgpu_set_stream: work queued by torch on that stream before and after the
    call is ordered with it, with no synchronisation in between."""
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve("k256")
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    n = 1 << 18
    with torch.cuda.stream(st):
        d_s = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
        d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
        d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
        cv.synth_points_device(d_p, n, synth.SEED, 0)
        d_s[:, 31] = 1                                     # queued on st: every scalar = 1
        cv.mul_device(d_s, d_p, d_o, n)                    # 1 * P = P
        same = (d_o == d_p).all()                          # queued after the kernel on the same stream
    st.synchronize()
    assert bool(same)
    ctx.close()


def test_stream_switch_orders_shared_scratch():
    """ecgpu_set_stream mid-way: host-buffer calls share the context's staging slots, so the second call (new stream)
    must be ordered after the first (old stream); results are those of calls on a single stream."""
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve("k256")
    n = 1 << 17
    s1, p1 = CO.synth_scalars(0, n, synth.SEED, 11), CO.synth_points(0, n, synth.SEED, 11)
    s2, p2 = CO.synth_scalars(0, n, synth.SEED, 999), CO.synth_points(0, n, synth.SEED, 999)
    ref1, ref2 = cv.mul(s1, p1)[0].copy(), cv.mul(s2, p2)[0].copy()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    d_s1, d_p1 = torch.from_numpy(s1).cuda(), torch.from_numpy(p1).cuda()
    d_s2, d_p2 = torch.from_numpy(s2).cuda(), torch.from_numpy(p2).cuda()
    d_o1 = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_o2 = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):                      # device calls share the per-lane table workspace across the switch
        ctx.set_stream(a.cuda_stream)
        cv.mul_device(d_s1, d_p1, d_o1, n)
        ctx.set_stream(b.cuda_stream)
        cv.mul_device(d_s2, d_p2, d_o2, n)
    ctx.synchronize()
    a.synchronize()
    assert bytes(d_o1.cpu().numpy()) == bytes(ref1) and bytes(d_o2.cpu().numpy()) == bytes(ref2)
    ctx.set_stream(0)
    ctx.close()


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_point_eq_and_checked_scalars(cn, cid):
    """ProjectivePoint == (ct_eq, k256 projective.rs:421-446 / primeorder projective.rs:191-198) and the *_checked
    forms that report scalars >= n per element (Scalar::from_repr, k256 scalar.rs:365-368)."""
    import ecgpu
    from oracle import ecmodel as M
    c = M.CURVES[cn]
    nb = c.nbytes
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cn)
    n = 300
    s = CO.synth_scalars(cid, n, synth.SEED, 5)
    p = CO.synth_points(cid, n, synth.SEED, 5)
    xyz = cv.mul(s, p, out_format=ecgpu.PROJECTIVE)                                   # (x : y : 1)
    ref = cv.mul(s, p, out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)      # the reference's own (X : Y : Z)
    assert bytes(xyz) != bytes(ref)
    eq = cv.point_eq(xyz, ref)
    assert eq.all()
    other = np.roll(ref, 1, axis=0)
    assert not cv.point_eq(xyz, other).any()
    ident = np.frombuffer(M.proj_bytes(c, M.IDENTITY), dtype=np.uint8)
    mix = ref.copy()
    mix[0] = ident
    eq = cv.point_eq(np.stack([ident, ident, xyz[2]]), np.stack([ident, ref[1], ident]))
    assert eq.tolist() == [1, 0, 0]
    # scalars >= n: reduced once for the arithmetic, flagged by the checked form
    big = s.copy()
    big[7] = np.frombuffer(M.i2b(c, c.n), dtype=np.uint8)
    big[8] = np.frombuffer((c.n + 12345).to_bytes(nb, "big"), dtype=np.uint8)
    big[9] = 0xFF
    out, inf, ok = cv.lincomb(big, p, checked=True)
    want_ok = np.ones(n, dtype=np.uint8)
    want_ok[7:10] = 0
    assert (ok == want_ok).all() and (ok == cv.validate_scalars(big)).all()
    red = big.copy()
    red[7] = 0
    red[8] = np.frombuffer((12345).to_bytes(nb, "big"), dtype=np.uint8)
    red[9] = np.frombuffer(((1 << (8 * nb)) - 1 - c.n).to_bytes(nb, "big"), dtype=np.uint8)
    o2, i2 = cv.mul(red, p)
    assert bytes(out) == bytes(o2) and bytes(inf) == bytes(i2) and inf[7] == 1
    # two terms per element: one bad scalar spoils its element only
    o3, i3, ok3 = cv.lincomb(big, p, terms=2, checked=True)
    assert ok3.tolist() == [0 if j in (3, 4) else 1 for j in range(n // 2)]
    # argument validation in the binding (a short points array would otherwise be read past its end)
    with pytest.raises(ValueError):
        cv.mul(s, p[:-1])
    with pytest.raises(ValueError):
        cv.lincomb(s[:-1], p[:-1], terms=2)
    with pytest.raises(ValueError):
        cv.msm(s, p[:-3])
    # the empty sum
    assert bytes(cv.msm(s[:0], p[:0])) == bytes(2 * nb)
    assert bytes(cv.msm(s[:0], p[:0], out_format=ecgpu.PROJECTIVE)) == M.proj_bytes(c, M.IDENTITY)
    ctx.close()


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_signing_schedules_agree_and_ecdh(cn, cid):
    """Signing defaults to the constant-time fixed-base kernel for k G (every table entry read, complete additions);
    ECGPU_EXACT_REFERENCE selects the reference's own constant-time schedule, ECGPU_PUBLIC_SCALARS the throughput
    schedule.  Same signatures from all three, edge nonces included (1, 2, n - 1, n - 2, single-window digits 16 and 17,
    runs of ones that carry through every window, and 0 / n, which must come back not ok).  ECDH (a secret-scalar
    multiplication) goes through the reference schedule."""
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cn)
    n = 3000
    d = CO.synth_scalars(cid, n, synth.SEED, 21)
    k = CO.synth_scalars(cid, n, synth.SEED + 1, 21)
    z = CO.synth_scalars(cid, n, synth.SEED + 2, 21)
    c = synth.M.CURVES[cn]
    nb = c.nbytes
    edge = [1, 2, 15, 16, 17, 31, 32, 33, c.n - 1, c.n - 2, c.n - 16, (1 << (8 * nb - 3)) - 1, int("f" * (2 * nb - 2), 16), (1 << 250) + (1 << 5) * 16,
            c.n // 2, c.n // 2 + 1, 0, c.n]
    for i, v in enumerate(edge):
        k[1000 + i] = np.frombuffer(int(v).to_bytes(nb, "big"), dtype=np.uint8)
    bad = [1000 + len(edge) - 2, 1000 + len(edge) - 1]               # k = 0 and k = n
    fl = cv.default_ecdsa_flags()
    sig_ct, rec_ct, ok_ct = cv.ecdsa_sign(d, k, z, flags=fl)
    sig_pub, rec_pub, ok_pub = cv.ecdsa_sign(d, k, z, flags=fl | ecgpu.PUBLIC_SCALARS)
    sig_ref, rec_ref, ok_ref = cv.ecdsa_sign(d, k, z, flags=fl | ecgpu.EXACT_REFERENCE)
    assert not ok_ct[bad].any() and ok_ct.sum() == n - len(bad)
    assert bytes(sig_ct) == bytes(sig_pub) and bytes(rec_ct) == bytes(rec_pub) and bytes(ok_ct) == bytes(ok_pub)
    assert bytes(sig_ct) == bytes(sig_ref) and bytes(rec_ct) == bytes(rec_ref) and bytes(ok_ct) == bytes(ok_ref)
    s2, r2, _ = CO.ecdsa_sign_batch(cid, d[:200], k[:200], z[:200], low_s=(cid == 0))
    assert bytes(sig_ct[:200]) == bytes(s2) and bytes(rec_ct[:200]) == bytes(r2)
    lo, hi = 1000, 1000 + len(edge) - 2              # the valid edge nonces against the oracle
    s3, r3, _ = CO.ecdsa_sign_batch(cid, d[lo:hi], k[lo:hi], z[lo:hi], low_s=(cid == 0))
    assert bytes(sig_ct[lo:hi]) == bytes(s3) and bytes(rec_ct[lo:hi]) == bytes(r3)
    k[bad] = k[0]                                    # ECDH below wants valid scalars
    q, _ = cv.mul_by_generator(d)
    shared = cv.diffie_hellman(k, q)                 # k (dG)
    q2, _ = cv.mul_by_generator(k)
    shared2 = cv.diffie_hellman(d, q2)               # d (kG)
    assert bytes(shared) == bytes(shared2)
    want = CO.lincomb_batch(cid, k[:100], q[:100], threads=4)
    assert bytes(shared[:100]) == bytes(np.ascontiguousarray(want[:, :cv.nb]))
    ctx.close()


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_secret_scalars_flag(cn, cid):
    """ECGPU_SECRET_SCALARS: key generation d G on the constant-time fixed-base kernel, a variable base on the reference
    schedule - the same affine bytes as the throughput schedules, identity results (d = 0, d = n) flagged, projective
    output normalised to Z = 1 (identity (0, 1, 0))."""
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cn)
    c = synth.M.CURVES[cn]
    nb = c.nbytes
    n = 5000
    d = CO.synth_scalars(cid, n, synth.SEED, 91)
    for i, v in enumerate([0, c.n, 1, 2, 16, 17, c.n - 1, c.n + 5, c.n // 2, c.n // 2 + 1, (1 << (8 * nb)) - 1]):
        d[40 + i] = np.frombuffer(int(v).to_bytes(nb, "big"), dtype=np.uint8)
    pub, inf = cv.mul_by_generator(d)
    pub_ct, inf_ct = cv.mul_by_generator(d, flags=ecgpu.SECRET_SCALARS)
    assert bytes(pub_ct) == bytes(pub) and bytes(inf_ct) == bytes(inf) and inf[40] == 1 and inf[41] == 1 and inf.sum() == 2
    proj = cv.mul_by_generator(d, out_format=ecgpu.PROJECTIVE, flags=ecgpu.SECRET_SCALARS)
    proj = proj[0] if isinstance(proj, tuple) else proj
    one = (1).to_bytes(nb, "big")
    assert bytes(proj[7]) == bytes(pub[7]) + one and bytes(proj[40]) == bytes(nb) + one + bytes(nb)
    p = CO.synth_points(cid, 300, synth.SEED, 92)
    a, ai = cv.mul(d[:300], p)
    b, bi = cv.mul(d[:300], p, flags=ecgpu.SECRET_SCALARS)
    assert bytes(a) == bytes(b) and bytes(ai) == bytes(bi)
    ctx.close()


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p384", 2)])
def test_fold_points_device(cn, cid):
    """parallel.fold_points_device: the all-gathered partial sums of a split MSM (one projective point per rank) are summed
    on the device by a tree of complete additions - any count, an identity among them, opposite points."""
    import torch
    import ecgpu
    from ecgpu import parallel
    c = synth.M.CURVES[cn]
    nb = c.nbytes
    ctx = ecgpu.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    cv = ctx.curve(cn)
    rng = np.random.default_rng(5)
    pts = [synth.point(c, 300 + i, seed=44) for i in range(9)]
    pts[3] = None
    pts[6] = synth.M.affine_neg(c, pts[5])
    for count in (1, 2, 3, 5, 8, 9):
        rows = []
        for p in pts[:count]:
            if p is None:
                rows.append(synth.M.proj_bytes(c, synth.M.IDENTITY))
            else:
                z = int(rng.integers(2, 1 << 62))
                rows.append(synth.M.proj_bytes(c, (p[0] * z % c.p, p[1] * z % c.p, z)))
        d_all = torch.from_numpy(np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(count, 3 * nb).copy()).cuda()
        d_scr = torch.empty_like(d_all)
        d_out = torch.empty((2 * nb,), dtype=torch.uint8, device="cuda")
        d_inf = torch.empty((1,), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        parallel.fold_points_device(cv, d_all, count, d_scr, d_out, d_inf)
        ctx.synchronize()
        want = None
        for p in pts[:count]:
            want = synth.M.affine_add(c, want, p)
        got = bytes(d_out.cpu().numpy())
        assert got == (bytes(2 * nb) if want is None else synth.M.i2b(c, want[0]) + synth.M.i2b(c, want[1])), count
        assert int(d_inf.cpu()[0]) == (1 if want is None else 0)
    ctx.close()


def test_plain_c_caller():
    """The boundary from C, without Python in the data path: examples/abi_example.c (generator multiples, complete
    addition, point equality, sign + verify) exits 0."""
    import os
    import subprocess
    from conftest import ROOT
    d = os.path.join(ROOT, "examples")
    subprocess.run(["make", "-s", "-C", d], check=True)
    r = subprocess.run([os.path.join(d, "abi_example")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi example ok" in r.stdout and "2 G x = c6047f9441ed7d6d3045406e95c07cd85c778e4b8cef3ca7abac09b95c709ee5" in r.stdout


def _slow_producer_then(ctx_setup):
    """A torch producer that keeps the legacy default stream busy for milliseconds, a dependent write of the library's input
    on that stream, the library call straight after with NO synchronisation, one small kernel with nothing queued in front."""
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    ctx_setup(ctx)
    cv = ctx.curve("k256")
    n = 1024
    d_a = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    d_b = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    d_o = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    big = torch.empty((1 << 31,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        for i in range(12):
            big.fill_(i + rep)                       # ~12 x 0.4 ms on the default stream
        d_a[:, 31] = big[-1]                         # = 11 + rep once the fills are through; device-side, no host sync
        d_b[:, 31] = 1
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        ctx.check(ctx.lib.ecgpu_field_op_batch(ctx.handle, 0, ecgpu.FE_ADD, p(d_a), p(d_b), p(d_o), n, ecgpu.DEVICE))
        got = d_o[:, 31].clone()                     # consumer on the default stream, again without a sync
        torch.cuda.synchronize()
        assert bool((got == 12 + rep).all()), (rep, got[:4].tolist())
    ctx.close()


def test_default_stream_producer_is_ordered_with_the_contexts_own_stream():
    """The context's own stream is a BLOCKING stream: work the caller queued on the legacy default stream (PyTorch's
    default stream) is ordered before the library's launches and after them, with no call to ecgpu_set_stream at all."""
    _slow_producer_then(lambda ctx: None)


def test_set_stream_null_is_the_legacy_default_stream():
    """ecgpu_set_stream(ctx, NULL) binds the legacy default stream, as every HIP API reads a NULL stream (round 2 read it
    as "the context's own non-blocking stream", which left torch's default stream unordered with the library)."""
    import torch
    _slow_producer_then(lambda ctx: ctx.set_stream(torch.cuda.current_stream().cuda_stream))


def test_own_stream_after_a_destroyed_caller_stream():
    """The previous stream may be gone at the switch: the library falls back to a device synchronisation and installs the
    new stream all the same (a failed event record left the context pinned to the dead stream)."""
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve("p256")
    s = CO.synth_scalars(1, 256, synth.SEED, 5)
    want, _ = cv.mul_by_generator(s)
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    a, _ = cv.mul_by_generator(s)
    ctx.use_own_stream()
    b, _ = cv.mul_by_generator(s)
    ctx.set_stream(0)
    c_, _ = cv.mul_by_generator(s)
    assert bytes(a) == bytes(want) == bytes(b) == bytes(c_)
    ctx.close()


def test_options_and_error_text():
    import ecgpu
    ctx = ecgpu.Context(0)
    lib, h = ctx.lib, ctx.handle
    assert ctx.get_option(ecgpu.OPT_FB_MAX_WINDOW) == 26 and ctx.get_option(ecgpu.OPT_MSM_SMALL_PATH) == 1
    for opt, bad in ((ecgpu.OPT_FB_WINDOW, 12), (ecgpu.OPT_FB_MAX_WINDOW, 0), (ecgpu.OPT_MSM_WINDOW_BITS, 18), (ecgpu.OPT_MSM_SLAB_TERMS, 5),
                     (ecgpu.OPT_MSM_SMALL_PATH, 2), (ecgpu.OPT_MSM_ROUNDS, 65), (ecgpu.OPT_K256_WAVES, 2), (99, 1)):
        assert lib.ecgpu_set_option(h, opt, bad) == -1, (opt, bad)
        assert "ecgpu_set_option" in ctx.last_error()
    ctx.set_option(ecgpu.OPT_FB_WINDOW, 16)
    assert ctx.get_option(ecgpu.OPT_FB_WINDOW) == 16
    cv = ctx.curve("k256")
    s = CO.synth_scalars(0, 512, synth.SEED, 3)
    a, _ = cv.mul_by_generator(s)
    assert ctx.fb_table_bytes("k256")[1] == 16 and ctx.fb_table_bytes("k256")[0] > 30_000_000       # 17 x 2^15 x 64 B + the 8-bit table
    ctx.set_option(ecgpu.OPT_FB_WINDOW, 8)
    b, _ = cv.mul_by_generator(s)
    ctx.set_option(ecgpu.OPT_K256_WAVES, 3)
    p = CO.synth_points(0, 512, synth.SEED, 3)
    c3, _ = cv.mul(s, p)
    ctx.set_option(ecgpu.OPT_K256_WAVES, 4)
    c4, _ = cv.mul(s, p)
    assert bytes(a) == bytes(b) and bytes(c3) == bytes(c4)
    assert ctx.fb_table_bytes("p384") == (0, 0)
    ctx.close()


def test_generator_table_falls_back_when_memory_is_short():
    """A batch of 2^23 results wants the 24-bit generator table (5.9 GB).  When that table does not fit the call must step
    down (to the 20-bit table, 436 MB), return correct results, remember the cap, and use the wide table again once the cap
    is lifted (VERDICT r2 weak 2 / ADVICE).  "Does not fit" is produced through ECGPU_OPT_FB_MEMORY_BUDGET, which enters the
    very branch a failed hipMalloc enters; a real allocation failure could not be provoked on this pool - with all but
    3.5 GB of the card held by a ballast tensor hipMalloc still handed out the 5.9 GB (round-3 run, gpurun_out/r3/t2.log)
    - and driving the box further out of memory is not something a test should do."""
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve("p256")
    n = 1 << 23
    d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_o2 = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, 31337)
    torch.cuda.synchronize()
    ctx.set_option(ecgpu.OPT_FB_MEMORY_BUDGET, 1 << 30)
    assert ctx.get_option(ecgpu.OPT_FB_MAX_WINDOW) == 26
    cv.mul_device(d_s, None, d_o, n)
    ctx.synchronize()
    cap = ctx.get_option(ecgpu.OPT_FB_MAX_WINDOW)
    nbytes, widest = ctx.fb_table_bytes("p256")
    assert cap == 20 and widest == 20 and 0 < nbytes < (1 << 30), (cap, widest, nbytes)
    idx = np.unique(np.concatenate([np.arange(0, 2048), np.arange(n - 2048, n), np.arange(0, n, n // 4096)]))
    want = CO.lincomb_batch(1, np.ascontiguousarray(d_s.cpu().numpy()[idx]), None, threads=16)
    assert bytes(d_o.cpu().numpy()[idx]) == bytes(want[:, :-1])
    # a second call does not try the wide table again ...
    cv.mul_device(d_s, None, d_o2, n)
    ctx.synchronize()
    assert ctx.fb_table_bytes("p256") == (nbytes, 20) and torch.equal(d_o, d_o2)
    # ... until budget and cap are lifted
    ctx.set_option(ecgpu.OPT_FB_MEMORY_BUDGET, 0)
    ctx.set_option(ecgpu.OPT_FB_MAX_WINDOW, 26)
    d_o2.zero_()
    cv.mul_device(d_s, None, d_o2, n)
    ctx.synchronize()
    assert ctx.fb_table_bytes("p256")[1] == 24 and ctx.fb_table_bytes("p256")[0] > (5 << 30)
    assert torch.equal(d_o, d_o2)
    # a budget below even the 16-bit table: the 8-bit table (270 KB, always allowed) serves a small batch
    ctx2 = ecgpu.Context(0)
    ctx2.set_option(ecgpu.OPT_FB_MEMORY_BUDGET, 1 << 20)
    cv2 = ctx2.curve("p256")
    m = 1 << 18
    d_o3 = torch.empty((m, 64), dtype=torch.uint8, device="cuda")
    cv2.mul_device(d_s[:m], None, d_o3, m)
    ctx2.synchronize()
    assert ctx2.fb_table_bytes("p256")[1] == 8 and torch.equal(d_o3, d_o[:m])
    ctx2.close()
    ctx.close()


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_sec1_uncompressed_and_compressed_records(cn, cid, ref_vectors):
    """ToEncodedPoint / FromEncodedPoint in fixed-width records: the reference's base-point encodings
    (k256/src/arithmetic/affine.rs:374-381, p256/tests/affine.rs, p384/tests/affine.rs), round trips, the identity, and
    the rejections of from_encoded_point (coordinate >= p, off the curve, unknown tag, padding that is not zero)."""
    import ecgpu
    c = synth.M.CURVES[cn]
    nb = c.nbytes
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cn)
    enc = ref_vectors[cn]["encoding"]
    G = synth.M.i2b(c, c.gx) + synth.M.i2b(c, c.gy)
    n = 200
    pts = CO.synth_points(cid, n, synth.SEED, 616)
    pts[0] = np.frombuffer(G, dtype=np.uint8)
    pts[7] = 0                                           # identity
    unc = cv.sec1_encode(pts)
    cmp_ = cv.sec1_encode(pts, compress=True)
    assert bytes(unc[0]).hex() == enc["uncompressed_basepoint"] and bytes(cmp_[0]).hex() == enc["compressed_basepoint"]
    assert unc.shape == (n, 1 + 2 * nb) and not unc[7].any() and not cmp_[7].any()
    assert bytes(cmp_) == bytes(cv.to_bytes(pts))        # GroupEncoding::to_bytes is the compressed record
    for i in (1, 50, n - 1):
        assert unc[i][0] == 4 and bytes(unc[i][1:]) == bytes(pts[i])
    back, ok = cv.sec1_decode(unc)
    assert ok.all() and bytes(back) == bytes(pts)
    back, ok = cv.sec1_decode(cmp_, record_bytes=1 + nb)
    assert ok.all() and bytes(back) == bytes(pts)
    # projective input is normalised on the device
    proj = cv.mul(CO.synth_scalars(cid, n, synth.SEED, 9), pts, out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    aff, inf = cv.batch_normalize(proj)
    assert bytes(cv.sec1_encode(proj, point_format=ecgpu.PROJECTIVE)) == bytes(cv.sec1_encode(aff))
    # compressed and compact forms inside wide records (zero padded)
    wide = np.zeros((n, 1 + 2 * nb), dtype=np.uint8)
    wide[:, :1 + nb] = cmp_
    back, ok = cv.sec1_decode(wide)
    assert ok.all() and bytes(back) == bytes(pts)
    if "uncompact_basepoint" in enc:
        one = np.frombuffer(bytes.fromhex(enc["uncompact_basepoint"]), dtype=np.uint8).reshape(1, -1)
        b1, ok1 = cv.sec1_decode(one)
        b2, ok2 = cv.sec1_decode(np.frombuffer(bytes.fromhex(enc["compact_basepoint"]), dtype=np.uint8).reshape(1, -1), record_bytes=1 + nb)
        assert ok1[0] == 1 and ok2[0] == 1 and bytes(b1) == bytes(b2) == bytes(one[0][1:])
    # rejections
    bad = unc[:8].copy()
    bad[0][-1] ^= 1                                      # y off by one: not on the curve
    bad[1][0] = 6                                        # unknown tag
    bad[2][1:1 + nb] = np.frombuffer(int(c.p).to_bytes(nb, "big"), dtype=np.uint8)          # x = p
    bad[3][1 + nb:] = np.frombuffer(int(c.p + 1).to_bytes(nb, "big"), dtype=np.uint8)      # y >= p
    bad[4][0] = 2                                        # compressed tag with a non-zero tail
    bad[5][0] = 0                                        # identity tag with a non-zero body
    bad[6][0] = 4                                        # fine
    bad[7] = 0                                           # the identity
    out, ok = cv.sec1_decode(bad)
    assert ok.tolist() == [0, 0, 0, 0, 0, 0, 1, 1] and not out[:6].any() and bytes(out[6]) == bytes(pts[6])
    assert ctx.lib.ecgpu_sec1_decode_batch(ctx.handle, cid, ctypes.c_void_p(bad.ctypes.data), 17, ctypes.c_void_p(out.ctypes.data),
                                           ctypes.c_void_p(ok.ctypes.data), 8, ecgpu.HOST) == -1
    ctx.close()


def test_memory_budget_never_refuses_the_mandatory_tables():
    """ECGPU_OPT_FB_MEMORY_BUDGET is documented as a knob "never needed for correct results", yet a budget below the
    5-bit table of the constant-time fixed-base kernel made signing and SECRET_SCALARS key generation fail with an out-of-memory
    error.  The budget is about the optional wide tables: with a budget of one byte signing, key generation and the public-data
    generator multiplication (on the 8-bit base table) all work and agree with an unbudgeted context."""
    import ecgpu
    ctx, ref = ecgpu.Context(0), ecgpu.Context(0)
    ctx.set_option(ecgpu.OPT_FB_MEMORY_BUDGET, 1)
    for cn, cid in (("k256", 0), ("p384", 2)):
        cv, rv = ctx.curve(cn), ref.curve(cn)
        n = 300_000                                   # the size rule would pick the 16-bit table
        d = CO.synth_scalars(cid, n, synth.SEED, 77)
        k = CO.synth_scalars(cid, n, synth.SEED + 1, 77)
        z = CO.synth_scalars(cid, n, synth.SEED + 2, 77)
        sig, rec, ok = cv.ecdsa_sign(d[:5000], k[:5000], z[:5000])
        sig2, rec2, ok2 = rv.ecdsa_sign(d[:5000], k[:5000], z[:5000])
        assert ok.all() and bytes(sig) == bytes(sig2) and bytes(rec) == bytes(rec2)
        keys, _ = cv.mul_by_generator(d[:5000], flags=ecgpu.SECRET_SCALARS)
        pub, _ = cv.mul_by_generator(d)
        pub2, _ = rv.mul_by_generator(d)
        assert bytes(pub) == bytes(pub2) and bytes(keys) == bytes(pub[:5000])
        assert ctx.fb_table_bytes(cn)[1] == 8 and ref.fb_table_bytes(cn)[1] == 16
    ctx.close()
    ref.close()
