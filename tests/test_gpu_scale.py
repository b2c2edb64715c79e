"""Full-size, size-independent properties (k256, 2^22 units): the variable-base kernel, the MSM and the
fixed-base kernel are three independent code paths that must agree, and a 2^20 slice is compared
byte for byte with the C oracle."""
import numpy as np
import pytest

from oracle import coracle as CO
from oracle import ecmodel as M
from oracle import synth

pytestmark = pytest.mark.gpu
C = M.K256
N = C.n


@pytest.fixture(scope="module")
def env():
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    yield ctx, ctx.curve("k256")
    ctx.close()


def test_varbase_outputs_sum_to_the_msm(env):
    """sum_i (k_i P_i) computed as (a) 2^22 independent scalar multiplications folded with an MSM with unit
    scalars and (b) one Pippenger MSM over the same terms: a checksum of checksums across kernels."""
    import torch
    ctx, cv = env
    n = 1 << 22
    d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, 9000000)
    cv.synth_points_device(d_p, n, synth.SEED, 9000000)
    cv.mul_device(d_s, d_p, d_o, n)
    ones = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    ones[:, 31] = 1
    d_a = torch.empty((64,), dtype=torch.uint8, device="cuda")
    d_b = torch.empty((64,), dtype=torch.uint8, device="cuda")
    cv.msm_device(ones, d_o, n, d_a)          # sum of the outputs
    cv.msm_device(d_s, d_p, n, d_b)           # the MSM of the inputs
    ctx.synchronize()
    a, b = bytes(d_a.cpu().numpy()), bytes(d_b.cpu().numpy())
    assert a == b and a != bytes(64)


def test_linearity_and_fixed_base_consistency(env):
    """(k + l) P = kP + lP on 2^20 random inputs; k (mG) = (k m) G ties the variable-base kernel to the
    fixed-base kernel (8- and 16-bit tables) through scalar arithmetic done on the host."""
    import torch
    import ecgpu
    ctx, cv = env
    n = 1 << 20
    ks = CO.synth_scalars(0, n, synth.SEED, 5)
    ms = CO.synth_scalars(0, n, synth.SEED, 7000000)
    kb, mb = ks.tobytes(), ms.tobytes()
    prod = np.frombuffer(b"".join(((int.from_bytes(kb[32 * i:32 * i + 32], "big") * int.from_bytes(mb[32 * i:32 * i + 32], "big")) % N).to_bytes(32, "big")
                                  for i in range(0, n, 16)), dtype=np.uint8).reshape(-1, 32).copy()
    d_m = torch.from_numpy(ms).cuda()
    d_pts = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    cv.mul_device(d_m, None, d_pts, n)                       # m_i G  (wide fixed-base table)
    d_k = torch.from_numpy(ks).cuda()
    d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    cv.mul_device(d_k, d_pts, d_o, n)                        # k_i (m_i G)
    sub = torch.from_numpy(prod).cuda()
    d_chk = torch.empty((sub.shape[0], 64), dtype=torch.uint8, device="cuda")
    cv.mul_device(sub, None, d_chk, sub.shape[0])            # (k_i m_i) G  (8-bit table: 2^16 scalars)
    ctx.synchronize()
    assert bytes(d_o.cpu().numpy()[::16].copy()) == bytes(d_chk.cpu().numpy())


def test_slice_against_c_oracle(env):
    import torch
    ctx, cv = env
    n = 1 << 20
    first = 31337
    d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, first)
    cv.synth_points_device(d_p, n, synth.SEED, first)
    cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i)
    ctx.synchronize()
    m = 1 << 15                                            # ~2 s of single-thread oracle time
    idx = np.arange(0, n, n // m)
    s, p = d_s.cpu().numpy()[idx].copy(), d_p.cpu().numpy()[idx].copy()
    want = CO.lincomb_batch(0, s, p, threads=8)
    got = np.concatenate([d_o.cpu().numpy()[idx], d_i.cpu().numpy()[idx][:, None]], axis=1)
    assert bytes(got) == bytes(want)


def test_host_buffer_pipeline_matches_device_path(env):
    """Host-buffer calls of 2^21 units and more stream through three device slots in chunks that grow from an eighth of a kernel
    pass to a whole pass (upload, kernels and download overlapped, csrc/host_pipe.hpp): results must be those of the one-shot
    device-resident call, including the ragged last chunk.  Here: pageable numpy arrays (the bounce-buffer path), two chunks."""
    import torch
    ctx, cv = env
    n = (1 << 21) + 12345
    d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, 777)
    cv.synth_points_device(d_p, n, synth.SEED, 777)
    d_s[5] = 0                                   # an identity result inside the batch
    cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i)
    ctx.synchronize()
    hs, hp = d_s.cpu().numpy(), d_p.cpu().numpy()
    out, inf = cv.mul(hs, hp)
    assert bytes(out) == bytes(d_o.cpu().numpy()) and bytes(inf) == bytes(d_i.cpu().numpy())
    assert inf[5] == 1 and inf.sum() == 1
    # fixed base and ECDSA through the same pipeline
    out_g, _ = cv.mul_by_generator(hs)
    cv.mul_device(d_s, None, d_o, n, d_out_inf=d_i)
    ctx.synchronize()
    assert bytes(out_g) == bytes(d_o.cpu().numpy())
    hk = CO.synth_scalars(0, n, synth.SEED + 1, 0)
    hz = CO.synth_scalars(0, n, synth.SEED + 2, 0)
    hs[5, 31] = 9                                # secret keys must be non-zero
    sig, rec, ok = cv.ecdsa_sign(hs, hk, hz)
    assert ok.all()
    q, _ = cv.mul_by_generator(hs)
    sig[::4099, 7] ^= 0x10
    v = cv.ecdsa_verify(hz, sig, q)
    want = np.ones(n, dtype=np.uint8)
    want[::4099] = 0
    assert (v == want).all()
    # spot-check the signatures against the C oracle
    idx = np.arange(0, n, 65537)
    s2, r2, k2 = CO.ecdsa_sign_batch(0, hs[idx], hk[idx], hz[idx], low_s=True)
    good = idx % 4099 != 0
    assert bytes(sig[idx][good]) == bytes(s2[good]) and bytes(rec[idx]) == bytes(r2)


@pytest.mark.parametrize("cn,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_ragged_batch_sizes_are_prefix_consistent(cn, cid):
    """Element i of a batch must not depend on the batch size: every kernel walks its batch with grid strides and
    per-lane sub-batches (shared inversions), so sizes around the wave, workgroup and grid boundaries are compared
    with the prefix of one large batch (whose head is checked against the C oracle)."""
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cn)
    nb = cv.nb
    big = 70001
    s = CO.synth_scalars(cid, big, synth.SEED, 31)
    p = CO.synth_points(cid, big, synth.SEED, 31)
    s[3] = 0                                        # identity result
    p[9] = 0                                        # identity input
    out, inf = cv.mul(s, p)
    gen, ginf = cv.mul_by_generator(s)
    want = CO.lincomb_batch(cid, s[:300], p[:300], threads=4)
    assert bytes(np.concatenate([out[:300], inf[:300, None]], axis=1)) == bytes(want)
    z = CO.synth_scalars(cid, big, synth.SEED + 2, 31)
    k = CO.synth_scalars(cid, big, synth.SEED + 1, 31)
    d = s.copy()
    d[3, -1] = 5
    sig, rec, okd = cv.ecdsa_sign(d, k, z)
    keys, _ = cv.mul_by_generator(d)
    sig[::7, 5] ^= 1
    ver = cv.ecdsa_verify(z, sig, keys)
    enc = cv.to_bytes(out)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 1023, 1025, 4095, 4097, 16385, 65537):
        o2, i2 = cv.mul(s[:n], p[:n])
        assert bytes(o2) == bytes(out[:n]) and bytes(i2) == bytes(inf[:n]), n
        g2, gi2 = cv.mul_by_generator(s[:n])
        assert bytes(g2) == bytes(gen[:n]) and bytes(gi2) == bytes(ginf[:n]), n
        assert bytes(cv.ecdsa_verify(z[:n], sig[:n], keys[:n])) == bytes(ver[:n]), n
        s2, r2, k2 = cv.ecdsa_sign(d[:n], k[:n], z[:n])
        s2[::7, 5] ^= 1
        assert bytes(s2) == bytes(sig[:n]) and bytes(r2) == bytes(rec[:n]), n
        assert bytes(cv.to_bytes(out[:n])) == bytes(enc[:n]), n
    assert ver.sum() == big - len(range(0, big, 7))
    ctx.close()


def test_host_buffer_pipeline_slot_reuse_pinned_and_pageable(env):
    """2^22 + 2^20 + 4321 units = four chunks (2^20, 2^21, 2^20 + 4321, 2^20: the fourth reuses the first slot) from page-locked
    buffers (direct DMA) and from pageable ones (bounce pool, helper threads): byte-identical to the device-resident call, with the
    per-element scalar verdicts of the checked entry point riding along as a fifth argument."""
    import torch
    import ecgpu
    ctx, cv = env
    n = (1 << 22) + (1 << 20) + 4321
    assert ecgpu.host_chunk_schedule(n, 1 << 23) == [1 << 20, 1 << 21, (1 << 20) + 4321, 1 << 20]
    d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, 424242)
    cv.synth_points_device(d_p, n, synth.SEED, 424242)
    edges = [0, (1 << 20) - 1, 1 << 20, (3 << 20) - 1, 3 << 20, (4 << 20) + 4320, (4 << 20) + 4321, n - 1]      # first / last unit of every chunk
    for j, i in enumerate(edges):
        if j % 2:
            d_s[i] = 0                              # identity results at chunk boundaries
        else:
            d_s[i] = 255                            # 2^256 - 1 >= n: scalar_ok = 0, the result is that of the reduced scalar
    cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i)
    ctx.synchronize()
    want_o, want_i = d_o.cpu().numpy(), d_i.cpu().numpy()
    hs, hp = d_s.cpu().numpy(), d_p.cpu().numpy()
    out, inf, ok = cv.lincomb(hs, hp, checked=True)                  # pageable in, pageable out
    assert bytes(out) == bytes(want_o) and bytes(inf) == bytes(want_i)
    want_ok = np.ones(n, dtype=np.uint8)
    want_ok[edges[0::2]] = 0
    assert (ok == want_ok).all() and inf.sum() == len(edges) // 2
    ps, pp = ctx.pinned_array((n, 32)), ctx.pinned_array((n, 64))
    po, pi = ctx.pinned_array((n, 64)), ctx.pinned_array((n,))
    ps[:] = hs
    pp[:] = hp
    po[:] = 0xA5
    cv.mul(ps, pp, out=po, out_inf=pi)                                # page-locked in and out
    assert bytes(po) == bytes(want_o) and bytes(pi) == bytes(want_i)
    out2, inf2 = cv.mul(ps, hp)                                       # mixed: page-locked scalars, pageable points and outputs
    assert bytes(out2) == bytes(want_o) and bytes(inf2) == bytes(want_i)


def test_msm_from_host_memory_streams_in_parts(env):
    """ecgpu_msm with host buffers cuts sums of 2^22 terms and more into parts that upload while the previous part is summed
    (ecgpu.hip; round 3 staged the whole input first): 2^23 + 2^22 + 777 terms = three parts, equal to the device-resident sum
    and to the closed form over structured scalars; affine and projective output."""
    import torch
    import ecgpu
    ctx, cv = env
    n = (1 << 23) + (1 << 22) + 777
    d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, 606060)
    cv.synth_points_device(d_p, n, synth.SEED, 606060)
    d_s[12345] = 0
    d_p[(1 << 22) + 5] = 0                       # an identity point in the second part
    d_r = torch.empty((96,), dtype=torch.uint8, device="cuda")
    cv.msm_device(d_s, d_p, n, d_r[:64])
    ctx.synchronize()
    want = bytes(d_r[:64].cpu().numpy())
    hs, hp = d_s.cpu().numpy(), d_p.cpu().numpy()
    assert bytes(cv.msm(hs, hp)) == want                                              # pageable
    proj = bytes(cv.msm(hs, hp, out_format=ecgpu.PROJECTIVE))
    assert proj[:64] == want and proj[64:] == (1).to_bytes(32, "big")
    ps, pp = ctx.pinned_array((n, 32)), ctx.pinned_array((n, 64))
    ps[:] = hs
    pp[:] = hp
    assert bytes(cv.msm(ps, pp)) == want                                              # page-locked
    # all-identity parts: the sum of the first 2^22 + 5 terms with zero scalars is the identity, whatever the parts hold
    zs = np.zeros((1 << 22) + 5, dtype=np.uint8).reshape(-1, 1).repeat(32, axis=1)
    assert bytes(cv.msm(zs, hp[:len(zs)])) == bytes(64)
