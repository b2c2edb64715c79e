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
    assert lib.ecgpu_msm(h, 1, p, p, 0, 1, p, 0, ecgpu.HOST) == -4                        # MSM is k256 only
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
