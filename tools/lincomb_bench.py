"""Linear combinations of 3 .. 1024 terms (ecgpu_lincomb_batch): the shared-doubling schedule of csrc/straus.hpp against the
term-by-term form it replaced (ECGPU_OPT_LINCOMB_TERM_BY_TERM = 1: one reference-schedule multiplication per term), device-resident.
Usage: python tools/lincomb_bench.py [log2 of the total number of terms]   -> rows for profiles/r04_secondary_entry_points.txt"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import torch
import ecgpu
from oracle import synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
curves = sys.argv[2].split(",") if len(sys.argv) > 2 else ["k256", "p256", "p384"]
ctx = ecgpu.Context(0)
vp = ctypes.c_void_p
for cn in curves:
    cv = ctx.curve(cn)
    nb = cv.nb
    total = 1 << (lg if cn != "p384" else lg - 2)
    u8 = dict(dtype=torch.uint8, device="cuda")
    s = torch.empty((total, nb), **u8); p = torch.empty((total, 2 * nb), **u8)
    cv.synth_scalars_device(s, total, synth.SEED, 0); cv.synth_points_device(p, total, synth.SEED, 0)
    ctx.synchronize()
    for terms in (3, 5, 16, 64, 1024):
        n = total // terms
        o = torch.empty((n, 2 * nb), **u8); f = torch.empty((n,), **u8)
        o2 = torch.empty((n, 2 * nb), **u8); f2 = torch.empty((n,), **u8)
        res = {}
        for mode, oo, ff in (("shared doublings", o, f), ("term by term", o2, f2)):
            ctx.set_option(ecgpu.OPT_LINCOMB_TERM_BY_TERM, 1 if mode == "term by term" else 0)
            best = 1e9
            for rep in range(3 if mode == "shared doublings" else 2):
                ctx.timer_start()
                ctx.check(ctx.lib.ecgpu_lincomb_batch(ctx.handle, cv.id, vp(s.data_ptr()), vp(p.data_ptr()), 0, terms, vp(oo.data_ptr()), 0, vp(ff.data_ptr()), n, 1, 0))
                best = min(best, ctx.timer_stop())
            res[mode] = best
        ctx.set_option(ecgpu.OPT_LINCOMB_TERM_BY_TERM, 0)
        same = bool(torch.equal(o, o2) and torch.equal(f, f2))
        a, b = res["shared doublings"], res["term by term"]
        print(f"{cn:5s} lincomb {terms:5d} terms x {n:8d}: shared doublings {a:9.3f} ms = {n * terms / a / 1e3:8.2f} M terms/s ({n / a / 1e3:8.3f} M combinations/s)   "
              f"term by term {b:9.3f} ms = {n * terms / b / 1e3:7.2f} M terms/s   x{b / a:5.2f}   same bytes: {same}", flush=True)
        del o, f, o2, f2
    del s, p
    torch.cuda.empty_cache()
ctx.close()
