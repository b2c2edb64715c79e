"""Randomised cross-check of the many-term linear combination (csrc/straus.hpp: dynamic pass draws, balanced groups, second-stage fold) against the
term-by-term form (ECGPU_OPT_LINCOMB_TERM_BY_TERM) and, for secp256k1, of the exact-(X, Y, Z) schedule's affine form: random term counts 3 .. 1024, random
numbers of combinations around the pass / lane boundaries, zero scalars, identity points, repeated and cancelling terms.   python tools/lincomb_stress.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import numpy as np
import ecgpu
from oracle import coracle as CO
from oracle import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
ctx = ecgpu.Context(0)
bad = 0
for case in range(cases):
    cn = ["k256", "p256", "p384"][case % 3]
    cid = ecgpu.CURVE_IDS[cn]
    cv = ctx.curve(cn)
    nb = cv.nb
    terms = int(rng.choice([3, 4, 5, 7, 8, 15, 16, 17, 31, 33, 64, 100, 255, 1024]))
    budget = 400_000 if cn != "p384" else 120_000
    n = max(1, int(rng.integers(1, max(2, budget // terms))))
    if rng.random() < 0.3:
        n = int(rng.choice([1, 2, 63, 64, 65, 255, 257]))
    tot = n * terms
    s = CO.synth_scalars(cid, tot, synth.SEED + case, 0)
    p = CO.synth_points(cid, tot, synth.SEED + case, 0)
    for _ in range(min(50, tot)):
        j = int(rng.integers(0, tot))
        r = rng.random()
        if r < 0.3:
            s[j] = 0
        elif r < 0.6:
            p[j] = 0
        elif j + 1 < tot and (j + 1) % terms:
            p[j + 1] = p[j]                         # the same point twice in one combination
            if r < 0.8:
                s[j + 1] = s[j]
    out, inf = cv.lincomb(s, p, terms=terms)
    ctx.set_option(ecgpu.OPT_LINCOMB_TERM_BY_TERM, 1)
    o2, i2 = cv.lincomb(s, p, terms=terms)
    ctx.set_option(ecgpu.OPT_LINCOMB_TERM_BY_TERM, 0)
    ok = bytes(out) == bytes(o2) and bytes(inf) == bytes(i2)
    if cn == "k256" and tot <= 60_000:
        o3, i3 = cv.lincomb(s, p, terms=terms, flags=ecgpu.EXACT_REFERENCE)
        ok = ok and bytes(o3) == bytes(out) and bytes(i3) == bytes(inf)
    bad += 0 if ok else 1
    print(f"case {case:3d} {cn} terms={terms:5d} n={n:7d} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
