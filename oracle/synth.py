"""Synthetic-input specification shared by tests, bench.py, the C oracle and the HIP generator.

TEST INFRASTRUCTURE (see oracle/ecmodel.py header).  The same streams are produced on the
device by `ecgpu_synth_scalars` / `ecgpu_synth_points` (csrc/synth.hip) and in C by
oracle/ecoracle.c, so that any index of a 2^24-element batch can be regenerated on the CPU
without ever materialising the batch there.

Stream definition (counter based, random access):

    word(seed, stream, index, j) = splitmix64_finalise(
            seed ^ (stream * 0xD1342543DE82EF95)  +  (8*index + j + 1) * 0x9E3779B97F4A7C15 )
    value(seed, stream, index)   = word_0 || word_1 || ... || word_{L-1}   (big endian, L = nbytes/8)

* scalar_i  = reduce(value(seed, 0, i))                reduce(w) = w - n if w >= n
              (mirrors Reduce<U256>::reduce, k256/src/arithmetic/scalar.rs:700-713)
* point_i   = try-and-increment: for t = 0, 1, ...:   x = value(seed, 1 + t, i) mod-reduced the same
              way against p; y_is_odd = word(seed, 1 + t, i, 7) & 1; accept the first x for which
              decompress(x, y_is_odd) exists (k256 affine.rs:184-202 / primeorder affine.rs:129-150).
"""
from __future__ import annotations

from . import ecmodel as M

SEED = 0xEC5CA1A5
MASK = (1 << 64) - 1
MAX_TRIES = 64


def _mix(z: int) -> int:
    z &= MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def word(seed: int, stream: int, index: int, j: int) -> int:
    z = ((seed ^ ((stream * 0xD1342543DE82EF95) & MASK)) + (8 * index + j + 1) * 0x9E3779B97F4A7C15) & MASK
    return _mix(z)


def value(seed: int, stream: int, index: int, nwords: int) -> int:
    v = 0
    for j in range(nwords):
        v = (v << 64) | word(seed, stream, index, j)
    return v


def scalar(c: M.Curve, index: int, seed: int = SEED) -> int:
    w = value(seed, 0, index, c.nbytes // 8)
    return w - c.n if w >= c.n else w


def point(c: M.Curve, index: int, seed: int = SEED):
    for t in range(MAX_TRIES):
        x = value(seed, 1 + t, index, c.nbytes // 8)
        if x >= c.p:
            x -= c.p
        odd = word(seed, 1 + t, index, 7) & 1
        P = M.decompress(c, x, odd)
        if P is not None:
            return P
    raise RuntimeError("no point found")


def scalars(c: M.Curve, n: int, seed: int = SEED, start: int = 0):
    return [scalar(c, start + i, seed) for i in range(n)]


def points(c: M.Curve, n: int, seed: int = SEED, start: int = 0):
    return [point(c, start + i, seed) for i in range(n)]
