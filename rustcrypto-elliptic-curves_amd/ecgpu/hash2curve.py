"""GroupDigest::hash_from_bytes for batches (RFC 9380 *_XMD:SHA-*_SSWU_RO_ suites of k256 / p256 / p384).

Host: expand_message_xmd and hash_to_field (SHA-2 and a reduction mod p per element - byte glue).  Device: the two
map_to_curve evaluations and their sum (`Curve.map_to_curve`, ecgpu_map_to_curve_batch)."""
from __future__ import annotations

import hashlib
from typing import Sequence

from . import K256, P256, P384, Curve

MODULUS = {
    K256: 2**256 - 2**32 - 977,
    P256: 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF,
    P384: 2**384 - 2**128 - 2**96 + 2**32 - 1,
}
HASH = {K256: "sha256", P256: "sha256", P384: "sha384"}
OKM_LEN = {K256: 48, P256: 48, P384: 72}          # FromOkm::Length


def expand_message_xmd(hash_name: str, msg: bytes, dst: bytes, length: int) -> bytes:
    h = lambda b: hashlib.new(hash_name, b).digest()
    b_in, s_in = hashlib.new(hash_name).digest_size, hashlib.new(hash_name).block_size
    ell = -(-length // b_in)
    if ell > 255 or len(dst) > 255:
        raise ValueError("expand_message_xmd: output or DST too long")
    dst_prime = dst + bytes([len(dst)])
    b0 = h(bytes(s_in) + msg + length.to_bytes(2, "big") + b"\x00" + dst_prime)
    blocks = [h(b0 + b"\x01" + dst_prime)]
    for i in range(2, ell + 1):
        blocks.append(h(bytes(x ^ y for x, y in zip(b0, blocks[-1])) + bytes([i]) + dst_prime))
    return b"".join(blocks)[:length]


def hash_to_field(curve: Curve, msg: bytes, dst: bytes, count: int = 2) -> bytes:
    L, p = OKM_LEN[curve.id], MODULUS[curve.id]
    okm = expand_message_xmd(HASH[curve.id], msg, dst, count * L)
    return b"".join((int.from_bytes(okm[L * i:L * (i + 1)], "big") % p).to_bytes(curve.nb, "big") for i in range(count))


def hash_from_bytes(curve: Curve, msgs: Sequence[bytes], dst: bytes):
    """-> (points_xy, inf): one curve point per message."""
    u = b"".join(hash_to_field(curve, m, dst, 2) for m in msgs)
    return curve.map_to_curve(u, count=2)
