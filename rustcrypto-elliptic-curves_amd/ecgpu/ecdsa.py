"""Batched ECDSA verification of DER-encoded signatures over messages: the shape of the reference's
`VerifyingKey::verify(msg, &Signature::from_der(..)?)` call (k256/src/ecdsa.rs:366-387), for a batch.

Host side: DER decoding (der.py), hashing (the curve's DigestPrimitive: SHA-256 for k256 / p256, SHA-384 for
p384 - k256/src/ecdsa.rs:177-180, p256/src/ecdsa.rs:66-69, p384/src/ecdsa.rs:63-66) and bits2field.  Device
side: `Curve.ecdsa_verify` (ecgpu_ecdsa_verify_batch).  No CPU verification path exists here."""
from __future__ import annotations

import hashlib
from typing import Sequence

import numpy as np

from . import K256, P256, P384, Curve
from .der import decode_signature

DIGEST = {K256: hashlib.sha256, P256: hashlib.sha256, P384: hashlib.sha384}


def bits2field(nb: int, digest: bytes) -> bytes:
    if len(digest) < nb // 2:
        raise ValueError("digest too short")
    return digest[:nb] if len(digest) >= nb else bytes(nb - len(digest)) + digest


def verify_der_batch(curve: Curve, pubkeys_xy: Sequence[bytes], messages: Sequence[bytes], der_sigs: Sequence[bytes],
                     normalize_s: bool = False) -> np.ndarray:
    """ok[i] = 1 iff der_sigs[i] decodes and verifies over messages[i] under pubkeys_xy[i] (affine x || y).
    normalize_s=True first maps s to the low half, as the k256 Wycheproof runner does (k256/src/ecdsa.rs:377)."""
    nb = curve.nb
    n = len(der_sigs)
    order = ORDER[curve.id]
    z = bytearray(n * nb)
    sig = bytearray(n * 2 * nb)
    decoded = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        rs = decode_signature(der_sigs[i], nb)
        if rs is None:
            continue                       # Signature::from_der -> Err
        r, s = rs
        if normalize_s and s > order // 2 and s < order:
            s = order - s
        decoded[i] = 1
        sig[2 * nb * i:2 * nb * (i + 1)] = r.to_bytes(nb, "big") + s.to_bytes(nb, "big")
        z[nb * i:nb * (i + 1)] = bits2field(nb, DIGEST[curve.id](messages[i]).digest())
    ok = curve.ecdsa_verify(bytes(z), bytes(sig), b"".join(pubkeys_xy))
    return ok & decoded


ORDER = {
    K256: 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    P256: 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551,
    P384: 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFC7634D81F4372DDF581A0DB248B0A77AECEC196ACCC52973,
}
