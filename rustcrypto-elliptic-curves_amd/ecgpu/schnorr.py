"""BIP340 Schnorr over secp256k1 for batches: host side of k256/src/schnorr/{signing,verifying}.rs.

Host: tagged SHA-256 hashes (challenge, aux, nonce: k256/src/schnorr.rs:180-186, signing.rs:80-118) and scalar
arithmetic mod n (Python integers).  Device: every elliptic-curve operation - `Curve.schnorr_verify` for verification,
`Curve.mul_by_generator` for the public keys and nonce points of signing.  No CPU curve arithmetic exists here."""
from __future__ import annotations

import hashlib
from typing import Sequence

import numpy as np

from . import K256, Curve

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def tagged_hash(tag: bytes, *parts: bytes) -> bytes:
    t = hashlib.sha256(tag).digest()
    h = hashlib.sha256(t + t)
    for p in parts:
        h.update(p)
    return h.digest()


def challenges(pubkeys_x: Sequence[bytes], prehashes: Sequence[bytes], sigs: Sequence[bytes]) -> bytes:
    return b"".join(tagged_hash(b"BIP0340/challenge", s[:32], px, m) for px, m, s in zip(pubkeys_x, prehashes, sigs))


def verify_batch(curve: Curve, pubkeys_x: Sequence[bytes], prehashes: Sequence[bytes], sigs: Sequence[bytes]) -> np.ndarray:
    """ok[i] = 1 iff sigs[i] (64 bytes r || s) verifies over the 32-byte prehashes[i] under the x-only key."""
    if curve.id != K256:
        raise ValueError("BIP340 is defined over secp256k1")
    if any(len(m) != 32 for m in prehashes):
        raise ValueError("verify_prehash takes 32-byte digests")     # verifying.rs:68
    return curve.schnorr_verify(b"".join(pubkeys_x), b"".join(sigs), challenges(pubkeys_x, prehashes, sigs))


def sign_batch(curve: Curve, secret_keys: Sequence[bytes], prehashes: Sequence[bytes], aux_rands: Sequence[bytes]):
    """sign_prehash_with_aux_rand (signing.rs:80-131) for a batch -> (signatures, public keys x-only)."""
    if curve.id != K256:
        raise ValueError("BIP340 is defined over secp256k1")
    n = len(secret_keys)
    d0 = [int.from_bytes(k, "big") for k in secret_keys]
    if any(not (0 < d < N) for d in d0):
        raise ValueError("secret key out of range")
    P, _ = curve.mul_by_generator(b"".join(secret_keys))                       # device
    d = [(N - x) if (P[i][63] & 1) else x for i, x in enumerate(d0)]            # SigningKey::from: even-y key
    px = [bytes(P[i][:32]) for i in range(n)]
    ks = []
    for i in range(n):
        t = (d[i] ^ int.from_bytes(tagged_hash(b"BIP0340/aux", aux_rands[i]), "big")).to_bytes(32, "big")
        k0 = int.from_bytes(tagged_hash(b"BIP0340/nonce", t, px[i], prehashes[i]), "big") % N
        if k0 == 0:
            raise ValueError("zero nonce")
        ks.append(k0)
    R, _ = curve.mul_by_generator(b"".join(k.to_bytes(32, "big") for k in ks))  # device
    sigs = []
    for i in range(n):
        k = (N - ks[i]) if (R[i][63] & 1) else ks[i]
        r = bytes(R[i][:32])
        e = int.from_bytes(tagged_hash(b"BIP0340/challenge", r, px[i], prehashes[i]), "big") % N
        s = (k + e * d[i]) % N
        if s == 0:
            raise ValueError("zero s")
        sigs.append(r + s.to_bytes(32, "big"))
    return sigs, px
