"""Strict ASN.1 DER decoding of an ECDSA signature, the host-side step in front of the batch verifier.

Mirrors what `ecdsa::der::Signature::from_der` (ecdsa 0.16.9, used by the Wycheproof runners
k256/src/ecdsa.rs:376-380) accepts: SEQUENCE { INTEGER r, INTEGER s } with definite minimal lengths, minimal
non-negative integers, no trailing bytes; r and s must fit the curve's scalar width.  Returns None on any
violation (the reference returns Err)."""


def _read_len(b: bytes, pos: int):
    if pos >= len(b):
        return None
    first = b[pos]
    pos += 1
    if first < 0x80:
        return first, pos
    nb = first & 0x7F
    if nb == 0 or nb > 4 or pos + nb > len(b):      # indefinite or absurd
        return None
    val = int.from_bytes(b[pos:pos + nb], "big")
    if b[pos] == 0 or val < 0x80:                    # not minimal
        return None
    return val, pos + nb


def _read_uint(b: bytes, pos: int, end: int):
    if pos >= end or b[pos] != 0x02:
        return None
    got = _read_len(b, pos + 1)
    if got is None:
        return None
    ln, pos = got
    if ln == 0 or pos + ln > end:
        return None
    body = b[pos:pos + ln]
    if body[0] & 0x80:                               # negative
        return None
    if ln > 1 and body[0] == 0 and not (body[1] & 0x80):   # superfluous leading zero
        return None
    return int.from_bytes(body, "big"), pos + ln


def decode_signature(der: bytes, scalar_bytes: int):
    """-> (r, s) as integers, or None."""
    if len(der) < 2 or der[0] != 0x30:
        return None
    got = _read_len(der, 1)
    if got is None:
        return None
    ln, pos = got
    end = pos + ln
    if end != len(der):
        return None
    got = _read_uint(der, pos, end)
    if got is None:
        return None
    r, pos = got
    got = _read_uint(der, pos, end)
    if got is None:
        return None
    s, pos = got
    if pos != end:
        return None
    if r >> (8 * scalar_bytes) or s >> (8 * scalar_bytes):
        return None
    return r, s
