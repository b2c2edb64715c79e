"""Pins the oracle's ECDSA restatement (and the host-side DER decoder) to the reference's own vectors:
<curve>/src/test_vectors/ecdsa.rs (d, k, m -> r, s) and the Wycheproof blobs (accept / reject)."""
import hashlib
import json
import os

import pytest

from oracle import ecmodel as M
from ecgpu import der

HERE = os.path.dirname(os.path.abspath(__file__))
HASH = {"k256": hashlib.sha256, "p256": hashlib.sha256, "p384": hashlib.sha384}


def wycheproof(cn):
    with open(os.path.join(HERE, "golden", f"wycheproof_{cn}.json")) as f:
        return json.load(f)["rows"]


def padded(c, hx):
    """element_from_padded_slice of the runner: left-pad short input, strip zero bytes of long input."""
    b = bytes.fromhex(hx)
    if len(b) >= c.nbytes:
        assert not any(b[:len(b) - c.nbytes])
        return int.from_bytes(b[len(b) - c.nbytes:], "big")
    return int.from_bytes(b, "big")


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_sign_and_verify_kats(cn, ref_vectors):
    c = M.CURVES[cn]
    for v in ref_vectors[cn]["ecdsa"]:
        d, k = int(v["d"], 16), int(v["k"], 16)
        z = M.bits2field(c, bytes.fromhex(v["m"]))
        r, s, _ = M.ecdsa_sign_prehashed(c, d, k, z, normalize_s=False)
        assert (r, s) == (int(v["r"], 16), int(v["s"], 16))
        Q = (int(v["q_x"], 16), int(v["q_y"], 16))
        assert M.ecdsa_verify_prehashed(c, Q, z, r, s)
        assert not M.ecdsa_verify_prehashed(c, Q, z, r, (s + 1) % c.n)


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_wycheproof(cn):
    c = M.CURVES[cn]
    n_pass = 0
    for i, (wx, wy, msg, sig, ok) in enumerate(wycheproof(cn)):
        Q = (padded(c, wx), padded(c, wy))
        assert M.on_curve(c, Q)
        rs = der.decode_signature(bytes.fromhex(sig), c.nbytes)
        if rs is None:
            assert not ok, f"row {i}: DER rejected but the vector passes"
            continue
        r, s = rs
        if cn == "k256" and s > c.n // 2:
            s = c.n - s                   # the k256 runner normalises s first (k256/src/ecdsa.rs:377)
        z = M.bits2field(c, HASH[cn](bytes.fromhex(msg)).digest())
        got = M.ecdsa_verify_prehashed(c, Q, z, r, s, reject_high_s=(cn == "k256"))
        assert got == bool(ok), f"row {i}"
        n_pass += got
    assert n_pass > 100


@pytest.mark.parametrize("cn", ["k256", "p256", "p384"])
def test_c_oracle_matches_model_and_vectors(cn, ref_vectors):
    """oracle/ecoracle.c's ECDSA (the cpu_baseline of the ECDSA bench workloads) against the KATs, the
    Wycheproof flags and the big-integer model on corrupted inputs."""
    import numpy as np
    from oracle import coracle as CO
    c = M.CURVES[cn]
    cid = M.CURVE_IDS[cn]
    nb = c.nbytes
    tob = lambda v: v.to_bytes(nb, "big")
    vs = ref_vectors[cn]["ecdsa"]
    d = np.frombuffer(b"".join(bytes.fromhex(v["d"]) for v in vs), dtype=np.uint8)
    k = np.frombuffer(b"".join(bytes.fromhex(v["k"]) for v in vs), dtype=np.uint8)
    z = np.frombuffer(b"".join(M.bits2field(c, bytes.fromhex(v["m"])) for v in vs), dtype=np.uint8)
    sig, rec, ok = CO.ecdsa_sign_batch(cid, d, k, z)
    assert ok.all()
    for i, v in enumerate(vs):
        assert bytes(sig[i]).hex() == v["r"] + v["s"]
    rows = wycheproof(cn)
    zs, sgs, qs, want = [], [], [], []
    for wx, wy, msg, sg, flag in rows:
        rs = der.decode_signature(bytes.fromhex(sg), nb)
        if rs is None:
            continue
        r, s = rs
        if cn == "k256" and c.n // 2 < s < c.n:
            s = c.n - s
        zs.append(M.bits2field(c, HASH[cn](bytes.fromhex(msg)).digest()))
        sgs.append(tob(r) + tob(s))
        qs.append(tob(padded(c, wx)) + tob(padded(c, wy)))
        want.append(flag)
    got = CO.ecdsa_verify_batch(cid, np.frombuffer(b"".join(zs), dtype=np.uint8), np.frombuffer(b"".join(sgs), dtype=np.uint8),
                                np.frombuffer(b"".join(qs), dtype=np.uint8), low_s=(cn == "k256"))
    assert list(got) == want


def test_bip340_vectors_pin_the_schnorr_model(ref_vectors):
    v = ref_vectors["k256"]["bip340"]
    assert len(v["sign"]) == 4 and len(v["verify"]) >= 10
    for s in v["sign"]:
        sig, px = M.schnorr_sign_prehash(bytes.fromhex(s["secret_key"]), bytes.fromhex(s["message"]), bytes.fromhex(s["aux_rand"]))
        assert px.hex() == s["public_key"] and sig.hex() == s["signature"]
        assert M.schnorr_verify_prehash(px, bytes.fromhex(s["message"]), sig)
    for t in v["verify"]:
        assert M.schnorr_verify_prehash(bytes.fromhex(t["public_key"]), bytes.fromhex(t["message"]), bytes.fromhex(t["signature"])) == t["valid"], t["index"]


def test_recovery_vectors_pin_the_model(ref_vectors):
    c = M.K256
    for v in ref_vectors["k256"]["recovery"]:
        z = hashlib.sha256(v["msg"].encode()).digest()
        sig = bytes.fromhex(v["sig"])
        Q = M.ecdsa_recover_prehashed(c, z, int.from_bytes(sig[:32], "big"), int.from_bytes(sig[32:], "big"), v["recid"], reject_high_s=True)
        assert M.group_to_bytes(c, Q).hex() == v["pk"]
