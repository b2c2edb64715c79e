#!/usr/bin/env python3
"""Scrape the reference's own known-answer vectors for the hot path into JSON fixtures.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_fixtures.py

Only DATA is extracted (hex literals: inputs and expected outputs); no reference source text is
kept.  Sources (relative to /root/reference), see SURVEY.md section 8c:

  <curve>/src/test_vectors/group.rs     ADD_TEST_VECTORS (k*G for k = 1..20), MUL_TEST_VECTORS (k, x, y)
  <curve>/src/test_vectors/ecdsa.rs     d -> (q_x, q_y), k -> r   (fixed-base KATs)
  <curve>/src/arithmetic/hash2curve.rs  Q0 + Q1 = P triples       (arbitrary-point addition KATs)
  {k256,p256}/src/test_vectors/field.rs DBL_TEST_VECTORS          (repeated doubling of 1)
  k256/src/arithmetic/field/field_8x32_risc0.rs:225-303           (add / negate / mul / square KATs)
  k256/src/arithmetic/field/field_5x52.rs:510-544                 (2^256 normalises to 0x1000003d1)

Also writes the config-1 fixture (BASELINE.json configs[0]): 1024 seeded k256 (scalar, point)
pairs with the expected affine outputs computed by oracle/ecmodel.py (the model that the vectors
above pin), and matching smaller sets for p256 / p384.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import ecmodel as M  # noqa: E402
from oracle import synth  # noqa: E402

HEX = re.compile(r'hex!\(\s*"([0-9A-Fa-f]+)"\s*\)')


def read(path):
    with open(os.path.join(REF, path)) as f:
        return f.read()


def const_block(text, name):
    """Text of `pub const NAME ... = &[ ... ];`"""
    i = text.index(name)
    j = text.index("];", i)
    return text[i:j]


def group_vectors(curve):
    t = read(f"{curve}/src/test_vectors/group.rs")
    add = HEX.findall(const_block(t, "ADD_TEST_VECTORS"))
    mul = HEX.findall(const_block(t, "MUL_TEST_VECTORS"))
    assert len(add) % 2 == 0 and len(mul) % 3 == 0
    return {
        "add": [[add[i], add[i + 1]] for i in range(0, len(add), 2)],
        "mul": [[mul[i], mul[i + 1], mul[i + 2]] for i in range(0, len(mul), 3)],
    }


def ecdsa_vectors(curve):
    t = read(f"{curve}/src/test_vectors/ecdsa.rs")
    out = []
    for blk in t.split("TestVector {")[1:]:
        fields = dict(re.findall(r'(\w+):\s*&hex!\(\s*"([0-9A-Fa-f]+)"\s*\)', blk))
        out.append({k: fields[k] for k in ("d", "q_x", "q_y", "k", "m", "r", "s")})
    return out


def _read_vlq(d, pos):
    b = d[pos]
    pos += 1
    val = b & 0x7F
    while b & 0x80:
        b = d[pos]
        pos += 1
        val = ((val + 1) << 7) + (b & 0x7F)
    return val, pos


def wycheproof_rows(curve):
    """<curve>/src/test_vectors/data/wycheproof.blb: a blobby file (de-duplicated blob table, then entries that
    are either a table reference or an inline blob), five blobs per row: wx, wy, msg, DER signature, pass flag
    (runner: k256/src/ecdsa.rs:340-425)."""
    with open(os.path.join(REF, f"{curve}/src/test_vectors/data/wycheproof.blb"), "rb") as f:
        d = f.read()
    n, pos = _read_vlq(d, 0)
    table = []
    for _ in range(n):
        ln, pos = _read_vlq(d, pos)
        table.append(d[pos:pos + ln])
        pos += ln
    items = []
    while pos < len(d):
        v, pos = _read_vlq(d, pos)
        if v & 1:
            items.append(table[v >> 1])
        else:
            items.append(d[pos:pos + (v >> 1)])
            pos += v >> 1
    assert len(items) % 5 == 0
    return [[items[i].hex(), items[i + 1].hex(), items[i + 2].hex(), items[i + 3].hex(), items[i + 4][0]]
            for i in range(0, len(items), 5)]


def encoding_vectors(curve):
    """SEC1 encodings of the base point held by the affine tests (k256/src/arithmetic/affine.rs:374-381,
    p256/tests/affine.rs:12-31, p384/tests/affine.rs)."""
    t = read(f"{curve}/src/arithmetic/affine.rs" if curve == "k256" else f"{curve}/tests/affine.rs")
    out = {}
    for name in ("UNCOMPRESSED_BASEPOINT", "COMPRESSED_BASEPOINT", "COMPACT_BASEPOINT", "UNCOMPACT_BASEPOINT"):
        m = re.search(r"const %s: &\[u8\] =\s*&hex!\(\s*((?:\"[^\"]*\"\s*)+)\)" % name, t)
        if m:
            out[name.lower()] = re.sub(r"[^0-9A-Fa-f]", "", m.group(1)).lower()
    return out


def bip340_vectors():
    """k256/src/schnorr.rs:217-270 (sign vectors 0-3) and :307-430 (verify vectors 4-14)."""
    t = read("k256/src/schnorr.rs")
    t = t[t.index("mod tests"):]

    def hexes(blk, name):
        m = re.search(name + r":\s*hex!\(\s*((?:\"[^\"]*\"\s*)+)\)", blk)
        return re.sub(r"[^0-9A-Fa-f]", "", m.group(1)).lower()

    sign, verify = [], []
    a, b = t.index("const BIP340_SIGN_VECTORS"), t.index("const BIP340_VERIFY_VECTORS")
    for blk in t[a:b].split("SignVector {")[1:]:
        if "secret_key: hex!" not in blk:
            continue
        sign.append({k: hexes(blk, k) for k in ("secret_key", "public_key", "aux_rand", "message", "signature")})
    for blk in t[b:].split("VerifyVector {")[1:]:
        if "public_key: hex!" not in blk:
            continue
        v = {k: hexes(blk, k) for k in ("public_key", "message", "signature")}
        v["valid"] = bool(re.search(r"valid:\s*true", blk))
        v["index"] = int(re.search(r"index:\s*(\d+)", blk).group(1))
        verify.append(v)
    return {"sign": sign, "verify": verify}


def recovery_vectors():
    """k256/src/ecdsa.rs:277-298: compressed public key, message, signature, recovery id (y_is_odd, x_reduced)."""
    t = read("k256/src/ecdsa.rs")
    t = t[t.index("const RECOVERY_TEST_VECTORS"):t.index("fn public_key_recovery")]
    out = []
    for blk in t.split("RecoveryTestVector {")[1:]:
        pk = re.search(r'pk:\s*hex!\("([0-9a-fA-F]+)"\)', blk).group(1)
        msg = re.search(r'msg:\s*b"([^"]*)"', blk).group(1)
        sig = re.sub(r"[^0-9A-Fa-f]", "", re.search(r"sig:\s*hex!\(\s*((?:\"[^\"]*\"\s*)+)\)", blk).group(1))
        rid = re.search(r"RecoveryId::new\((true|false),\s*(true|false)\)", blk)
        out.append({"pk": pk.lower(), "msg": msg, "sig": sig.lower(), "recid": (1 if rid.group(1) == "true" else 0) | (2 if rid.group(2) == "true" else 0)})
    return out


def h2c_vectors(curve):
    """RFC 9380 vectors of <curve>/src/arithmetic/hash2curve.rs: message, u_0, u_1 (hash_to_field), Q0 = map(u_0),
    Q1 = map(u_1), P = Q0 + Q1; the domain separation tag is kept with them."""
    t = read(f"{curve}/src/arithmetic/hash2curve.rs")
    dst = re.search(r'const DST: &\[u8\] = b"([^"]*)"', t).group(1)
    i = t.index("const TEST_VECTORS")
    out = []
    for blk in t[i:].split("TestVector {")[1:]:
        fields = dict(re.findall(r'(\w+):\s*hex!\(\s*"([0-9A-Fa-f]+)"\s*\)', blk))
        if "q0_x" not in fields:
            continue
        v = {k: fields[k] for k in ("p_x", "p_y", "u_0", "u_1", "q0_x", "q0_y", "q1_x", "q1_y")}
        v["msg"] = re.search(r'msg:\s*b"([^"]*)"', blk).group(1)
        v["dst"] = dst
        out.append(v)
    return out


def field_dbl(curve):
    t = read(f"{curve}/src/test_vectors/field.rs")
    return HEX.findall(const_block(t, "DBL_TEST_VECTORS"))


def risc0_field_kats():
    t = read("k256/src/arithmetic/field/field_8x32_risc0.rs")
    t = t[t.index("mod tests"):]
    hexes = re.findall(r'hex!\(\s*"([0-9A-Fa-f]+)"\s*\)', t)
    # order of appearance in the test module: VAL_A, VAL_B, add, add_negated, negate, mul, square
    a, b, add, add_neg, neg, mul, sqr = hexes[:7]
    return {"a": a, "b": b, "add": add, "add_negated": add_neg, "negate_a": neg, "mul": mul, "square_a": sqr}


def config_fixture(curve_name, n, seed):
    c = M.CURVES[curve_name]
    scalars = synth.scalars(c, n, seed)
    points = synth.points(c, n, seed)
    rows = []
    for k, P in zip(scalars, points):
        R = M.to_affine(c, M.mul_ref(c, (P[0], P[1], 1), k))
        want = M.affine_mul(c, k, P)
        assert (want is None and R[2] == 1) or (want == (R[0], R[1]))
        rows.append([M.i2b(c, k).hex(), M.i2b(c, P[0]).hex(), M.i2b(c, P[1]).hex(),
                     M.affine_bytes(c, R).hex()])
    return {"curve": curve_name, "seed": seed, "n": n,
            "columns": ["scalar", "px", "py", "out_affine(x||y||inf)"], "rows": rows}


def main():
    fx = {}
    for c in ("k256", "p256", "p384"):
        fx[c] = {"group": group_vectors(c), "ecdsa": ecdsa_vectors(c), "hash2curve": h2c_vectors(c), "encoding": encoding_vectors(c)}
    fx["k256"]["bip340"] = bip340_vectors()
    fx["k256"]["recovery"] = recovery_vectors()
    fx["k256"]["field_dbl"] = field_dbl("k256")
    fx["p256"]["field_dbl"] = field_dbl("p256")
    fx["k256"]["field_kat"] = risc0_field_kats()
    # field_5x52.rs:510-544 overflow_check_after_weak_normalize: value 2^256 -> 0x1000003d1
    fx["k256"]["field_2pow256"] = "%064x" % 0x1000003D1
    counts = {c: {k: (len(v) if isinstance(v, list) else {kk: len(vv) for kk, vv in v.items()} if isinstance(v, dict) else 1)
                  for k, v in fx[c].items()} for c in fx}
    print(json.dumps(counts, indent=1))
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump(fx, f, indent=0, sort_keys=True)
    for c in ("k256", "p256", "p384"):
        rows = wycheproof_rows(c)
        with open(os.path.join(HERE, f"wycheproof_{c}.json"), "w") as f:
            json.dump({"curve": c, "columns": ["wx", "wy", "msg", "der_sig", "pass"], "rows": rows}, f, indent=0)
        print("wycheproof", c, len(rows))
    cfg = {
        "k256": config_fixture("k256", 1024, synth.SEED),
        "p256": config_fixture("p256", 128, synth.SEED),
        "p384": config_fixture("p384", 64, synth.SEED),
    }
    for name, d in cfg.items():
        with open(os.path.join(HERE, f"config1_{name}.json"), "w") as f:
            json.dump(d, f, indent=0)
    print("wrote fixtures")


if __name__ == "__main__":
    main()
