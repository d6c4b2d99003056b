"""ISCC codec against the reference's schema example strings (tests/golden/kat_codec.json)."""

import json
import os

import pytest

from iscc_search_amd import codec

with open(os.path.join(os.path.dirname(__file__), "golden", "kat_codec.json")) as f:
    KAT = json.load(f)


@pytest.mark.parametrize("case", KAT["units"], ids=lambda c: c["type"])
def test_unit_decode(case):
    u = codec.Iscc(case["iscc"])
    assert [u.mtype, u.stype, u.version, u.length] == case["header"]
    assert u.body.hex() == case["body"]
    assert u.unit_type == case["type"]
    assert len(u) == case["bits"]
    assert str(u) == case["iscc"]
    assert codec.encode_unit(u.mtype, u.stype, u.version, u.body) == case["iscc"]


def test_iscc_id_roundtrip():
    c = KAT["iscc_id"]
    obj = codec.validate_iscc_id(c["iscc"])
    assert [obj.mtype, obj.stype, obj.version, obj.length] == c["header"]
    assert obj.body.hex() == c["body"]
    assert codec.iscc_id_to_int(c["iscc"]) == c["int"]
    assert codec.iscc_id_from_int(c["int"], c["realm"]) == c["iscc"]
    realm1 = codec.iscc_id_from_int(c["int"], 1)
    assert codec.validate_iscc_id(realm1).stype == 1
    with pytest.raises(ValueError, match="Realm mismatch"):
        codec.validate_iscc_id(realm1, expected_realm=0)


@pytest.mark.parametrize("bad", ["", "MAIGIIFJRDGEQQAA", "ISCC:AAAUHBUDQUT3LPWR", "ISCC:MAIGIIFJRDGEQQ", "ISCC:!!!"])
def test_invalid_iscc_id_rejected(bad):
    with pytest.raises(ValueError):
        codec.validate_iscc_id(bad)


def test_code_decomposition_and_composition():
    c = KAT["code"]
    obj = codec.Iscc(c["iscc"])
    assert [obj.mtype, obj.stype, obj.version, obj.length] == c["header"]
    assert [str(u) for u in codec.code_units(c["iscc"])] == c["units"]
    assert codec.gen_iscc_code(c["units"]) == c["iscc"]
    assert codec.gen_iscc_code(list(reversed(c["units"]))) == c["iscc"]   # order independent
    c2 = KAT["code2"]
    units2 = codec.code_units(c2["iscc"])
    assert [u.unit_type for u in units2] == c2["unit_types"]
    assert codec.gen_iscc_code([str(u) for u in units2]) == c2["iscc"]


def test_wide_code_roundtrip():
    data = codec.encode_unit(codec.MT_DATA, 0, 0, bytes(range(16)))
    inst = codec.encode_unit(codec.MT_INSTANCE, 0, 0, bytes(range(16, 32)))
    code = codec.gen_iscc_code([data, inst], wide=True)
    obj = codec.Iscc(code)
    assert obj.stype == codec.ST_ISCC_WIDE and len(obj.body) == 32
    assert [str(u) for u in codec.code_units(code)] == [data, inst]
    narrow = codec.gen_iscc_code([data, inst], wide=False)
    assert codec.Iscc(narrow).stype == codec.ST_ISCC_SUM and len(codec.Iscc(narrow).body) == 16


def test_gen_iscc_code_rejects_invalid_combinations():
    u = KAT["code"]["units"]
    with pytest.raises(ValueError):
        codec.gen_iscc_code(u[:1])
    with pytest.raises(ValueError):
        codec.gen_iscc_code([u[0], u[2]])          # no DATA + INSTANCE
    image = codec.encode_unit(codec.MT_CONTENT, 1, 0, bytes(8))
    with pytest.raises(ValueError):
        codec.gen_iscc_code([u[1], image, u[3], u[4]])  # TEXT semantic + IMAGE content


def test_header_varnibbles_roundtrip():
    for vals in [(0, 0, 0, 0), (7, 7, 7, 7), (8, 0, 0, 1), (5, 71, 0, 72), (1, 2, 583, 584)]:
        hdr = codec.encode_header(*vals)
        assert codec.decode_header(hdr + b"\xab\xcd")[:4] == vals
    assert codec.decode_header(codec.encode_header(3, 0, 0, 1) + b"\x01\x02")[4] == b"\x01\x02"


def test_base64_accepts_urlsafe_and_standard():
    raw = bytes(range(250, 256)) + b"\xfb\xff"
    s = codec.encode_base64(raw)
    assert "=" not in s and "+" not in s and "/" not in s
    assert codec.decode_base64(s) == raw
    assert codec.decode_base64(s.replace("-", "+").replace("_", "/") + "=") == raw


def test_new_iscc_id_is_valid_and_time_ordered():
    a = codec.new_iscc_id()
    b = codec.new_iscc_id()
    assert codec.validate_iscc_id(a).stype == 0
    assert codec.iscc_id_to_int(a) >> 12 <= codec.iscc_id_to_int(b) >> 12


def test_base32_decode_gives_the_verdicts_of_the_standard_library():
    """``decode_base32`` converts well-formed unpadded strings through one big integer; value and verdict must be ``base64.b32decode``'s."""
    import base64
    import math
    import random

    from iscc_search_amd import codec

    def reference(code):
        pad = math.ceil(len(code) / 8) * 8 - len(code)
        return base64.b32decode(code + "=" * pad, casefold=True)

    rnd = random.Random(5)
    for _ in range(3000):
        data = bytes(rnd.getrandbits(8) for _ in range(rnd.randint(0, 40)))
        text = base64.b32encode(data).decode().rstrip("=")
        assert codec.decode_base32(text) == data and codec.decode_base32(text.lower()) == data
    for _ in range(20000):
        text = "".join(rnd.choice("ABCDEFGHIJKLMNOPQRSTUVWXYZ234567abcxyz01 89=!_-\nÄ") for _ in range(rnd.randint(0, 12)))
        try:
            want = ("ok", reference(text))
        except Exception:
            want = ("error",)
        try:
            got = ("ok", codec.decode_base32(text))
        except ValueError:
            got = ("error",)
        assert got == want, repr(text)


def test_header_fast_paths_equal_the_general_form():
    import itertools

    from iscc_search_amd import codec

    for fields in itertools.product((0, 1, 5, 7, 8, 9, 71, 72), repeat=4):
        bits = "".join(codec._encode_varnibble_bits(v) for v in fields)
        bits += "0" * (-len(bits) % 8)
        general = int(bits, 2).to_bytes(len(bits) // 8, "big")
        assert codec.encode_header(*fields) == general
        assert codec.decode_header(general + b"xy")[:4] == fields
