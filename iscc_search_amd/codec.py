"""
Minimal ISCC codec: just what the ``hip:///`` backend needs to derive keys, table names and code
bodies from canonical ISCC strings.

The reference delegates all of this to the third-party ``iscc-core 1.3.0`` (``uv.lock:651-652``),
which is not available here.  This is an own implementation of the published ISCC header layout
(ISO 24138): a header of four variable-length nibbles (MainType, SubType, Version, Length) followed
by the body, base32 (RFC 4648, no padding) for codes and base64url (no padding) for simprints.
Pinned by the example strings in the reference's schema (``iscc_search/schema.py:100-124``, decoded
in SURVEY.md section 8c) -- see ``tests/golden/kat_codec.json``.

Mirrors (behaviour, not code) of the reference's use sites:
  ``IsccBase``/``IsccID``/``IsccUnit``/``IsccCode``  ``iscc_search/models.py:68-316``
  ``normalize_query`` inputs                       ``iscc_search/indexes/common.py:275-330``
"""

import base64
import math
import random
import time

# MainTypes (header nibble 1)
MT_META, MT_SEMANTIC, MT_CONTENT, MT_DATA, MT_INSTANCE, MT_ISCC, MT_ID, MT_FLAKE = range(8)
MT_NAMES = ("META", "SEMANTIC", "CONTENT", "DATA", "INSTANCE", "ISCC", "ID", "FLAKE")

# SubType name tables selected by (MainType, Version)
_ST_NONE = ("NONE",)
_ST_CC = ("TEXT", "IMAGE", "AUDIO", "VIDEO", "MIXED")
_ST_ISCC = ("TEXT", "IMAGE", "AUDIO", "VIDEO", "MIXED", "SUM", "NONE", "WIDE")
_ST_ID = ("PRIVATE", "BITCOIN", "ETHEREUM", "POLYGON")
_ST_ID_REALM = ("REALM_0", "REALM_1")
ST_ISCC_SUM, ST_ISCC_NONE, ST_ISCC_WIDE = 5, 6, 7

_SUBTYPE_NAMES = {
    (MT_META, 0): _ST_NONE,
    (MT_SEMANTIC, 0): _ST_CC,
    (MT_CONTENT, 0): _ST_CC,
    (MT_DATA, 0): _ST_NONE,
    (MT_INSTANCE, 0): _ST_NONE,
    (MT_ISCC, 0): _ST_ISCC,
    (MT_ID, 0): _ST_ID,
    (MT_ID, 1): _ST_ID_REALM,
    (MT_FLAKE, 0): _ST_NONE,
}

# unit combinations an ISCC-CODE can carry besides DATA + INSTANCE, indexed by the header's length field
_UNITS = (
    (),
    (MT_CONTENT,),
    (MT_SEMANTIC,),
    (MT_SEMANTIC, MT_CONTENT),
    (MT_META,),
    (MT_META, MT_CONTENT),
    (MT_META, MT_SEMANTIC),
    (MT_META, MT_SEMANTIC, MT_CONTENT),
)


# -- text encodings ------------------------------------------------------------------------------
def encode_base32(data):
    # type: (bytes) -> str
    return base64.b32encode(data).decode("ascii").rstrip("=")


# RFC 4648 digits -> the digits of int(x, 32); every other ASCII character -> "!" (no digit: int() refuses the string)
_B32_DIGITS = {i: "!" for i in range(128)}
_B32_DIGITS.update(str.maketrans("ABCDEFGHIJKLMNOPQRSTUVWXYZ234567abcdefghijklmnopqrstuvwxyz", "0123456789abcdefghijklmnopqrstuv0123456789abcdefghijklmnop"))
_B32_LENGTHS = (True, False, True, False, True, True, False, True)      # unpadded lengths mod 8 that are whole bytes + allowed slack


def decode_base32(code):
    # type: (str) -> bytes
    # (five decodes per search request: base64.b32decode -- Python code in CPython 3.10 -- takes 2.6 us each, one big-integer
    #  conversion 0.6; anything but a well-formed unpadded string goes to the library call below and gets its verdict)
    n = len(code)
    if n and _B32_LENGTHS[n & 7] and code.isascii():
        try:
            return (int(code.translate(_B32_DIGITS), 32) >> (5 * n & 7)).to_bytes(5 * n >> 3, "big")
        except ValueError:
            pass
    pad = math.ceil(len(code) / 8) * 8 - len(code)
    try:
        return base64.b32decode(code + "=" * pad, casefold=True)
    except Exception as e:
        raise ValueError(f"invalid base32: {e}")


def encode_base64(data):
    # type: (bytes) -> str
    return base64.urlsafe_b64encode(data).decode("ascii").rstrip("=")


def decode_base64(code):
    # type: (str) -> bytes
    code = code.replace("+", "-").replace("/", "_").rstrip("=")
    try:
        return base64.urlsafe_b64decode(code + "=" * (-len(code) % 4))
    except Exception as e:
        raise ValueError(f"invalid base64: {e}")


# -- header ----------------------------------------------------------------------------------------
def _encode_varnibble_bits(n):
    # type: (int) -> str
    if 0 <= n < 8:
        return format(n, "04b")
    if 8 <= n < 72:
        return "10" + format(n - 8, "06b")
    if 72 <= n < 584:
        return "110" + format(n - 72, "09b")
    if 584 <= n < 4680:
        return "1110" + format(n - 584, "012b")
    raise ValueError(f"header value {n} out of range")


def encode_header(mtype, stype, version=0, length=1):
    # type: (int, int, int, int) -> bytes
    if 0 <= mtype < 8 and 0 <= stype < 8 and 0 <= version < 8 and 0 <= length < 8:
        return bytes(((mtype << 4) | stype, (version << 4) | length))       # four plain nibbles (as decode_header's fast path)
    bits = "".join(_encode_varnibble_bits(v) for v in (mtype, stype, version, length))
    bits += "0" * (-len(bits) % 8)
    return int(bits, 2).to_bytes(len(bits) // 8, "big")


def decode_header(data):
    # type: (bytes) -> tuple[int, int, int, int, bytes]
    """(MainType, SubType, Version, Length, tail bytes)."""
    if len(data) >= 2 and not (data[0] & 0x88) and not (data[1] & 0x88):
        # every field below 8: four plain nibbles (all unit and ISCC-ID headers in use)
        return data[0] >> 4, data[0] & 15, data[1] >> 4, data[1] & 15, bytes(data[2:])
    return _decode_header_general(data)


def _decode_header_general(data):
    # type: (bytes) -> tuple[int, int, int, int, bytes]
    bits = "".join(format(b, "08b") for b in data)
    pos = 0
    vals = []
    for _ in range(4):
        if pos + 4 > len(bits):
            raise ValueError("truncated ISCC header")
        if bits[pos] == "0":
            width, pre, off = 4, 1, 0
        elif bits[pos : pos + 2] == "10":
            width, pre, off = 8, 2, 8
        elif bits[pos : pos + 3] == "110":
            width, pre, off = 12, 3, 72
        elif bits[pos : pos + 4] == "1110":
            width, pre, off = 16, 4, 584
        else:
            raise ValueError("invalid ISCC header")
        if pos + width > len(bits):
            raise ValueError("truncated ISCC header")
        vals.append(int(bits[pos + pre : pos + width], 2) + off)
        pos += width
    rest = bits[pos:]
    if len(rest) % 8:
        if rest[:4] != "0000":
            raise ValueError("invalid ISCC header padding")
        rest = rest[4:]
    tail = int(rest, 2).to_bytes(len(rest) // 8, "big") if rest else b""
    return vals[0], vals[1], vals[2], vals[3], tail


def encode_length(mtype, bits):
    # type: (int, int) -> int
    if mtype in (MT_META, MT_SEMANTIC, MT_CONTENT, MT_DATA, MT_INSTANCE, MT_FLAKE):
        if bits >= 32 and bits % 32 == 0:
            return bits // 32 - 1
        raise ValueError(f"invalid length {bits} for {MT_NAMES[mtype]}")
    if mtype == MT_ISCC:
        if 0 <= bits <= 7:
            return bits
        raise ValueError(f"invalid unit combination {bits}")
    if mtype == MT_ID:
        if 64 <= bits <= 96:
            return (bits - 64) // 8
        raise ValueError(f"invalid length {bits} for ID")
    raise ValueError(f"invalid MainType {mtype}")


def decode_length(mtype, length, stype=None):
    # type: (int, int, int | None) -> int
    """Body length in bits."""
    if mtype in (MT_META, MT_SEMANTIC, MT_CONTENT, MT_DATA, MT_INSTANCE, MT_FLAKE):
        return (length + 1) * 32
    if mtype == MT_ISCC:
        if stype == ST_ISCC_WIDE:
            return 256
        return len(decode_units(length)) * 64 + 128
    if mtype == MT_ID:
        return length * 8 + 64
    raise ValueError(f"invalid MainType {mtype}")


def decode_units(unit_id):
    # type: (int) -> tuple[int, ...]
    if not 0 <= unit_id < len(_UNITS):
        raise ValueError(f"invalid unit combination {unit_id}")
    return _UNITS[unit_id]


def encode_units(main_types):
    # type: (tuple[int, ...]) -> int
    try:
        return _UNITS.index(tuple(sorted(main_types)))
    except ValueError:
        raise ValueError(f"invalid combination of ISCC-UNITs: {main_types}")


def type_name(mtype, stype, version):
    # type: (int, int, int) -> str
    """``{MAINTYPE}_{SUBTYPE}_V{n}`` (``iscc_search/models.py:112-122``)."""
    try:
        st_names = _SUBTYPE_NAMES[(mtype, version)]
        return f"{MT_NAMES[mtype]}_{st_names[stype]}_V{version}"
    except (KeyError, IndexError):
        raise ValueError(f"unknown ISCC type ({mtype}, {stype}, {version})")


# -- objects ---------------------------------------------------------------------------------------
def clean(iscc):
    # type: (str) -> str
    return iscc.split(":")[-1].replace("-", "").strip()


# ISCC strings this process produced or parsed recently -> their decoded form.  A request normalised from an ISCC-CODE derives its
# unit strings (code_units -> str) and searches each of them a moment later (HipIndex._search_units): four decodes saved per request.
# Iscc objects never change after construction; the memo is bounded and dropped whole when full.
_PARSED = {}
_PARSED_MAX = 8192


def parse(iscc):
    # type: (str) -> Iscc
    """``Iscc(iscc)`` through the memo."""
    obj = _PARSED.get(iscc)
    if obj is None:
        obj = Iscc(iscc)
        if len(_PARSED) >= _PARSED_MAX:
            _PARSED.clear()
        _PARSED[iscc] = obj
    return obj


class Iscc:
    """Decoded ISCC (any kind): digest = header + body."""

    __slots__ = ("digest", "mtype", "stype", "version", "length", "body", "_type")

    def __init__(self, iscc):
        # type: (str | bytes) -> None
        if isinstance(iscc, str):
            self.digest = decode_base32(iscc[5:] if iscc.startswith("ISCC:") else iscc)
        elif isinstance(iscc, (bytes, bytearray)):
            self.digest = bytes(iscc)
        else:
            raise TypeError("`iscc` must be str, bytes")
        if len(self.digest) < 2:
            raise ValueError("ISCC too short")
        self.mtype, self.stype, self.version, self.length, tail = decode_header(self.digest)
        # the reference takes the body as everything after the 2-byte header (models.py:92-99)
        self.body = self.digest[2:]
        self._type = None

    @property
    def iscc_type(self):
        # type: () -> str
        if self._type is None:
            self._type = type_name(self.mtype, self.stype, self.version)
        return self._type

    unit_type = iscc_type

    def __len__(self):
        return len(self.body) * 8

    def __str__(self):
        text = "ISCC:" + encode_base32(self.digest)
        if len(_PARSED) < _PARSED_MAX:
            _PARSED[text] = self          # whoever parses this string next (`parse`) gets the object back
        return text

    def __bytes__(self):
        return self.digest


_ID_HEADERS = (encode_header(MT_ID, 0, 1, 0), encode_header(MT_ID, 1, 1, 0))


def validate_iscc_id(iscc_id, expected_realm=None):
    # type: (str, int | None) -> Iscc
    """Checks of ``iscc_search/indexes/common.py:223-272``; returns the decoded ID."""
    if not iscc_id or not iscc_id.startswith("ISCC:"):
        raise ValueError(f"Invalid ISCC-ID format: '{iscc_id}' (must start with 'ISCC:')")
    try:
        obj = Iscc(iscc_id)
    except ValueError as e:
        raise ValueError(f"Invalid ISCC-ID base32 encoding: {e}")
    if len(obj.digest) != 10:
        raise ValueError(f"Invalid ISCC-ID length: {len(obj.digest)} bytes (expected 10 bytes = 2-byte header + 8-byte body)")
    if obj.mtype != MT_ID:
        raise ValueError(f"Invalid ISCC-ID main type: {obj.mtype} (expected {MT_ID})")
    if obj.length != 0:
        raise ValueError(f"Invalid ISCC-ID length field: {obj.length} (expected 0 for 64-bit ISCC-ID v1). ISCC-ID '{iscc_id}' appears to be malformed.")
    if expected_realm is not None and obj.stype != expected_realm:
        raise ValueError(
            f"Realm mismatch: ISCC-ID '{iscc_id}' has realm={obj.stype}, but expected realm={expected_realm}. "
            f"Cannot query assets from different realm."
        )
    return obj


def iscc_id_to_int(iscc_id):
    # type: (str) -> int
    """Body as big-endian u64: the table key (``models.py:166-176``, ``usearch/index.py:286-289``)."""
    return int.from_bytes(validate_iscc_id(iscc_id).body, "big")


def iscc_id_from_int(key, realm_id):
    # type: (int, int) -> str
    if realm_id not in (0, 1):
        raise ValueError(f"Invalid realm_id {realm_id}, must be 0 or 1")
    return "ISCC:" + encode_base32(_ID_HEADERS[realm_id] + int(key).to_bytes(8, "big"))


def new_iscc_id(realm_id=0):
    # type: (int) -> str
    """Random ISCC-ID: 52-bit microsecond timestamp | 12-bit random server id (``models.py:29-43``)."""
    ident = ((time.time_ns() // 1000) << 12) | random.randint(0, 4095)
    return iscc_id_from_int(ident & (2**64 - 1), realm_id)


def encode_unit(mtype, stype, version, body):
    # type: (int, int, int, bytes) -> str
    return "ISCC:" + encode_base32(encode_header(mtype, stype, version, encode_length(mtype, len(body) * 8)) + body)


def code_units(iscc_code):
    # type: (str | bytes) -> list[Iscc]
    """Decompose an ISCC-CODE into its ISCC-UNITs (behaviour of ``models.py:267-316``)."""
    raw = Iscc(iscc_code).digest
    units = []
    while raw:
        mt, st, vs, ln, body = decode_header(raw)
        if mt != MT_ISCC:
            nbytes = decode_length(mt, ln) // 8
            units.append(Iscc(encode_header(mt, st, vs, ln) + body[:nbytes]))
            raw = body[nbytes:]
            continue
        if st == ST_ISCC_WIDE:
            ln128 = encode_length(MT_DATA, 128)
            units.append(Iscc(encode_header(MT_DATA, 0, vs, ln128) + body[:16]))
            units.append(Iscc(encode_header(MT_INSTANCE, 0, vs, ln128) + body[16:32]))
            break
        ln64 = encode_length(MT_META, 64)
        for idx, mtype in enumerate(decode_units(ln)):
            stype = 0 if mtype == MT_META else st
            units.append(Iscc(encode_header(mtype, stype, vs, ln64) + body[idx * 8 : (idx + 1) * 8]))
        units.append(Iscc(encode_header(MT_DATA, 0, vs, ln64) + body[-16:-8]))
        units.append(Iscc(encode_header(MT_INSTANCE, 0, vs, ln64) + body[-8:]))
        break
    return units


def gen_iscc_code(units, wide=True):
    # type: (list[str], bool) -> str
    """
    Compose an ISCC-CODE from ISCC-UNITs (behaviour of iscc-core's ``gen_iscc_code_v0`` as the reference
    calls it at ``indexes/common.py:309``).  Raises ValueError when the units do not form a valid code.
    """
    cleaned = [clean(u) for u in units]
    if len(cleaned) < 2:
        raise ValueError("Minimum two ISCC units required to generate valid ISCC-CODE")
    for c in cleaned:
        if len(c) < 16:
            raise ValueError(f"Cannot build ISCC-CODE from units shorter than 64-bits: {c}")
    decoded = sorted((decode_header(decode_base32(c)) for c in cleaned), key=lambda t: t[0])
    main_types = tuple(d[0] for d in decoded)
    if main_types[-2:] != (MT_DATA, MT_INSTANCE):
        raise ValueError("ISCC-CODE requires at least MT.DATA and MT.INSTANCE units.")
    is_wide = (
        wide and len(cleaned) == 2 and main_types == (MT_DATA, MT_INSTANCE)
        and all(decode_length(t[0], t[3]) >= 128 for t in decoded)
    )
    if is_wide:
        st = ST_ISCC_WIDE
    else:
        sub_types = [t[1] for t in decoded if t[0] in (MT_SEMANTIC, MT_CONTENT)]
        if len(set(sub_types)) > 1:
            raise ValueError("Semantic-Code and Content-Code must be of same SubType")
        st = sub_types.pop() if sub_types else (ST_ISCC_SUM if len(cleaned) == 2 else ST_ISCC_NONE)
    encoded_length = encode_units(main_types[:-2])
    per_unit = 16 if is_wide else 8
    digest = b"".join(t[4][:per_unit] for t in decoded)
    return "ISCC:" + encode_base32(encode_header(MT_ISCC, st, 0, encoded_length) + digest)
