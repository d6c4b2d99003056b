"""Deterministic ISCC sample data for the host-logic tests (built with this repo's own codec)."""

import numpy as np

from iscc_search_amd import codec
from iscc_search_amd.schema import IsccEntry, IsccSimprint


def make_iscc_id(i, realm=0):
    # type: (int, int) -> str
    """ISCC-ID from a (timestamp, hub id) pair like the reference's fixtures (tests/conftest.py:70-79)."""
    return codec.iscc_id_from_int(((1_000_000 + i) << 12) | (i & 0xFFF), realm)


def rnd_unit(rng, mtype, stype=0, bits=64):
    return codec.encode_unit(mtype, stype, 0, rng.integers(0, 256, size=bits // 8, dtype=np.uint8).tobytes())


def flip_bits(data, n):
    # type: (bytes, int) -> bytes
    """Flip the first n bits (controlled Hamming distance)."""
    ba = bytearray(data)
    for b in range(n):
        ba[b // 8] ^= 1 << (7 - b % 8)
    return bytes(ba)


def make_units(rng, bits=64, with_meta=True, with_content=True):
    units = []
    if with_meta:
        units.append(rnd_unit(rng, codec.MT_META, 0, bits))
    if with_content:
        units.append(rnd_unit(rng, codec.MT_CONTENT, 0, bits))
    units.append(rnd_unit(rng, codec.MT_DATA, 0, bits))
    units.append(rnd_unit(rng, codec.MT_INSTANCE, 0, bits))
    return units


def make_asset(rng, i, bits=64, metadata=None, simprints=None, **kw):
    units = make_units(rng, bits, **kw)
    return IsccEntry(
        iscc_id=make_iscc_id(i), iscc_code=codec.gen_iscc_code(units), units=units, metadata=metadata, simprints=simprints
    )


def sp(raw, offset=0, size=100):
    return IsccSimprint(simprint=codec.encode_base64(raw), offset=offset, size=size)
