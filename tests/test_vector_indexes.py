"""
Reference-shaped vector index wrappers (``iscc_search_amd/nphd.py``) against the behaviours the
reference pins for the objects they replace (tests/test_usearch_{add,get,contains,remove,search}.py,
literal vectors [178,204,60,240] etc. reused as data).  CPU tier: oracle-backed engine; gpu tier: HIP.
"""

import numpy as np
import pytest

from iscc_search_amd.nphd import HipHammingIndex, HipIndex128, HipNphdIndex
from oracle_engine import OracleEngine

A = np.array([178, 204, 60, 240], dtype=np.uint8)
B = np.array([100, 150, 200, 250], dtype=np.uint8)
C = np.array([1, 2, 3, 4], dtype=np.uint8)


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request):
    if request.param == "oracle":
        yield OracleEngine()
    else:
        from iscc_search_amd.engine import HipEngine

        e = HipEngine(0)
        yield e
        e.close()


def test_get_contains_remove_roundtrip(engine):
    idx = HipHammingIndex(engine, 32)
    assert idx.get(1) is None and 1 not in idx and len(idx) == 0          # empty index (test_usearch_get.py:15-30)
    idx.add(1, A)
    got = idx.get(1)
    assert isinstance(got, np.ndarray) and got.ndim == 1 and got.tolist() == A.tolist()
    assert idx.get(2) is None
    assert idx.get([1, 999]) [0].tolist() == A.tolist() and idx.get([1, 999])[1] is None
    assert idx.contains(1) is True and idx.contains(2) is False and (1 in idx)
    assert idx.contains([1, 2, 1]).tolist() == [True, False, True]
    assert idx.contains(np.array([], dtype=np.uint64)).tolist() == []
    assert idx.remove(1) == 1 and idx.remove(1) == 0 and idx.remove([]) == 0
    assert not idx.contains(1) and len(idx) == 0                            # test_usearch_contains.py:53-65
    idx.close()


def test_duplicate_add_keeps_original(engine):
    """tests/test_usearch_add.py:53-62."""
    idx = HipHammingIndex(engine, 32)
    idx.add(1, A)
    idx.add(1, B)
    assert len(idx) == 1 and idx.get(1).tolist() == A.tolist()
    idx.add([2, 2, 3], np.stack([B, C, C]))                               # inside a batch the first occurrence wins
    assert len(idx) == 3 and idx.get(2).tolist() == B.tolist()
    idx.close()


def test_remove_then_readd_update_pattern(engine):
    """tests/test_usearch_remove.py:226-275: the pattern add_assets uses for updates."""
    idx = HipHammingIndex(engine, 32)
    idx.add([1, 2, 3], np.stack([A, B, C]))
    new = np.array([[255] * 4, [128] * 4, [64] * 4], dtype=np.uint8)
    assert idx.remove([1, 2, 3, 77]) == 3
    idx.add([1, 2, 3], new)
    assert len(idx) == 3 and [idx.get(k).tolist() for k in (1, 2, 3)] == new.tolist()
    m = idx.search(np.array([255, 255, 255, 254], dtype=np.uint8), count=1)
    assert m.keys.tolist() == [1] and m.distances.tolist() == [1.0]
    idx.close()


def test_large_and_zero_keys(engine):
    """tests/test_usearch_contains.py:214-235."""
    idx = HipHammingIndex(engine, 32)
    big = 2**63 - 1
    idx.add([0, big, 2**64 - 1], np.stack([A, B, C]))
    assert idx.contains(0) and idx.contains(big) and idx.contains(2**64 - 1)
    m = idx.search(B, count=3)
    assert m.keys.tolist() == [big, 0, 2**64 - 1] and m.keys.dtype == np.uint64
    idx.close()


def test_search_shapes_single_and_batch(engine):
    """tests/test_usearch_search.py: single -> Matches, 2-D -> BatchMatches, count clipped by size."""
    idx = HipHammingIndex(engine, 32)
    idx.add([1, 2], np.stack([A, B]))
    single = idx.search(A, count=100)
    assert len(single) == 2 and single.keys.shape == (2,) and single.distances.dtype == np.float32
    assert single.to_list() == [(1, 0.0), (2, 16.0)]
    batch = idx.search(np.stack([A, B]), count=2)
    assert len(batch) == 2 and batch[0].keys.tolist() == [1, 2] and batch[1].keys.tolist() == [2, 1]
    assert batch[1].distances.tolist() == [0.0, 16.0]
    with pytest.raises(ValueError, match="`count` must be >= 1"):
        idx.search(A, count=0)
    with pytest.raises(ValueError):
        idx.search(np.zeros(5, dtype=np.uint8))
    idx.close()


def test_nphd_index_variable_lengths(engine):
    idx = HipNphdIndex(engine, max_dim=256)
    v64, v128, v256 = bytes(range(8)), bytes(range(16)), bytes(range(32))
    idx.add([10, 11, 12], [np.frombuffer(v, dtype=np.uint8) for v in (v64, v128, v256)])
    assert idx.size == 3 and 11 in idx and 13 not in idx
    assert idx.get(12).tobytes() == v256 and idx.get(10).tobytes() == v64 and idx.get(99) is None
    m = idx.search(np.frombuffer(v128, dtype=np.uint8), count=10)
    assert m.keys.tolist() == [10, 11, 12] and m.distances.tolist() == [0.0, 0.0, 0.0]   # every prefix agrees
    assert m.prefix_bits.tolist() == [64, 128, 128] and m.distances.dtype == np.float32
    far = bytearray(v128)
    far[0] ^= 0xFF
    m2 = idx.search(bytes(far), count=3)
    assert m2.hamming.tolist() == [8, 8, 8] and [float(d) for d in m2.distances] == [float(np.float32(8) / np.float32(128))] * 2 + [float(np.float32(8) / np.float32(64))]
    assert m2.keys.tolist() == [11, 12, 10]                                          # 8/128 before 8/64
    assert idx.remove([10, 555]) == 1 and idx.size == 2
    idx.add(11, np.frombuffer(v64, dtype=np.uint8))                                   # duplicate: original kept
    assert idx.get(11).tobytes() == v128
    with pytest.raises(ValueError):
        HipNphdIndex(engine, max_dim=257)
    idx.close()


def test_index128_composite_keys(engine):
    idx = HipIndex128(engine, ndim=64)
    k1, k2 = b"\x01" * 8 + b"\x00\x00\x00\x05\x00\x00\x00\x09", b"\x01" * 8 + b"\x00\x00\x00\x06\x00\x00\x00\x09"
    idx.add([k1, k2], np.stack([np.arange(8, dtype=np.uint8), np.arange(8, dtype=np.uint8)]))
    assert len(idx) == 2 and k1 in idx and (b"\x02" * 16) not in idx
    m = idx.search(np.arange(8, dtype=np.uint8).reshape(1, 8), count=5)[0]
    assert [bytes(k) for k in m.keys] == [k1, k2] and m.distances.tolist() == [0.0, 0.0]   # tie broken by key bytes
    assert idx.get(k2).tolist() == list(range(8)) and idx.get(b"\x09" * 16) is None
    assert [None if v is None else v.tolist() for v in idx.get_many([k1, b"\x09" * 16])] == [list(range(8)), None]
    assert idx.remove([k1]) == 1 and len(idx) == 1
    with pytest.raises(ValueError):
        idx.add([b"short"], np.zeros((1, 8), dtype=np.uint8))
    idx.close()
