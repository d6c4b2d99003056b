"""
Simprint (chunk-level) path: pure helpers against the reference's literals, and the scoring pipeline
of ``usearch_core.py:137-269`` fed by exact neighbours.  CPU tier uses the oracle-backed engine; the
gpu-marked variant runs the same assertions through the HIP engine.
"""

import json
import math
import os

import numpy as np
import pytest

from helpers import flip_bits, make_asset, make_iscc_id, sp
from iscc_search_amd import codec
from iscc_search_amd.index import HipIndexManager
from iscc_search_amd.schema import IsccIndex, IsccQuery
from iscc_search_amd.simprint import HipSimprintIndex, calculate_idf, pack_chunk_pointer, unpack_chunk_pointer
from oracle_engine import OracleEngine

with open(os.path.join(os.path.dirname(__file__), "golden", "kat_simprint.json")) as f:
    KAT = json.load(f)


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request):
    if request.param == "oracle":
        yield OracleEngine()
    else:
        from iscc_search_amd.engine import HipEngine

        e = HipEngine(0)
        yield e
        e.close()


def test_chunk_pointer_kats():
    for c in KAT["chunk_pointer"]["cases"]:
        packed = pack_chunk_pointer(bytes.fromhex(c["body"]), c["offset"], c["size"])
        assert packed.hex() == c["packed"] and len(packed) == 16
        assert unpack_chunk_pointer(packed) == (bytes.fromhex(c["body"]), c["offset"], c["size"])
    with pytest.raises(ValueError, match="ISCC-ID body must be 8 bytes"):
        pack_chunk_pointer(b"\x01" * 7, 0, 0)
    with pytest.raises(ValueError, match="exceeds max"):
        pack_chunk_pointer(b"\x01" * 8, 2**32, 0)
    with pytest.raises(ValueError, match="exceeds max"):
        pack_chunk_pointer(b"\x01" * 8, 0, 2**32)
    with pytest.raises(ValueError, match="Expected 16 bytes"):
        unpack_chunk_pointer(b"\x00" * 15)


def test_idf_kats():
    for c in KAT["idf"]["cases"]:
        expected = c["value"] if "value" in c else eval(c["expr"], {"log": math.log})
        assert calculate_idf(c["freq"], c["total"]) == expected


def test_threshold_filters_weak_matches(engine):
    """approx.py:308-330."""
    t = KAT["threshold"]
    idx = HipSimprintIndex(engine, ndim=t["ndim"])
    query = b"\xaa" * 8
    idx.add_raw([pack_chunk_pointer(b"\x04" * 8, 0, 100)], [np.frombuffer(flip_bits(query, t["stored_flip_bits"]), dtype=np.uint8)])
    assert idx.search_raw([query], limit=10, threshold=0.9, total_assets=1) == []
    res = idx.search_raw([query], limit=10, threshold=0.0, total_assets=1, detailed=True)
    assert len(res) == 1 and res[0].chunks[0].score == t["score"] and res[0].score == t["score"]
    idx.close()


def test_best_chunk_per_query_per_asset_and_partial_coverage(engine):
    """approx.py:985-1034, :392-414: best chunk wins; an unmatched query simprint lowers the score."""
    idx = HipSimprintIndex(engine, ndim=64)
    a, b = b"\x01" * 8, b"\x02" * 8
    s1, s2 = bytes(range(8)), bytes(range(100, 108))
    idx.add_raw(
        [pack_chunk_pointer(a, 0, 10), pack_chunk_pointer(a, 10, 10), pack_chunk_pointer(b, 0, 10)],
        [np.frombuffer(s1, np.uint8), np.frombuffer(flip_bits(s1, 4), np.uint8), np.frombuffer(s2, np.uint8)],
    )
    full = idx.search_raw([s1], limit=10, threshold=0.75, detailed=True, total_assets=2)
    assert [r.iscc_id_body for r in full] == [a]
    assert full[0].score == 1.0 and full[0].matches == 1 and full[0].queried == 1
    assert (full[0].chunks[0].offset, full[0].chunks[0].size, full[0].chunks[0].match) == (0, 10, s1)
    unknown = bytes([0xF0] * 8)
    part = idx.search_raw([s1, unknown], limit=10, threshold=0.75, detailed=True, total_assets=2)
    assert part[0].iscc_id_body == a and part[0].matches == 1 and part[0].queried == 2
    assert 0.0 < part[0].score < full[0].score
    idx.close()


def test_idf_weighting_and_doc_freq_called_with_stored_bytes(engine):
    """approx.py:333-390 (rare beats common) and :1222-1254 (doc_freq_fn sees the STORED simprint)."""
    idx = HipSimprintIndex(engine, ndim=64)
    stored = b"\xaa" * 8
    query = flip_bits(stored, 4)
    idx.add_raw([pack_chunk_pointer(b"\x09" * 8, 7, 70)], [np.frombuffer(stored, np.uint8)])
    seen = []

    def freq(sp_bytes):
        seen.append(sp_bytes)
        return 3

    res = idx.search_raw([query], limit=5, threshold=0.5, detailed=True, doc_freq_fn=freq, total_assets=10)
    assert stored in seen and query not in seen
    c = res[0].chunks[0]
    assert (c.query, c.match, c.freq, c.score) == (query, stored, 3, 1.0 - 4 / 64)
    # two assets, same similarity on different simprints: the asset matching the RARE simprint ranks first
    idx2 = HipSimprintIndex(engine, ndim=64)
    rare, common = b"\x11" * 8, b"\x22" * 8
    idx2.add_raw([pack_chunk_pointer(b"\x01" * 8, 0, 1), pack_chunk_pointer(b"\x02" * 8, 0, 1)],
                 [np.frombuffer(rare, np.uint8), np.frombuffer(common, np.uint8)])
    ranked = idx2.search_raw([rare, common], limit=5, threshold=0.9, doc_freq_fn=lambda s: 1 if s == rare else 500, total_assets=1000)
    assert [r.iscc_id_body for r in ranked] == [b"\x01" * 8, b"\x02" * 8]
    assert ranked[0].score > ranked[1].score
    idx.close()
    idx2.close()


def test_batch_dedup_keeps_first_and_remove(engine):
    """usearch_core.py:85-108, :110-135."""
    idx = HipSimprintIndex(engine, ndim=128)
    k = pack_chunk_pointer(b"\x05" * 8, 1, 2)
    v1, v2 = np.full(16, 0xAA, np.uint8), np.full(16, 0x55, np.uint8)
    idx.add_raw([k, k], [v1, v2])
    assert idx.size == 1 and k in idx
    assert idx.search_raw([v1.tobytes()], limit=1, total_assets=1, detailed=True)[0].chunks[0].match == v1.tobytes()
    idx.remove([k])
    assert idx.size == 0 and k not in idx
    assert idx.search_raw([v1.tobytes()], limit=1, total_assets=1) == []
    idx.add_raw([], [])
    idx.close()


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def manager(request):
    m = HipIndexManager("hip:///", engine=OracleEngine()) if request.param == "oracle" else HipIndexManager("hip:///")
    yield m
    m.close()


def test_chunk_matches_through_the_protocol(manager):
    """End to end: add assets with simprints, query by simprints only (usearch/index.py:1357-1469)."""
    rng = np.random.default_rng(1)
    manager.create_index(IsccIndex(name="t"))
    s = [rng.integers(0, 256, size=16, dtype=np.uint8).tobytes() for _ in range(4)]
    a = make_asset(rng, 0, metadata={"source": "https://example.com/a"},
                   simprints={"CONTENT_TEXT_V0": [sp(s[0], 0, 100), sp(s[1], 100, 120)]})
    b = make_asset(rng, 1, simprints={"CONTENT_TEXT_V0": [sp(flip_bits(s[0], 6), 5, 50)], "SEMANTIC_TEXT_V0": [sp(s[2], 0, 9)]})
    manager.add_assets("t", [a, b])
    q = IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(s[0]), codec.encode_base64(s[1])]})
    res = manager.search_assets("t", q, limit=10)
    assert res.global_matches == []
    assert [m.iscc_id for m in res.chunk_matches] == [a.iscc_id, b.iscc_id]
    top = res.chunk_matches[0]
    assert top.score == 1.0 and top.source == "https://example.com/a"
    t = top.types["CONTENT_TEXT_V0"]
    assert (t.matches, t.queried, t.score) == (2, 2, 1.0)
    assert sorted((c.offset, c.size, c.score) for c in t.chunks) == [(0, 100, 1.0), (100, 120, 1.0)]
    second = res.chunk_matches[1].types["CONTENT_TEXT_V0"]
    assert (second.matches, second.queried) == (1, 2) and second.chunks[0].score == 1.0 - 6 / 128
    # unknown simprint type is ignored gracefully; iscc_id self-exclusion applies to chunk matches too
    res2 = manager.search_assets("t", IsccQuery(simprints={"NOPE_V0": [codec.encode_base64(s[0])]}))
    assert res2.chunk_matches == []


def test_simprint_update_replaces_old_chunks(manager):
    """approx.py:1174-1219."""
    r = KAT["replace"]
    old, new = bytes.fromhex(r["old"]), bytes.fromhex(r["new"])
    rng = np.random.default_rng(2)
    manager.create_index(IsccIndex(name="t"))
    base = make_asset(rng, 0)
    manager.add_assets("t", [base.model_copy(update={"simprints": {"CONTENT_TEXT_V0": [sp(old, 0, 10)]}})])
    manager.add_assets("t", [base.model_copy(update={"simprints": {"CONTENT_TEXT_V0": [sp(new, 0, 10)]}})])
    hit = manager.search_assets("t", IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(new)]}))
    assert hit.chunk_matches[0].score == 1.0
    miss = manager.search_assets("t", IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(old)]}))
    assert all(m.score < 1.0 for m in miss.chunk_matches)


def test_large_limit_lists_everything_within_the_threshold_or_refuses(engine):
    """
    ``count = limit * oversampling`` beyond the engine's 4 096-neighbour cap: the threshold turns the
    request into a radius, so the answer is still complete; a radius that would hold more than the cap is refused.
    """
    idx = HipSimprintIndex(engine, ndim=64, oversampling_factor=20)
    base = bytes([0xAA] * 8)
    keys, vecs = [], []
    for i in range(40):
        keys.append(pack_chunk_pointer((1000 + i).to_bytes(8, "big"), 0, 10))
        vecs.append(np.frombuffer(flip_bits(base, i % 5), dtype=np.uint8))
    idx.add_raw(keys, vecs)
    small = idx.search_raw([base], limit=10, threshold=0.9, total_assets=40, detailed=True)
    big = idx.search_raw([base], limit=1000, threshold=0.9, total_assets=40, detailed=True)          # 20 000 neighbours asked for
    assert len(big) == 40 and [r.iscc_id_body for r in big[:10]] == [r.iscc_id_body for r in small]
    assert all(r.chunks[0].score >= 0.9 for r in big) and min(r.chunks[0].score for r in big) == 1.0 - 4 / 64
    assert idx.search_raw([base], limit=1000, threshold=0.97, total_assets=40) != [] and len(idx.search_raw([base], limit=1000, threshold=0.97)) == 16
    with pytest.raises(ValueError, match="exceeds the"):
        many_k, many_v = [], []
        for i in range(4200):
            many_k.append(pack_chunk_pointer((5000 + i).to_bytes(8, "big"), 1, 1))
            many_v.append(np.frombuffer(base, dtype=np.uint8))
        idx.add_raw(many_k, many_v)
        idx.search_raw([base], limit=1000, threshold=0.9)
    idx.close()


def _score_by_the_reference_loop(index, simprints, limit, threshold, freqs, total_assets):
    """
    The scoring of ``usearch_core.py:171-269`` written as the reference writes it: per asset, Python float additions over
    the matched query simprints (dict order) and then over EVERY unmatched query simprint in ascending order.
    """
    queries = np.stack([np.frombuffer(s, dtype=np.uint8) for s in simprints])
    key_words, ham, cnt = index._index.search_arrays(queries, count=limit * index.oversampling_factor)
    best = {}
    from iscc_search_amd.nphd import words_to_key128

    for qi in range(len(simprints)):
        for pos in range(int(cnt[qi])):
            score = 1.0 - float(ham[qi, pos]) / index.ndim
            if score < threshold:
                continue
            raw = words_to_key128(key_words[qi, pos : pos + 1])[0]
            cur = best.setdefault(raw[:8], {}).get(qi)
            if cur is None or score > cur[0]:
                best[raw[:8]][qi] = (score, raw)
    out = {}
    for asset, per_q in best.items():
        total = weighted = 0.0
        for qi, (score, raw) in per_q.items():
            idf = calculate_idf(freqs[bytes(index._index.get(raw))], total_assets)
            total += idf
            weighted += idf * score
        for qi in range(len(simprints)):
            if qi not in per_q:
                total += calculate_idf(freqs[simprints[qi]], total_assets)
        out[asset] = weighted / total if total > 0 else 0.0
    return out


def test_asset_scores_have_the_bits_of_the_reference_loop():
    """
    The per-asset IDF total is accumulated with np.cumsum instead of a Python loop over every unmatched query simprint
    (O(assets x queries) in the reference): same additions in the same order, so the scores must be EQUAL, not close.
    """
    rng = np.random.default_rng(2718)
    index = HipSimprintIndex(OracleEngine(), ndim=64)
    pool = [rng.integers(0, 256, size=8, dtype=np.uint8).tobytes() for _ in range(40)]
    keys, vecs, freqs = [], [], {}
    for asset in range(300):
        body = (asset + 1).to_bytes(8, "big")
        for c in range(6):
            base = pool[int(rng.integers(0, len(pool)))]
            v = flip_bits(base, int(rng.integers(0, 4)))
            keys.append(pack_chunk_pointer(body, c * 10, 10))
            vecs.append(np.frombuffer(v, dtype=np.uint8))
    index.add_raw(keys, vecs)
    simprints = [flip_bits(pool[i], i % 3) for i in range(23)]
    freqs = {v.tobytes(): 1 + (hash(v.tobytes()) % 17) for v in vecs}
    freqs.update({s: 1 + (hash(s) % 17) for s in simprints})
    got = index.search_raw(simprints, limit=400, threshold=0.8, doc_freq_fn=lambda s: freqs[bytes(s)], total_assets=300)
    want = _score_by_the_reference_loop(index, simprints, 400, 0.8, freqs, 300)
    assert len(got) == len(want) > 50
    for r in got:
        assert r.score == want[r.iscc_id_body], (r.iscc_id_body.hex(), r.score, want[r.iscc_id_body])
