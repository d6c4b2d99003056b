"""
``isccsearch_simprint_score`` (search + asset scoring on the device, ``csrc/simprint_score.hip``) against the plain-loop
checker of ``usearch_core.py:171-269`` (``tests/simprint_checker.py``): float64 scores must compare ``==``, the order
(-score, asset) and the cut to ``limit`` must agree, and so must every matched chunk (query, stored bytes, score, offset,
size, document frequency).  The checker is fed by the ORACLE's neighbour lists where the table is small enough, and by the
device's own lists (whose parity has its own tests) at 10 M rows.
"""

import json
import os
import struct

import numpy as np
import pytest

from helpers import flip_bits
from iscc_search_amd.simprint import DOC_FREQ_DUP_LIMIT, HipSimprintIndex, pack_chunk_pointer
from oracle_engine import OracleEngine
from simprint_checker import score_lists

pytestmark = pytest.mark.gpu


def _corpus(rng, ndim, assets, chunks_per_asset, pool_size, max_flip, asset_ids=None):
    nbytes = ndim // 8
    pool = [rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes() for _ in range(pool_size)]
    keys, vecs = [], []
    for a in range(assets):
        body = (asset_ids[a] if asset_ids is not None else a + 1).to_bytes(8, "big")
        for c in range(chunks_per_asset):
            v = flip_bits(pool[int(rng.integers(0, pool_size))], int(rng.integers(0, max_flip + 1)))
            keys.append(pack_chunk_pointer(body, c * 10, 10 + c))
            vecs.append(np.frombuffer(v, dtype=np.uint8))
    return pool, keys, vecs


def _check(index, oracle_index, simprints, limit, threshold, total_assets, device_doc_freq, detailed=True):
    """Device path against the checker fed by the oracle-backed index's lists / stored vectors / frequencies."""
    ndim = index.ndim
    got = index.search_raw(simprints, limit=limit, threshold=threshold, detailed=detailed, total_assets=total_assets, device_doc_freq=device_doc_freq)
    count = max(1, limit * index.oversampling_factor)
    queries = np.stack([np.frombuffer(s, dtype=np.uint8) for s in simprints])
    batch = oracle_index._index.search(queries, count=count)
    lists = [[(bytes(k), int(h)) for k, h in zip(batch[q].keys, batch[q].hamming)] for q in range(len(simprints))]
    freq = (lambda s: int(oracle_index._index.doc_freq(np.frombuffer(s, dtype=np.uint8).reshape(1, -1), DOC_FREQ_DUP_LIMIT)[0])) if device_doc_freq else None
    stored = lambda key: (lambda v: None if v is None else v.tobytes())(oracle_index._index.get(key))
    want = score_lists(simprints, lists, ndim, limit, threshold, stored, freq, total_assets)
    assert [r.iscc_id_body for r in got] == [w[0] for w in want]
    for r, (asset, score, matches, chunks) in zip(got, want):
        assert r.score == score, (asset.hex(), r.score, score)
        assert r.matches == matches and r.queried == len(simprints)
        if detailed:
            assert [(c.query, c.match, c.score, c.offset, c.size, c.freq) for c in r.chunks] == \
                   [(simprints[qi], match, sim, offset, size, f) for qi, match, sim, offset, size, f in chunks]
        else:
            assert r.chunks is None
    return got


@pytest.fixture(scope="module")
def engine():
    from iscc_search_amd.engine import HipEngine

    e = HipEngine(0)
    yield e
    e.close()


@pytest.mark.parametrize("ndim", [64, 128, 256])
@pytest.mark.parametrize("device_doc_freq", [True, False])
def test_scores_chunks_and_order_equal_the_reference_loop(engine, ndim, device_doc_freq):
    rng = np.random.default_rng(1000 + ndim)
    pool, keys, vecs = _corpus(rng, ndim, assets=400, chunks_per_asset=6, pool_size=50, max_flip=ndim // 16)
    index, oracle = HipSimprintIndex(engine, ndim=ndim), HipSimprintIndex(OracleEngine(), ndim=ndim)
    index.add_raw(keys, vecs)
    oracle.add_raw(keys, vecs)
    simprints = [flip_bits(pool[i % len(pool)], i % 4) for i in range(37)] + [rng.integers(0, 256, size=ndim // 8, dtype=np.uint8).tobytes() for _ in range(3)]
    for limit, threshold in ((10, 0.8), (3, 0.9), (100, 0.0), (7, 1.0), (5, 1.01)):
        got = _check(index, oracle, simprints, limit, threshold, 400, device_doc_freq)
        assert (len(got) > 0) == (threshold <= 1.0)
    _check(index, oracle, simprints[:1], 10, 0.8, 400, device_doc_freq)
    _check(index, oracle, simprints, 10, 0.8, 0, device_doc_freq, detailed=False)      # empty-index IDF: every score 0.0, order by asset
    index.close()


def test_the_data_of_the_cpu_tier_bit_test(engine):
    """``tests/test_simprint.py::test_asset_scores_have_the_bits_of_the_reference_loop``'s corpus, frequencies from the device."""
    rng = np.random.default_rng(2718)
    pool, keys, vecs = _corpus(rng, 64, assets=300, chunks_per_asset=6, pool_size=40, max_flip=3)
    index, oracle = HipSimprintIndex(engine, ndim=64), HipSimprintIndex(OracleEngine(), ndim=64)
    index.add_raw(keys, vecs)
    oracle.add_raw(keys, vecs)
    simprints = [flip_bits(pool[i], i % 3) for i in range(23)]
    got = _check(index, oracle, simprints, 400, 0.8, 300, True)
    assert len(got) > 50
    index.close()


def test_reference_literals_through_the_device_path(engine):
    """``kat_simprint.json`` (approx.py:308-330): a 48-bit-off chunk scores 0.25 -- dropped at 0.9, kept at 0.0."""
    with open(os.path.join(os.path.dirname(__file__), "golden", "kat_simprint.json")) as f:
        t = json.load(f)["threshold"]
    idx = HipSimprintIndex(engine, ndim=t["ndim"])
    query = b"\xaa" * 8
    idx.add_raw([pack_chunk_pointer(b"\x04" * 8, 0, 100)], [np.frombuffer(flip_bits(query, t["stored_flip_bits"]), dtype=np.uint8)])
    for device_doc_freq in (False, True):
        assert idx.search_raw([query], limit=10, threshold=0.9, total_assets=1, device_doc_freq=device_doc_freq) == []
        res = idx.search_raw([query], limit=10, threshold=0.0, total_assets=1, detailed=True, device_doc_freq=device_doc_freq)
        assert len(res) == 1 and res[0].chunks[0].score == t["score"] and res[0].score == t["score"]
        assert res[0].chunks[0].match == flip_bits(query, t["stored_flip_bits"]) and (res[0].chunks[0].offset, res[0].chunks[0].size) == (0, 100)
    idx.close()


def test_more_query_simprints_than_one_batch(engine):
    """1 500 query simprints = two batches of the search pipeline; the entry list is appended across them."""
    rng = np.random.default_rng(77)
    pool, keys, vecs = _corpus(rng, 64, assets=200, chunks_per_asset=5, pool_size=30, max_flip=2)
    index, oracle = HipSimprintIndex(engine, ndim=64, oversampling_factor=2), HipSimprintIndex(OracleEngine(), ndim=64, oversampling_factor=2)
    index.add_raw(keys, vecs)
    oracle.add_raw(keys, vecs)
    simprints = [flip_bits(pool[i % len(pool)], i % 3) for i in range(1500)]
    _check(index, oracle, simprints, 8, 0.85, 200, True)
    index.close()


def test_a_query_with_more_equal_rows_than_neighbours_asked_for(engine):
    """
    count = limit x oversampling = 40 < dup_limit and 300 stored chunks EQUAL a query simprint: its own document frequency
    cannot be read off its 40-row list -- the library asks the collision scan (the ``unknown`` path).
    """
    rng = np.random.default_rng(5)
    hot = rng.integers(0, 256, size=8, dtype=np.uint8).tobytes()
    other = rng.integers(0, 256, size=8, dtype=np.uint8).tobytes()
    keys, vecs = [], []
    for a in range(150):                                  # 150 assets x 2 chunks equal to `hot`
        for c in range(2):
            keys.append(pack_chunk_pointer((a + 1).to_bytes(8, "big"), c, 1))
            vecs.append(np.frombuffer(hot, dtype=np.uint8))
    for a in range(150, 170):
        keys.append(pack_chunk_pointer((a + 1).to_bytes(8, "big"), 0, 1))
        vecs.append(np.frombuffer(flip_bits(other, a % 3), dtype=np.uint8))
    index, oracle = HipSimprintIndex(engine, ndim=64, oversampling_factor=20), HipSimprintIndex(OracleEngine(), ndim=64, oversampling_factor=20)
    index.add_raw(keys, vecs)
    oracle.add_raw(keys, vecs)
    got = _check(index, oracle, [hot, other], 2, 0.9, 170, True)
    assert len(got) == 2
    index.close()


def test_an_asset_id_of_all_ones_and_zero(engine):
    """The marking kernel's hash keeps all-ones as its EMPTY value: that asset id takes a side path."""
    rng = np.random.default_rng(9)
    ids = [0xFFFFFFFFFFFFFFFF, 0, 1, 0xFFFFFFFFFFFFFFFE] + list(range(10, 40))
    pool, keys, vecs = _corpus(rng, 64, assets=len(ids), chunks_per_asset=4, pool_size=6, max_flip=2, asset_ids=ids)
    index, oracle = HipSimprintIndex(engine, ndim=64), HipSimprintIndex(OracleEngine(), ndim=64)
    index.add_raw(keys, vecs)
    oracle.add_raw(keys, vecs)
    got = _check(index, oracle, [flip_bits(p, 1) for p in pool], 40, 0.8, len(ids), True)
    assert {r.iscc_id_body for r in got} >= {b"\xff" * 8, b"\x00" * 8}
    index.close()


def test_large_limit_becomes_a_radius_on_the_device_path_too(engine):
    idx = HipSimprintIndex(engine, ndim=64, oversampling_factor=20)
    oracle = HipSimprintIndex(OracleEngine(), ndim=64, oversampling_factor=20)
    base = bytes([0xAA] * 8)
    keys = [pack_chunk_pointer((1000 + i).to_bytes(8, "big"), 0, 10) for i in range(40)]
    vecs = [np.frombuffer(flip_bits(base, i % 5), dtype=np.uint8) for i in range(40)]
    idx.add_raw(keys, vecs)
    oracle.add_raw(keys, vecs)
    big = idx.search_raw([base], limit=1000, threshold=0.9, total_assets=40, detailed=True, device_doc_freq=True)
    want = oracle.search_raw([base], limit=1000, threshold=0.9, total_assets=40, detailed=True, device_doc_freq=True)
    assert len(big) == 40 and [(r.iscc_id_body, r.score, r.matches) for r in big] == [(r.iscc_id_body, r.score, r.matches) for r in want]
    idx.close()


def test_ten_million_chunks(engine):
    """
    10 M random 128-bit chunks + planted near-duplicates of 64 query simprints; the device's scores against the checker fed
    with the device's own neighbour lists, stored vectors and frequencies (each of which has its own parity test).
    """
    rng = np.random.default_rng(31337)
    ndim, n = 128, 10_000_000
    index = HipSimprintIndex(engine, ndim=ndim)
    table = index._index._table
    table.add_synthetic(ndim // 8, n, 0x5151, 0, 0)          # keys (0, row): asset 0 holds every random chunk
    simprints = [rng.integers(0, 256, size=16, dtype=np.uint8).tobytes() for _ in range(64)]
    keys, vecs = [], []
    for a in range(300):                                      # 300 assets, each near some of the query simprints
        body = (a + 1).to_bytes(8, "big")
        for c, qi in enumerate(rng.choice(64, size=int(rng.integers(1, 12)), replace=False)):
            keys.append(pack_chunk_pointer(body, c, 7))
            vecs.append(np.frombuffer(flip_bits(simprints[int(qi)], int(rng.integers(0, 20))), dtype=np.uint8))
    index.add_raw(keys, vecs)
    limit, threshold, total = 20, 0.75, 301
    got = index.search_raw(simprints, limit=limit, threshold=threshold, detailed=True, total_assets=total, device_doc_freq=True)
    from iscc_search_amd.nphd import words_to_key128

    queries = np.stack([np.frombuffer(s, dtype=np.uint8) for s in simprints])
    kw, ham, cnt = index._index.search_arrays(queries, count=limit * index.oversampling_factor)
    lists = [[(k, int(h)) for k, h in zip(words_to_key128(kw[q, : int(cnt[q])]), ham[q, : int(cnt[q])])] for q in range(len(simprints))]
    stored = lambda key: index._index.get(key).tobytes()
    freq = lambda s: int(index._index.doc_freq(np.frombuffer(s, dtype=np.uint8).reshape(1, -1), DOC_FREQ_DUP_LIMIT)[0])
    want = score_lists(simprints, lists, ndim, limit, threshold, stored, freq, total)
    assert len(want) == limit
    assert [(r.iscc_id_body, r.score, r.matches) for r in got] == [(w[0], w[1], w[2]) for w in want]
    for r, w in zip(got, want):
        assert [(c.query, c.match, c.score, c.offset, c.size, c.freq) for c in r.chunks] == [(simprints[qi], m, s, o, z, f) for qi, m, s, o, z, f in w[3]]
    index.close()


def test_scores_follow_additions_and_removals(engine):
    """
    Differential run over a changing table: chunks are added and removed between requests (the frequency column is dropped by
    every change and rebuilt by the next request); after every change the device's answer must equal the checker's over the
    oracle-backed twin.
    """
    rng = np.random.default_rng(4242)
    ndim = 128
    pool = [rng.integers(0, 256, size=ndim // 8, dtype=np.uint8).tobytes() for _ in range(25)]
    index, oracle = HipSimprintIndex(engine, ndim=ndim), HipSimprintIndex(OracleEngine(), ndim=ndim)
    live = {}                                   # chunk pointer -> simprint bytes
    next_asset = [1]

    def add(n_assets):
        keys, vecs = [], []
        for _ in range(n_assets):
            body = next_asset[0].to_bytes(8, "big")
            next_asset[0] += 1
            for c in range(int(rng.integers(1, 7))):
                v = flip_bits(pool[int(rng.integers(0, len(pool)))], int(rng.integers(0, 9)))
                k = pack_chunk_pointer(body, c * 16, 16)
                keys.append(k)
                vecs.append(np.frombuffer(v, dtype=np.uint8))
                live[k] = v
        index.add_raw(keys, vecs)
        oracle.add_raw(keys, vecs)

    def remove(n):
        victims = [list(live)[int(i)] for i in rng.choice(len(live), size=min(n, len(live)), replace=False)]
        for k in victims:
            del live[k]
        index.remove(victims)
        oracle.remove(victims)

    add(120)
    for round_ in range(12):
        if round_ % 3 == 2:
            remove(int(rng.integers(5, 60)))
        else:
            add(int(rng.integers(5, 40)))
        simprints = [flip_bits(pool[int(rng.integers(0, len(pool)))], int(rng.integers(0, 5))) for _ in range(int(rng.integers(1, 30)))]
        assets = len({k[:8] for k in live})
        _check(index, oracle, simprints, int(rng.integers(1, 15)), float(rng.choice([0.0, 0.8, 0.9, 0.95])), assets, device_doc_freq=bool(round_ % 2))
        assert index.size == len(live)
    index.close()


def test_concurrent_requests_and_writers(engine):
    """Scoring calls from several threads beside a writer: every call is serialised by the handle and answers a consistent table."""
    import threading

    rng = np.random.default_rng(99)
    pool, keys, vecs = _corpus(rng, 64, assets=200, chunks_per_asset=4, pool_size=20, max_flip=3)
    index = HipSimprintIndex(engine, ndim=64)
    index.add_raw(keys, vecs)
    simprints = [flip_bits(p, 1) for p in pool]
    want = index.search_raw(simprints, limit=10, threshold=0.8, detailed=True, total_assets=200, device_doc_freq=True)
    errors, stop = [], threading.Event()

    def reader():
        try:
            while not stop.is_set():
                got = index.search_raw(simprints, limit=10, threshold=0.8, detailed=True, total_assets=200, device_doc_freq=True)
                if [(r.iscc_id_body, r.score) for r in got] != [(r.iscc_id_body, r.score) for r in want]:
                    errors.append("a reader saw another answer")
                    return
        except BaseException as e:      # noqa: BLE001
            errors.append(repr(e))

    def writer():
        # far from every query (all-ones / all-zero codes of assets beyond the corpus): the answer must not change
        try:
            for i in range(30):
                k = [pack_chunk_pointer((10_000 + i).to_bytes(8, "big"), j, 1) for j in range(3)]
                index.add_raw(k, [np.frombuffer(bytes([0xFF * (j & 1)] * 8), dtype=np.uint8) for j in range(3)])
                index.remove(k[:2])
        except BaseException as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=reader) for _ in range(4)] + [threading.Thread(target=writer)]
    for t in threads:
        t.start()
    threads[-1].join()
    stop.set()
    for t in threads[:-1]:
        t.join()
    assert errors == []
    index.close()


@pytest.mark.parametrize("ndim", [64, 128, 256])
def test_hard_boundary_search_scored_on_the_device_equals_the_host_scoring(engine, ndim):
    """
    ``isccsearch_simprint_exact`` against ``HipSimprintIndex._search_exact_host`` (which the CPU tier holds to the reference's
    literals and to a model of the LMDB dupsort walk): scores ``==``, order, matches and chunks -- repeated query simprints, a
    simprint of another length, thresholds that drop assets, limits that cut, dup_limit below / above the collision counts.
    """
    rng = np.random.default_rng(7000 + ndim)
    nb = ndim // 8
    pool = [rng.integers(0, 256, size=nb, dtype=np.uint8).tobytes() for _ in range(30)]
    keys, vecs, seen = [], [], set()
    ids = [0xFFFFFFFFFFFFFFFF, 0] + list(range(1, 120))
    while len(keys) < 2500:
        body = ids[int(rng.integers(0, len(ids)))].to_bytes(8, "big")
        off, size = int(rng.integers(0, 60)), int(rng.integers(1, 4))
        if (body, off, size) in seen:
            continue
        seen.add((body, off, size))
        # a skewed pool: a few simprints collide hundreds of times, most a few times
        sp = pool[min(int(rng.exponential(4.0)), len(pool) - 1)]
        keys.append(pack_chunk_pointer(body, off, size))
        vecs.append(np.frombuffer(sp, dtype=np.uint8))
    index = HipSimprintIndex(engine, ndim=ndim)
    index.add_raw(keys, vecs)
    absent = rng.integers(0, 256, size=nb, dtype=np.uint8).tobytes()
    cases = [
        [pool[0]],
        [pool[0], pool[1], pool[0], pool[5], absent, pool[29], b"\x01\x02\x03"],
        [pool[i % 30] for i in range(70)],
        [absent],
    ]
    for query in cases:
        distinct = [sp for sp in dict.fromkeys(query) if len(sp) == nb]
        for limit, threshold, dup_limit in ((10, 0.0, 1000), (3, 0.0, 25), (50, 0.2, 70), (1000, 0.05, 1000), (5, 0.9, 1)):
            if not distinct:
                assert index.search_exact(query, limit=limit, threshold=threshold, detailed=True, dup_limit=dup_limit) == []
                continue
            # (search_exact itself sends requests of fewer than EXACT_DEVICE_FROM simprints to the host scoring: the device path is called directly)
            got = index._search_exact_device(query, distinct, limit, threshold, True, dup_limit)
            want = index._search_exact_host(query, distinct, limit, threshold, True, dup_limit)
            assert [(r.iscc_id_body, r.score, r.matches, r.queried) for r in got] == [(r.iscc_id_body, r.score, r.matches, r.queried) for r in want]
            for g, w in zip(got, want):
                assert [(c.query, c.match, c.score, c.offset, c.size, c.freq) for c in g.chunks] == [(c.query, c.match, c.score, c.offset, c.size, c.freq) for c in w.chunks]
            plain = index._search_exact_device(query, distinct, limit, threshold, False, dup_limit)
            assert [(r.iscc_id_body, r.score, r.matches) for r in plain] == [(r.iscc_id_body, r.score, r.matches) for r in want] and all(r.chunks is None for r in plain)
    assert index.search_exact([pool[0]] * 3 + [pool[2]], limit=5, detailed=True)[0].queried == 4
    many = [pool[i % 30] for i in range(200)]                    # past EXACT_DEVICE_FROM: the public entry takes the device path itself
    assert [(r.iscc_id_body, r.score, r.matches) for r in index.search_exact(many, limit=20, detailed=True)] == \
           [(r.iscc_id_body, r.score, r.matches) for r in index._search_exact_host(many, list(dict.fromkeys(many)), 20, 0.0, True, 1000)]
    index.close()


def test_hard_boundary_search_with_more_lookups_than_one_batch(engine):
    rng = np.random.default_rng(31)
    pool = [rng.integers(0, 256, size=16, dtype=np.uint8).tobytes() for _ in range(1300)]
    keys = [pack_chunk_pointer((1 + i % 211).to_bytes(8, "big"), i, 1) for i in range(4000)]
    vecs = [np.frombuffer(pool[i % len(pool)], dtype=np.uint8) for i in range(4000)]
    index = HipSimprintIndex(engine, ndim=128)
    index.add_raw(keys, vecs)
    query = pool[:1200] + pool[:100]
    got = index.search_exact(query, limit=30, threshold=0.0, detailed=True)
    want = index._search_exact_host(query, list(dict.fromkeys(query)), 30, 0.0, True, 1000)
    assert [(r.iscc_id_body, r.score, r.matches) for r in got] == [(r.iscc_id_body, r.score, r.matches) for r in want] and len(got) == 30
    for g, w in zip(got, want):
        assert [(c.query, c.offset, c.size, c.freq) for c in g.chunks] == [(c.query, c.offset, c.size, c.freq) for c in w.chunks]
    index.close()


def test_the_largest_request_the_library_takes(engine):
    """
    8 192 query simprints (``ISCCSEARCH_MAX_SCORED_SIMPRINTS``) x limit 204 = 4 080 neighbours each (33 M records, none of them
    leaving the device) and a limit whose oversampled count exceeds ``MAX_K`` (radius lists): the device scoring against the host
    scoring of the same neighbour lists -- same assets, ``==`` scores, same chunks (``tools/probe_score_extremes.py`` at 1 M rows).
    """
    rng = np.random.default_rng(5)
    rows = 200_000
    pool = [rng.integers(0, 256, size=16, dtype=np.uint8).tobytes() for _ in range(4000)]
    keys, vecs = [], []
    for i in range(rows):
        a, c = divmod(i, 20)
        keys.append(pack_chunk_pointer((a + 1).to_bytes(8, "big"), c * 7, 7))
        v = flip_bits(pool[int(rng.integers(0, len(pool)))], int(rng.integers(0, 6))) if i % 3 == 0 else rng.integers(0, 256, size=16, dtype=np.uint8).tobytes()
        vecs.append(np.frombuffer(v, dtype=np.uint8))
    index = HipSimprintIndex(engine, ndim=128)
    index.add_raw(keys, vecs)
    for nq, limit in ((8192, 204), (100, 300)):
        simprints = [flip_bits(pool[i % len(pool)], i % 5) for i in range(nq)]
        for device_doc_freq in (True, False):
            got = index.search_raw(simprints, limit=limit, threshold=0.9, detailed=True, total_assets=rows // 20, device_doc_freq=device_doc_freq)
            want = index._search_raw_host(simprints, limit, 0.9, True, None, rows // 20, device_doc_freq)
            assert len(got) == len(want) == limit
            for x, y in zip(got, want):
                assert (x.iscc_id_body, x.score, x.matches, x.queried) == (y.iscc_id_body, y.score, y.matches, y.queried)
                assert [(c.query, c.match, c.score, c.offset, c.size, c.freq) for c in x.chunks] == [(c.query, c.match, c.score, c.offset, c.size, c.freq) for c in y.chunks]
    # one more simprint than the library scores in one call: refused there (search_raw itself takes the host path for such requests)
    with pytest.raises((ValueError, RuntimeError, OSError)):
        index._search_raw_device([pool[0]] * 8193, 1, 0.9, False, 10, 1000)
    index.close()
