"""
Randomised differential test: a sequence of add / remove / search operations on HIP tables, mirrored on
the oracle-backed model table (tests/oracle_engine.py), compared bit for bit after every search.
Covers both metrics, both key widths, every code length 1..32 bytes, k from 1 to 4096, batches larger
than one pipeline run (1 024 queries), all queries-per-pass settings, heavy ties and tiny tables; plus the
range-limited search and both document-frequency entry points.
"""

import os

import numpy as np
import pytest

from oracle_engine import OracleTable

N_SEEDS = int(os.environ.get("ISCC_FUZZ_SEEDS", "12"))   # raise for a longer soak

pytestmark = pytest.mark.gpu


def _mask(words, nbytes):
    n, mw = words.shape
    nb = np.broadcast_to(np.asarray(nbytes, dtype=np.int64), (n,))
    out = words.copy()
    for j in range(mw):
        valid = np.clip(nb - 8 * j, 0, 8)
        shift = ((8 * (8 - valid)) % 64).astype(np.uint64)
        m = np.where(valid == 8, ~np.uint64(0), np.where(valid == 0, np.uint64(0), (~np.uint64(0)) << shift))
        out[:, j] &= m
    return out


def _compare(got, exp, tag):
    np.testing.assert_array_equal(got[3], exp[3], err_msg=f"{tag}: counts")
    for q in range(len(exp[3])):
        c = int(exp[3][q])
        np.testing.assert_array_equal(got[1][q, :c], exp[1][q, :c], err_msg=f"{tag}: hamming q={q}")
        np.testing.assert_array_equal(got[2][q, :c], exp[2][q, :c], err_msg=f"{tag}: prefix q={q}")
        np.testing.assert_array_equal(got[0][q, :c], exp[0][q, :c], err_msg=f"{tag}: keys q={q}")


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_operation_sequences(hip_engine, seed):
    rng = np.random.default_rng(1000 + seed)
    metric = int(rng.integers(0, 2))
    key_words = int(rng.integers(1, 3))
    max_bytes = int(rng.choice([1, 3, 8, 12, 16, 24, 32]))
    mw = (max_bytes + 7) // 8
    tq = int(rng.choice([8, 16]))
    hip_engine.set_option("queries_per_pass", tq)
    # odd seeds keep the three-launch path for small segments too (by default they are answered by ONE launch, tiny_search_kernel)
    hip_engine.set_option("tiny_rows", 0 if seed % 2 else 16384)
    t = hip_engine.open_table(metric, key_words, max_bytes)
    model = OracleTable(metric, key_words, max_bytes)
    lengths = [max_bytes] if metric == 0 else sorted({max_bytes, max(1, max_bytes // 2), max(1, max_bytes - 3), 1})
    # a small alphabet of base codes makes near-duplicates and big tie classes common
    bases = rng.integers(0, 2**64, size=(6, mw), dtype=np.uint64)
    next_key = 1
    live = []
    try:
        for step in range(14):
            op = rng.choice(["add", "add", "remove", "search", "search", "within"])
            if op == "add" or not live:
                n = int(rng.choice([1, 7, 300, 5000, 40000]))
                lens = rng.choice(lengths, size=n).astype(np.uint8)
                words = bases[rng.integers(0, len(bases), size=n)].copy()
                flips = rng.integers(0, 4, size=n)
                for f in range(1, 4):                       # flip up to 3 random bits
                    sel = flips >= f
                    words[sel, 0] ^= np.uint64(1) << rng.integers(0, 64, size=int(sel.sum())).astype(np.uint64)
                if rng.random() < 0.5:
                    words = rng.integers(0, 2**64, size=(n, mw), dtype=np.uint64)
                words = _mask(words, lens)
                if key_words == 2:
                    keys = np.stack([rng.integers(0, 3, size=n).astype(np.uint64), np.arange(next_key, next_key + n, dtype=np.uint64)], axis=1)
                else:
                    keys = np.arange(next_key, next_key + n, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(2**61 - 1)
                next_key += n
                nb = lens if metric == 1 else None
                t.add(keys, words, nb)
                model.add(keys, words, nb)
                live.extend(range(len(live), len(live) + n))
            elif op == "remove":
                mk, _, _ = model._arrays()
                if len(mk):
                    pick = rng.choice(len(mk), size=min(len(mk), int(rng.choice([1, 50, 3000]))), replace=False)
                    victims = mk[pick]
                    assert t.remove(victims) == model.remove(victims)
            elif op == "within":
                # range-limited search, document frequency by code and by key (fixed-length tables)
                nq = int(rng.choice([1, 5, 24]))
                k = int(rng.choice([1, 10, 1000, 4096]))
                r = int(rng.choice([0, 0, 1, 3, 10, 8 * max_bytes]))
                qlens = rng.choice(lengths, size=nq).astype(np.uint8) if metric == 1 else None
                q = bases[rng.integers(0, len(bases), size=nq)].copy()
                q[:, 0] ^= rng.integers(0, 4, size=nq).astype(np.uint64) << np.uint64(61)
                q = _mask(q, qlens if metric == 1 else max_bytes)
                tag = f"seed={seed} step={step} tq={tq} nq={nq} k={k} r={r}"
                _compare(t.search_within(q, qlens, k, r), model.search_within(q, qlens, k, r), "within " + tag)
                dup = int(rng.choice([1000, 3]))
                np.testing.assert_array_equal(t.doc_freq(q, qlens, dup), model.doc_freq(q, qlens, dup), err_msg="doc_freq " + tag)
                if metric == 0:
                    mk, _, _ = model._arrays()
                    some = mk[rng.choice(len(mk), size=min(len(mk), 12), replace=False)]
                    np.testing.assert_array_equal(t.get_freq(some, dup), model.get_freq(some, dup), err_msg="get_freq " + tag)
            else:
                nq = int(rng.choice([1, 5, 37, 1100]))
                k = int(rng.choice([1, 10, 100, 1000, 4096]))
                if nq * k > 400_000:
                    k = 10
                qlens = rng.choice(lengths, size=nq).astype(np.uint8) if metric == 1 else None
                q = bases[rng.integers(0, len(bases), size=nq)].copy()
                q[:, 0] ^= rng.integers(0, 16, size=nq).astype(np.uint64)
                q = _mask(q, qlens if metric == 1 else max_bytes)
                assert t.size == model.size
                tag = f"seed={seed} step={step} tq={tq} nq={nq} k={k}"
                if rng.random() < 0.4:
                    # the same search as one of several requests of a search_many call (deferred or ordinary path)
                    k2, r2 = int(rng.choice([1, 10, 300])), int(rng.choice([0, 2]))
                    many = hip_engine.search_many([(t, q[:7], None if qlens is None else qlens[:7], k2, r2), (t, q, qlens, k, None)])
                    _compare(many[1], model.search(q, qlens, k), "many " + tag)
                    _compare(many[0], model.search_within(q[:7], None if qlens is None else qlens[:7], k2, r2), "many/within " + tag)
                else:
                    _compare(t.search(q, qlens, k), model.search(q, qlens, k), tag)
        assert t.size == model.size
    finally:
        t.drop()
        hip_engine.set_option("queries_per_pass", 8)
        hip_engine.set_option("tiny_rows", 16384)
