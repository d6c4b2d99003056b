"""
numpy restatement of exact NPHD / Hamming k-NN (test infrastructure only).

Independent of ``nphd_oracle.c``: works on unpacked bit arrays straight from the code *bytes*
(no 64-bit word packing, no popcount), compares distances as exact ``fractions.Fraction`` and
sorts the complete candidate list.  O(n * nq) memory -- small cases only.

Follows: ``docs/explanation/similarity-search.md:24-29`` (NPHD = hamming over the common prefix
divided by the prefix length), ``tests/test_usearch_search.py:122-167`` (fixed-length tables return
the raw bit count, ascending).
"""

from fractions import Fraction

import numpy as np


def pack_codes(codes, max_words):
    # type: (list[bytes], int) -> tuple[np.ndarray, np.ndarray]
    """
    Pack byte strings big-endian into zero-padded uint64 words (the C-ABI layout).

    :return: (words uint64 [n, max_words], nbytes uint8 [n])
    """
    n = len(codes)
    buf = np.zeros((n, max_words * 8), dtype=np.uint8)
    nb = np.zeros(n, dtype=np.uint8)
    for i, c in enumerate(codes):
        if not 1 <= len(c) <= max_words * 8:
            raise ValueError(f"code length {len(c)} out of range")
        buf[i, : len(c)] = np.frombuffer(c, dtype=np.uint8)
        nb[i] = len(c)
    words = buf.reshape(n, max_words, 8).view(">u8").reshape(n, max_words).astype(np.uint64)
    return words, nb


def ref_distance_pairs(codes, query, nphd):
    # type: (list[bytes], bytes, bool) -> list[tuple[int, int]]
    """(hamming, prefix_bits) of every stored code against one query, by bit arrays."""
    qbits = np.unpackbits(np.frombuffer(query, dtype=np.uint8))
    out = []
    for c in codes:
        cbits = np.unpackbits(np.frombuffer(c, dtype=np.uint8))
        if nphd:
            p = min(len(cbits), len(qbits))
        else:
            if len(cbits) != len(qbits):
                raise ValueError("fixed-length table: query length differs from code length")
            p = len(cbits)
        out.append((int(np.count_nonzero(cbits[:p] != qbits[:p])), p))
    return out


def ref_topk(codes, keys, query, k, nphd):
    # type: (list[bytes], list[int], bytes, int, bool) -> list[tuple[int, int, int]]
    """
    Exact top-k of one query: list of (key, hamming, prefix_bits) ascending by (distance, key).

    ``keys`` are Python ints (64- or 128-bit).  Distance is h/p for NPHD, h for Hamming.
    """
    pairs = ref_distance_pairs(codes, query, nphd)
    rows = []
    for key, (h, p) in zip(keys, pairs):
        dist = Fraction(h, p) if nphd else Fraction(h)
        rows.append((dist, key, h, p))
    rows.sort(key=lambda r: (r[0], r[1]))
    return [(key, h, p) for _, key, h, p in rows[:k]]


def ref_within(codes, keys, query, k, nphd, max_hamming):
    # type: (list[bytes], list[int], bytes, int, bool, int) -> list[tuple[int, int, int]]
    """
    Range-limited top-k: rows whose Hamming distance over the compared prefix is <= ``max_hamming``,
    ascending by (distance, key), at most ``k``.  ``max_hamming = 0`` restates the collision lookup of
    ``lmdb_ops.search_simprints_exact`` (``lmdb_ops.py:197-203``: every duplicate of the key in LMDB value
    order = ascending chunk-pointer bytes, at most ``dup_limit``).
    """
    pairs = ref_distance_pairs(codes, query, nphd)
    rows = []
    for key, (h, p) in zip(keys, pairs):
        if h <= max_hamming:
            rows.append((Fraction(h, p) if nphd else Fraction(h), key, h, p))
    rows.sort(key=lambda r: (r[0], r[1]))
    return [(key, h, p) for _, key, h, p in rows[:k]]


def ref_doc_freq(codes, keys, query, dup_limit=1000):
    # type: (list[bytes], list[int], bytes, int) -> int
    """
    Distinct assets among the first ``dup_limit`` rows equal to ``query`` in ascending key order
    (``count_doc_freq``, ``lmdb_ops.py:139-166``).  ``keys`` are 128-bit ints: asset = key >> 64.
    """
    equal = sorted(key for key, c in zip(keys, codes) if c == query)[:dup_limit]
    return len({key >> 64 for key in equal})


def np_within(words, nbytes, keys, q_words, q_nbytes, k, max_hamming):
    # type: (np.ndarray, np.ndarray | int, np.ndarray, np.ndarray, int, int, int) -> tuple
    """
    Vectorised range-limited top-k of ONE query over packed words (medium-size parity cases).

    ``words`` uint64 [n, W] big-endian packed, ``nbytes`` per-row lengths (or one int), ``keys`` uint64 [n] or
    [n, 2].  Returns (keys, hamming, prefix_bits) arrays of the <= k hits ordered by (h/p, key).
    """
    n, W = words.shape
    nb = np.full(n, nbytes, dtype=np.int64) if np.isscalar(nbytes) else nbytes.astype(np.int64)
    pbytes = np.minimum(nb, int(q_nbytes))
    ham = np.zeros(n, dtype=np.int64)
    for w in range(W):
        # bytes of word w that lie inside the compared prefix: big-endian packing keeps them at the top
        inside = np.clip(pbytes - 8 * w, 0, 8)
        mask = np.where(inside == 8, np.uint64(0xFFFFFFFFFFFFFFFF),
                        (~(np.uint64(0xFFFFFFFFFFFFFFFF) >> (inside.astype(np.uint64) * np.uint64(8)))) * (inside > 0).astype(np.uint64))
        ham += np.bitwise_count((words[:, w] ^ q_words[w]) & mask).astype(np.int64)
    sel = np.nonzero(ham <= max_hamming)[0]
    pbits = pbytes[sel] * 8
    h = ham[sel]
    kk = keys[sel]
    # exact order of h/p without floats: compare h * (L / p) with L = lcm of the prefix lengths present
    L = int(np.lcm.reduce(np.unique(pbits))) if len(sel) else 1
    num = h * (L // np.maximum(pbits, 1))
    if kk.ndim == 2:
        order = np.lexsort((kk[:, 1], kk[:, 0], num))
    else:
        order = np.lexsort((kk, num))
    order = order[:k]
    return kk[order], h[order], pbits[order]
