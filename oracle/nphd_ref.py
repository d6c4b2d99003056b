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
