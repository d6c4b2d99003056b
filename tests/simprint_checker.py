"""
Checker for the simprint scoring path (TEST INFRASTRUCTURE -- never imported by the product).

A plain-loop restatement of what ``UsearchSimprintIndex.search_raw`` does with its neighbour lists
(``iscc_search/indexes/simprint/usearch_core.py:171-269``), fed with exact neighbours (ascending (distance, key), as the
engine and the oracle define the order).  Every float operation happens in the reference's order, so a scoring
implementation -- the device kernels of ``csrc/simprint_score.hip``, the host path of ``simprint.py`` -- must return
float64 scores that compare ``==``.
"""

import math
import struct


def idf(freq, total_assets):
    """``lmdb_ops.calculate_idf`` (``lmdb_ops.py:67-81``)."""
    if total_assets <= 0:
        return 0.0
    return math.log(1 + total_assets / (1 + freq))


def score_lists(simprints, lists, ndim, limit, threshold, stored_vector, doc_freq, total_assets):
    """
    :param simprints: query simprints (bytes)
    :param lists: per query simprint, [(16-byte key, integer distance)] ascending (distance, key)
    :param stored_vector: key -> stored simprint bytes (``self._index.get``, ``usearch_core.py:221``)
    :param doc_freq: simprint bytes -> document frequency (the reference's ``doc_freq_fn``; None = every frequency is 1)
    :return: [(asset body, score, matches, [(query index, stored bytes, chunk score, offset, size, freq)])] best first, <= limit
    """
    # usearch_core.py:171-196 -- best chunk per (asset, query simprint), first one seen wins unless a later one scores higher
    asset_best = {}
    for qi, neighbours in enumerate(lists):
        for key, distance in neighbours:
            score = 1.0 - (float(distance) / ndim)
            if score < threshold:
                continue
            asset = key[:8]
            offset, size = struct.unpack("!II", key[8:16])
            per_query = asset_best.setdefault(asset, {})
            if qi not in per_query or score > per_query[qi][2]:
                per_query[qi] = (offset, size, score, key)
    cache = {}

    def freq_of(sp):
        if sp not in cache:
            cache[sp] = doc_freq(sp) if doc_freq is not None else 1
        return cache[sp]

    # :213-265 -- matched query simprints first (IDF of the STORED bytes), then every unmatched one in ascending order
    scored = []
    for asset, per_query in asset_best.items():
        total_idf = 0.0
        weighted = 0.0
        chunks = []
        for qi, (offset, size, sim, key) in per_query.items():
            stored = stored_vector(key)
            match = stored if stored is not None else simprints[qi]
            f = freq_of(match)
            w = idf(f, total_assets)
            total_idf += w
            weighted += w * sim
            chunks.append((qi, match, sim, offset, size, f))
        for qi in range(len(simprints)):
            if qi not in per_query:
                total_idf += idf(freq_of(simprints[qi]), total_assets)
        scored.append((asset, weighted / total_idf if total_idf > 0 else 0.0, len(per_query), chunks))
    # :267-269
    scored.sort(key=lambda r: (-r[1], r[0]))
    return scored[:limit]
