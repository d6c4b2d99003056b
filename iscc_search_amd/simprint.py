"""
Chunk-level (simprint) search over the HIP engine.

``HipSimprintIndex`` has the interface and scoring of the reference's ``UsearchSimprintIndex``
(``iscc_search/indexes/simprint/usearch_core.py:37-313``): batched fixed-length Hamming k-NN with
oversampling, best-chunk-per-query-per-asset, IDF-weighted asset score.  Where the reference asks an
approximate HNSW (``ShardedIndex128.search``, ``:165``) this asks the exact GPU scan, so the scoring
pipeline is fed the true nearest neighbours.  Pure helpers restate ``simprint/lmdb_ops.py:30-81``.
"""

import math
import struct
from collections import defaultdict
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np

from iscc_search_amd._lib import MAX_K, MAX_SCORED_SIMPRINTS
from iscc_search_amd.nphd import HipIndex128, words_to_key128

CHUNK_POINTER_BYTES = 16
EXACT_DEVICE_FROM = 128     # query simprints from which a hard-boundary search is scored on the device (isccsearch_simprint_exact)
EXACT_FIRST_K = 64          # records per lookup a hard-boundary search asks for first (a full list is asked again up to dup_limit)
DOC_FREQ_DUP_LIMIT = 1000   # duplicates looked at per simprint when counting its assets: the reference's safety cap
                            # (count_doc_freq(dup_limit=1000), lmdb_ops.py:139-166); HipIndex128.get_freq / doc_freq share it
MAX_OFFSET = 2**32 - 1
MAX_SIZE = 2**32 - 1


def pack_chunk_pointer(iscc_id_body, offset, size):
    # type: (bytes, int, int) -> bytes
    """16-byte composite key: iscc_id_body(8) | offset(4, BE) | size(4, BE)  (``lmdb_ops.py:30-49``)."""
    if len(iscc_id_body) != 8:
        raise ValueError(f"ISCC-ID body must be 8 bytes, got {len(iscc_id_body)}")
    if offset > MAX_OFFSET:
        raise ValueError(f"Offset {offset} exceeds max {MAX_OFFSET}")
    if size > MAX_SIZE:
        raise ValueError(f"Size {size} exceeds max {MAX_SIZE}")
    return iscc_id_body + struct.pack("!II", offset, size)


def unpack_chunk_pointer(data):
    # type: (bytes) -> tuple[bytes, int, int]
    if len(data) != CHUNK_POINTER_BYTES:
        raise ValueError(f"Expected {CHUNK_POINTER_BYTES} bytes, got {len(data)}")
    offset, size = struct.unpack("!II", data[8:16])
    return data[:8], offset, size


def _pack_simprints(simprints):
    # type: (list[bytes]) -> np.ndarray
    """[n, nbytes] uint8 of equal-length simprints in one copy (np.stack of 512 frombuffer views cost 0.8 ms); ragged input: numpy's own error."""
    nbytes = len(simprints[0])
    if set(map(len, simprints)) == {nbytes}:
        return np.frombuffer(b"".join(simprints), dtype=np.uint8).reshape(len(simprints), nbytes)
    return np.stack([np.frombuffer(sp, dtype=np.uint8) for sp in simprints])


def calculate_idf(freq, total_assets):
    # type: (int, int) -> float
    """Smooth IDF ``log(1 + total / (1 + freq))``; 0.0 when the index is empty (``lmdb_ops.py:67-81``)."""
    if total_assets <= 0:
        return 0.0
    return math.log(1 + total_assets / (1 + freq))


def coverage_quality_score(matched_simprints, doc_freq, queried):
    # type: (list[bytes], dict[bytes, int], int) -> float
    """
    Coverage x quality of one asset's collisions (``lmdb_ops._calculate_coverage_quality_score``, :252-301).

    coverage = distinct query simprints matched / simprints queried; quality = mean over those simprints of
    the min-max normalised inverse document frequency (1.0 when there is one simprint or all frequencies agree).
    """
    if not matched_simprints:
        return 0.0
    freqs = [doc_freq.get(sp, 1) for sp in dict.fromkeys(matched_simprints)]
    coverage = len(freqs) / queried
    lo, hi = min(freqs), max(freqs)
    if len(freqs) == 1 or lo == hi:
        return coverage * 1.0
    min_inv, max_inv = 1.0 / hi, 1.0 / lo
    quality = sum((1.0 / f - min_inv) / (max_inv - min_inv) for f in freqs) / len(freqs)
    return coverage * quality


@dataclass(slots=True)
class MatchedChunkRaw:
    query: bytes
    match: bytes
    score: float
    offset: int
    size: int
    freq: int


@dataclass(slots=True)
class SimprintMatchRaw:
    iscc_id_body: bytes
    score: float
    queried: int
    matches: int
    chunks: Optional[List[MatchedChunkRaw]]


class HipSimprintIndex:
    """Derived simprint index of one simprint type (fixed ``ndim``), 128-bit composite keys."""

    def __init__(self, engine, ndim=128, oversampling_factor=20):
        # type: (object, int, int) -> None
        self.ndim = ndim
        self.oversampling_factor = oversampling_factor
        self._index = HipIndex128(engine, ndim)

    def add_raw(self, composite_keys, vectors):
        # type: (list[bytes], list[np.ndarray]) -> None
        """Append vectors; duplicates inside the batch keep the first occurrence (``usearch_core.py:85-108``)."""
        if not composite_keys:
            return
        seen = set()
        keep = []
        for i, k in enumerate(composite_keys):
            k = bytes(k)
            if k not in seen:
                seen.add(k)
                keep.append(i)
        keys = [bytes(composite_keys[i]) for i in keep]
        arr = np.stack([np.asarray(vectors[i], dtype=np.uint8) for i in keep])
        self._index.add(keys, arr, trusted_unique=True)

    def remove(self, composite_keys):
        # type: (list[bytes]) -> None
        if composite_keys:
            self._index.remove([bytes(k) for k in composite_keys])

    def __contains__(self, composite_key):
        return bytes(composite_key) in self._index

    @property
    def size(self):
        return len(self._index)

    def search_raw(self, simprints, limit=10, threshold=0.0, detailed=False, doc_freq_fn=None, total_assets=0, device_doc_freq=False):
        # type: (list[bytes], int, float, bool, Callable[[bytes], int] | None, int, bool) -> list[SimprintMatchRaw]
        """
        Oversampled batched search + IDF-weighted asset scoring (``usearch_core.py:137-269``).

        ``doc_freq_fn`` is the reference's per-simprint callback (``None``: every frequency is 1).  With
        ``device_doc_freq`` the frequencies come from the device instead -- the stored simprints' from the
        table's frequency column (one gather for all matched chunks), the unmatched query simprints' from
        one collision scan -- which is what the reference's callback computes with an LMDB cursor walk per
        simprint (``usearch/index.py:1395-1403``, ``lmdb_ops.py:139-166``).
        """
        if not simprints or len(self._index) == 0:
            return []
        if (device_doc_freq or doc_freq_fn is None) and self._index.scores_on_device and len(simprints) <= MAX_SCORED_SIMPRINTS:
            return self._search_raw_device(simprints, limit, threshold, detailed, total_assets, DOC_FREQ_DUP_LIMIT if device_doc_freq else 0)
        return self._search_raw_host(simprints, limit, threshold, detailed, doc_freq_fn, total_assets, device_doc_freq)

    def _radius_for(self, threshold):
        # type: (float) -> int
        """Largest distance whose score ``1 - d / ndim`` still passes ``threshold`` (0 when none does)."""
        radius = self.ndim
        while radius > 0 and 1.0 - radius / self.ndim < threshold:
            radius -= 1
        return radius

    def _search_raw_device(self, simprints, limit, threshold, detailed, total_assets, dup_limit):
        # type: (list[bytes], int, float, bool, int, int) -> list[SimprintMatchRaw]
        """
        The whole of ``usearch_core.py:137-269`` in ONE library call (``isccsearch_simprint_score``): the neighbour lists stay in
        device memory, the kernels of ``csrc/simprint_score.hip`` apply the threshold, keep the best chunk per (asset, query),
        sum the IDF weights in the reference's order of float64 additions, sort by (-score, asset) and cut to ``limit``;
        only those assets (and their matched chunks) come back.  ``dup_limit`` 0: every document frequency is 1
        (``doc_freq_fn=None``); otherwise the device's own frequencies (``device_doc_freq``).
        """
        queries = _pack_simprints(simprints)
        count = max(1, limit * self.oversampling_factor)
        radius = None
        if count > MAX_K:
            # (see _search_raw_host: beyond the cap the request becomes "every row within the match threshold")
            radius, count = self._radius_for(threshold), MAX_K
        results, chunks, words, info = self._index.score_assets(queries, count, radius, threshold, limit, total_assets, dup_limit, detailed)
        if radius is not None and info[2] >= MAX_K:
            raise ValueError(
                f"limit {limit} x oversampling {self.oversampling_factor} = {limit * self.oversampling_factor} neighbours per simprint exceeds the "
                f"{MAX_K} this backend returns, and a query simprint has that many stored chunks within the match threshold"
            )
        # (columns as lists and positional construction: 520 chunk objects -- 13 assets x 40 matched simprints -- cost 0.67 ms built
        #  field by field from structured rows, 0.28 ms this way)
        body = [a.to_bytes(8, "big") for a in results["asset"].tolist()]
        if not detailed:
            return [SimprintMatchRaw(b, s, len(simprints), m, None) for b, s, m in zip(body, results["score"].tolist(), results["matches"].tolist())]
        nbytes = self.ndim // 8
        stride = (nbytes + 7) // 8 * 8
        raw = np.ascontiguousarray(words).astype(">u8").tobytes()
        sim = [1.0 - h / self.ndim for h in range(self.ndim + 1)]
        key_lo = chunks["key_lo"]
        cols = zip(chunks["query"].tolist(), chunks["hamming"].tolist(), (key_lo >> np.uint64(32)).tolist(),
                   (key_lo & np.uint64(0xFFFFFFFF)).tolist(), chunks["freq"].tolist())
        detail = [MatchedChunkRaw(simprints[q], raw[j * stride : j * stride + nbytes], sim[h], o, z, f) for j, (q, h, o, z, f) in enumerate(cols)]
        return [SimprintMatchRaw(b, s, len(simprints), m, detail[f : f + m])
                for b, s, m, f in zip(body, results["score"].tolist(), results["matches"].tolist(), results["first_chunk"].tolist())]

    def _search_raw_host(self, simprints, limit, threshold, detailed, doc_freq_fn, total_assets, device_doc_freq):
        # type: (list[bytes], int, float, bool, Callable[[bytes], int] | None, int, bool) -> list[SimprintMatchRaw]
        """
        The same pipeline with the scoring on the host: what a Python ``doc_freq_fn`` callback needs (it cannot run on the
        device) and what a table sharded over several GPUs uses (the stored simprints and their frequencies live on the
        owning ranks).  The neighbour search is the device's either way.
        """
        queries = _pack_simprints(simprints)
        count = max(1, limit * self.oversampling_factor)
        if count <= MAX_K:
            key_words, ham, cnt = self._index.search_arrays(queries, count=count)
        else:
            # The reference asks usearch for `count` neighbours unbounded (usearch_core.py:164); the engine returns at most
            # MAX_K per query.  Only neighbours scoring >= threshold survive the filter below, i.e. rows within a fixed
            # radius: list exactly those.  A list that still fills the cap cannot be represented -- refuse, never truncate.
            radius = self._radius_for(threshold)
            key_words, ham, cnt = self._index.search_arrays(queries, count=MAX_K, max_hamming=radius)
            if int(cnt.max(initial=0)) >= MAX_K and count > MAX_K:
                raise ValueError(
                    f"limit {limit} x oversampling {self.oversampling_factor} = {count} neighbours per simprint exceeds the "
                    f"{MAX_K} this backend returns, and a query simprint has that many stored chunks within the match threshold"
                )

        # Threshold first, on the whole [queries x count] block at once: the reference walks every neighbour in
        # Python (usearch_core.py:175-196); most of an oversampled list fails the threshold.  Same arithmetic
        # (float64 division of an exact integer), same visiting order (query, then rank) for the survivors.
        # (the score is a decreasing function of the integer distance: the largest distance whose score -- in this very
        #  arithmetic -- still passes is found once, the block is then filtered on integers and only survivors get a float)
        h_max = -1
        for h in range(self.ndim + 1):
            if 1.0 - np.float64(h) / self.ndim >= threshold:
                h_max = h
            else:
                break
        keep = ham <= h_max if h_max >= 0 else np.zeros(ham.shape, dtype=bool)
        if int(cnt.min(initial=ham.shape[1])) < ham.shape[1]:
            keep &= np.arange(ham.shape[1])[None, :] < cnt[:, None]
        q_of, pos_of = np.nonzero(keep)
        raw_keys = words_to_key128(key_words[q_of, pos_of])
        kept_scores = (1.0 - ham[q_of, pos_of].astype(np.float64) / self.ndim).tolist()

        # best chunk per (asset, query simprint)
        asset_best = defaultdict(dict)
        for qi, raw_key, score in zip(q_of.tolist(), raw_keys, kept_scores):
            asset_id = raw_key[:8]
            cur = asset_best[asset_id].get(qi)
            if cur is None or score > cur[2]:
                offset, size = struct.unpack("!II", raw_key[8:16])
                asset_best[asset_id][qi] = (offset, size, score, raw_key)
        if not asset_best:
            return []

        # stored vectors of every winning chunk in ONE device round trip (the reference does one
        # `get` per chunk, usearch_core.py:221)
        all_keys = [v[3] for best in asset_best.values() for v in best.values()]
        fetched = dict(zip(all_keys, self._index.get_many(all_keys)))

        freq_cache = {}
        if device_doc_freq:
            for ckey, f in zip(all_keys, self._index.get_freq(all_keys)):
                vec = fetched.get(ckey)
                if vec is not None:
                    freq_cache[vec.tobytes()] = int(f)
            missing = [sp for sp in dict.fromkeys(simprints) if sp not in freq_cache]
            if missing and any(len(best) < len(simprints) for best in asset_best.values()):
                freq_cache.update(zip(missing, self.doc_freq(missing)))

        def get_freq(sp):
            if sp not in freq_cache:
                if device_doc_freq:
                    freq_cache[sp] = self.doc_freq([sp])[0]
                else:
                    freq_cache[sp] = doc_freq_fn(sp) if doc_freq_fn is not None else 1
            return freq_cache[sp]

        # Scoring, in the reference's order of float additions (usearch_core.py:201-236): per asset first the matched query
        # simprints (IDF of the STORED bytes), then every unmatched query simprint in ascending order.  The reference walks
        # that second part in Python per asset -- O(assets x queries): 14 s for 512 query simprints that match 200 000 assets
        # (10 M random 64-bit chunks, profiles/r03_simprint_end_to_end.txt).  Here the unmatched part is one running sum per
        # asset over the row [matched total, idf(q0) or 0.0, idf(q1) or 0.0, ...] -- np.cumsum accumulates left to right in
        # float64 and x + 0.0 == x, so every asset's total has the bits the loop would give.
        nq = len(simprints)
        assets = list(asset_best.items())
        starts = np.zeros(len(assets), dtype=np.float64)
        weighteds = [0.0] * len(assets)
        stored_all = []
        matched_by = np.zeros(nq, dtype=np.int64)
        for ai, (asset_id, best) in enumerate(assets):
            total_idf = 0.0
            weighted = 0.0
            stored = {}
            for qi, (offset, size, sim, ckey) in best.items():
                vec = fetched.get(ckey)
                match_bytes = vec.tobytes() if vec is not None else simprints[qi]
                stored[qi] = match_bytes
                idf = calculate_idf(get_freq(match_bytes), total_assets)
                total_idf += idf
                weighted += idf * sim
                matched_by[qi] += 1
            starts[ai], weighteds[ai] = total_idf, weighted
            stored_all.append(stored)
        idf_q = np.zeros(nq, dtype=np.float64)
        for qi in range(nq):
            if matched_by[qi] < len(assets):      # some asset lacks it: only then does the reference ask for its frequency
                idf_q[qi] = calculate_idf(get_freq(simprints[qi]), total_assets)
        totals = np.empty(len(assets), dtype=np.float64)
        for lo in range(0, len(assets), 4096):
            part = assets[lo : lo + 4096]
            rows = np.empty((len(part), nq + 1), dtype=np.float64)
            rows[:, 0] = starts[lo : lo + len(part)]
            rows[:, 1:] = idf_q
            r_idx = np.fromiter((i for i, (_, best) in enumerate(part) for _ in best), dtype=np.int64)
            q_idx = np.fromiter((qi for _, best in part for qi in best), dtype=np.int64)
            rows[r_idx, q_idx + 1] = 0.0
            totals[lo : lo + len(part)] = np.cumsum(rows, axis=1)[:, -1]
        results = []
        for ai, (asset_id, best) in enumerate(assets):
            total_idf = float(totals[ai])
            asset_score = weighteds[ai] / total_idf if total_idf > 0 else 0.0
            chunks = None
            if detailed:
                stored = stored_all[ai]
                chunks = [
                    MatchedChunkRaw(query=simprints[qi], match=stored[qi], score=sim, offset=offset, size=size, freq=get_freq(stored[qi]))
                    for qi, (offset, size, sim, ckey) in best.items()
                ]
            results.append(SimprintMatchRaw(iscc_id_body=asset_id, score=asset_score, queried=len(simprints), matches=len(best), chunks=chunks))
        results.sort(key=lambda r: (-r.score, r.iscc_id_body))
        return results[:limit]

    def search_exact(self, simprints, limit=10, threshold=0.0, detailed=False, dup_limit=1000):
        # type: (list[bytes], int, float, bool, int) -> list[SimprintMatchRaw]
        """
        Hard-boundary search: only stored simprints EQUAL to a query simprint match; assets are scored by
        coverage x quality (``lmdb_ops.search_simprints_exact``, ``lmdb_ops.py:169-249``).

        The reference walks the LMDB duplicates of each simprint key (at most ``dup_limit``, in chunk-pointer
        byte order); here one range-limited scan with ``max_hamming=0`` lists the same rows in the same
        order for all distinct query simprints at once.
        """
        if not simprints or len(self._index) == 0:
            return []
        nbytes = self.ndim // 8
        given = [bytes(s) for s in simprints]
        distinct = [sp for sp in dict.fromkeys(given) if len(sp) == nbytes]
        # (a handful of lookups: their lists are tiny and the host scores them in less time than the device path's extra launches take)
        if distinct and self._index.scores_exact_on_device and EXACT_DEVICE_FROM <= len(given) <= MAX_SCORED_SIMPRINTS:
            return self._search_exact_device(given, distinct, limit, threshold, detailed, dup_limit)
        return self._search_exact_host(given, distinct, limit, threshold, detailed, dup_limit)

    def _search_exact_device(self, given, distinct, limit, threshold, detailed, dup_limit):
        # type: (list[bytes], list[bytes], int, float, bool, int) -> list[SimprintMatchRaw]
        """
        ``lmdb_ops.search_simprints_exact`` in ONE library call (``isccsearch_simprint_exact``): the collision lists stay in device
        memory; document frequencies, the matches of every asset in visiting order, coverage x quality in the reference's float64
        operations, threshold, order and cut are kernels of ``csrc/simprint_score.hip``.
        """
        index_of = {sp: i for i, sp in enumerate(distinct)}
        valid = [sp for sp in given if sp in index_of]              # (a simprint of another length matches nothing)
        lookups = np.fromiter((index_of[sp] for sp in valid), dtype=np.uint32, count=len(valid))
        results, chunks, _ = self._index.exact_assets(_pack_simprints(distinct), lookups, len(given), max(1, dup_limit), threshold, limit, detailed)
        body = [a.to_bytes(8, "big") for a in results["asset"].tolist()]
        scores, matches = results["score"].tolist(), results["matches"].tolist()
        if not detailed:
            return [SimprintMatchRaw(b, s, len(given), m, None) for b, s, m in zip(body, scores, matches)]
        key_lo = chunks["key_lo"]
        cols = zip(chunks["query"].tolist(), (key_lo >> np.uint64(32)).tolist(), (key_lo & np.uint64(0xFFFFFFFF)).tolist(), chunks["freq"].tolist())
        detail = [MatchedChunkRaw(valid[g], valid[g], 1.0, o, z, f) for g, o, z, f in cols]
        return [SimprintMatchRaw(b, s, len(given), m, detail[f : f + m]) for b, s, m, f in zip(body, scores, matches, results["first_chunk"].tolist())]

    def _search_exact_host(self, simprints, distinct, limit, threshold, detailed, dup_limit):
        # type: (list[bytes], list[bytes], int, float, bool, int) -> list[SimprintMatchRaw]
        """The same search scored on the host (tables sharded over several GPUs; the checker of the device path in the tests)."""
        hits = {}        # distinct query simprint -> [(asset body, offset, size)] in ascending key order
        if distinct:
            # A simprint's collisions are few: a short list first (64 records per query to select, copy and unpack instead of
            # dup_limit = 1 000 -- the result block of 512 lookups shrinks from 12 MB to 0.8 MB); only the lookups whose list comes
            # back full are repeated at the reference's limit.  Same rows in the same order either way (ascending key).
            # The lists are taken apart as ARRAYS (one key conversion for all hits): a Matches object per lookup cost 1 ms per 512.
            cap = min(MAX_K, max(1, dup_limit))
            first = min(cap, EXACT_FIRST_K)
            queries = _pack_simprints(distinct)

            def lookup(rows, count):
                key_words, _, cnt = self._index.search_arrays(queries[rows], count=count, max_hamming=0)
                cnt = np.minimum(cnt, count).astype(np.int64)
                flat = key_words[np.arange(count)[None, :] < cnt[:, None]]            # [hits, 2], query by query, ascending key
                bodies = flat[:, 0].astype(">u8").tobytes()
                lo = flat[:, 1]
                offs, sizes = (lo >> np.uint64(32)).tolist(), (lo & np.uint64(0xFFFFFFFF)).tolist()
                ends = np.cumsum(cnt).tolist()
                begin = 0
                for i, end in zip(rows, ends):
                    hits[distinct[i]] = [(bodies[8 * j : 8 * j + 8], offs[j], sizes[j]) for j in range(begin, end)]
                    begin = end

            lookup(list(range(len(distinct))), first)
            again = [i for i, sp in enumerate(distinct) if len(hits[sp]) == first] if first < cap else []
            if again:
                lookup(again, cap)

        asset_matches = defaultdict(list)   # asset body -> [(query simprint, offset, size)]
        doc_freq = {}
        for sp in simprints:                # as given: a repeated query simprint is matched again (:197)
            found = hits.get(sp)
            if not found:
                continue
            for body, offset, size in found:
                asset_matches[body].append((sp, offset, size))
            if sp not in doc_freq:
                doc_freq[sp] = len({body for body, _, _ in found})

        queried = len(simprints)
        results = []
        for body, matches in asset_matches.items():
            score = coverage_quality_score([m[0] for m in matches], doc_freq, queried)
            if score < threshold:
                continue
            chunks = None
            if detailed:
                chunks = [MatchedChunkRaw(sp, sp, 1.0, offset, size, doc_freq.get(sp, 1)) for sp, offset, size in matches]
            results.append(SimprintMatchRaw(iscc_id_body=body, score=score, queried=queried, matches=len(matches), chunks=chunks))
        results.sort(key=lambda r: (-r.score, r.iscc_id_body))
        return results[:limit]

    def doc_freq(self, simprints, dup_limit=1000):
        # type: (list[bytes], int) -> list[int]
        """Distinct assets holding each simprint, counted on the device (``lmdb_ops.count_doc_freq``, :139-166)."""
        if not simprints:
            return []
        if len(self._index) == 0:
            return [0] * len(simprints)
        nbytes = self.ndim // 8
        ok = [i for i, sp in enumerate(simprints) if len(sp) == nbytes]
        out = [0] * len(simprints)
        if ok:
            freq = self._index.doc_freq(_pack_simprints([bytes(simprints[i]) for i in ok]), dup_limit)
            for i, f in zip(ok, freq):
                out[i] = int(f)
        return out

    def save(self, path):
        # type: (str) -> None
        self._index.save(path)

    def load(self, path):
        # type: (str) -> None
        self._index.load(path)

    def rows(self):
        """(composite key, simprint bytes) of every stored chunk."""
        return self._index.rows()

    def reset(self):
        self._index.reset()

    close = reset
