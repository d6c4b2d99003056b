"""
Reference-shaped vector indexes over the HIP engine.

``HipNphdIndex``  ~ ``iscc_usearch.ShardedNphdIndex``  (variable-length codes, NPHD metric, u64 keys)
                    as used at ``iscc_search/indexes/usearch/index.py:1617-1625, :436-444, :560, :2036-2043``
``HipIndex128``   ~ ``iscc_usearch.ShardedIndex128``   (fixed ``ndim``, Hamming metric, 128-bit keys)
                    as used at ``iscc_search/indexes/simprint/usearch_core.py:73-83, :100-135, :165, :221``

Same method names, argument meaning and error behaviour as those call sites rely on; the search is
exact instead of approximate (HNSW), ties ordered by ascending key.  Both classes take an *engine*
(anything with ``open_table(metric, key_words, max_bytes)``): ``HipEngine`` in production.
"""

import numpy as np

from iscc_search_amd._lib import METRIC_HAMMING, METRIC_NPHD
from iscc_search_amd.engine import pack_bytes, unpack_words


class Matches:
    """Result of one query: ``keys`` and ``distances`` ascending (usearch ``Matches`` shape)."""

    __slots__ = ("keys", "distances", "hamming", "prefix_bits")

    def __init__(self, keys, distances, hamming, prefix_bits):
        self.keys = keys
        self.distances = distances
        self.hamming = hamming
        self.prefix_bits = prefix_bits

    def __len__(self):
        return len(self.keys)

    def to_list(self):
        return [(k if isinstance(k, bytes) else int(k), float(d)) for k, d in zip(self.keys, self.distances)]


class BatchMatches:
    """Results of a batch of queries; indexable like usearch ``BatchMatches``."""

    def __init__(self, matches):
        self._matches = matches

    def __len__(self):
        return len(self._matches)

    def __getitem__(self, i):
        return self._matches[i]

    def __iter__(self):
        return iter(self._matches)


def _drop_present(table, keys, words, nbytes):
    """
    usearch with multi=False keeps the ORIGINAL vector when a key is added again and skips the new one
    silently (tests/test_usearch_add.py:53-62); inside one batch the first occurrence wins.
    """
    present = table.contains(keys)
    _, first = np.unique(keys, return_index=True)
    keep = np.zeros(len(keys), dtype=bool)
    keep[first] = True
    keep &= ~present
    if keep.all():
        return keys, words, nbytes
    return keys[keep], words[keep], (None if nbytes is None else nbytes[keep])


def _as_bytes(v):
    if isinstance(v, (bytes, bytearray, memoryview)):
        return bytes(v)
    return np.asarray(v, dtype=np.uint8).tobytes()


class HipNphdIndex:
    """Variable-length binary codes under the Normalized Prefix Hamming Distance, 64-bit integer keys."""

    def __init__(self, engine, max_dim=256):
        # type: (object, int) -> None
        if max_dim % 8 or not 8 <= max_dim <= 256:
            raise ValueError("max_dim must be a multiple of 8 bits up to 256")
        self.max_dim = max_dim
        self._table = engine.open_table(METRIC_NPHD, 1, max_dim // 8)

    # -- mutation --------------------------------------------------------------------------------
    def add(self, keys, vectors):
        # type: (int | list[int] | np.ndarray, object) -> None
        """Append (key, vector) rows.  Keys must not be present already (remove first to update)."""
        if np.isscalar(keys):
            keys, vectors = [keys], [vectors]
        keys = np.asarray(keys, dtype=np.uint64)
        codes = [_as_bytes(v) for v in vectors] if not (isinstance(vectors, np.ndarray) and vectors.ndim == 2) else vectors
        if len(keys) != len(codes):
            raise ValueError("keys and vectors differ in length")
        if len(keys) == 0:
            return
        words, nbytes = pack_bytes(codes, self._table.max_words)
        keys, words, nbytes = _drop_present(self._table, keys, words, nbytes)
        if len(keys):
            self._table.add(keys, words, nbytes)

    def remove(self, keys):
        # type: (int | list[int] | np.ndarray) -> int
        if np.isscalar(keys):
            keys = [keys]
        return self._table.remove(np.asarray(keys, dtype=np.uint64))

    # -- lookup ----------------------------------------------------------------------------------
    def __contains__(self, key):
        return bool(self._table.contains(np.asarray([key], dtype=np.uint64))[0])

    def contains(self, keys):
        # type: (list[int] | np.ndarray) -> np.ndarray
        return self._table.contains(np.asarray(keys, dtype=np.uint64))

    def get(self, key):
        # type: (int) -> np.ndarray | None
        words, nb = self._table.get(np.asarray([key], dtype=np.uint64))
        if nb[0] == 0:
            return None
        return np.frombuffer(unpack_words(words[0], int(nb[0])), dtype=np.uint8)

    @property
    def size(self):
        return self._table.size

    def __len__(self):
        return self._table.size

    # -- search ----------------------------------------------------------------------------------
    def search(self, vectors, count=10):
        # type: (object, int) -> Matches | BatchMatches
        """
        Exact ``count`` nearest rows.  One vector (1-D array / bytes) returns ``Matches``; a list of
        vectors or a 2-D array returns ``BatchMatches``.  ``distances`` are float32 NPHD values.
        """
        return self._search(vectors, count, None)

    def search_within(self, vectors, count, max_hamming=0):
        # type: (object, int, int) -> Matches | BatchMatches
        """
        As ``search`` but only rows whose Hamming distance over the compared prefix is <= ``max_hamming``
        (one streaming pass at a fixed threshold).  ``max_hamming=0`` is the prefix-equality match the
        reference runs for INSTANCE units (``usearch/index.py:1957-2022``).
        """
        return self._search(vectors, count, int(max_hamming))

    def _search(self, vectors, count, max_hamming):
        if count < 1:
            raise ValueError("`count` must be >= 1")
        single = isinstance(vectors, (bytes, bytearray)) or (isinstance(vectors, np.ndarray) and vectors.ndim == 1)
        if single:
            codes = [_as_bytes(vectors)]
        elif isinstance(vectors, np.ndarray) and vectors.ndim == 2:
            codes = vectors
        else:
            codes = [_as_bytes(v) for v in vectors]
        q_words, q_nbytes = pack_bytes(codes, self._table.max_words)
        if max_hamming is None:
            keys, ham, pbits, cnt = self._table.search(q_words, q_nbytes, count)
        else:
            keys, ham, pbits, cnt = self._table.search_within(q_words, q_nbytes, count, max_hamming)
        out = []
        for q in range(q_words.shape[0]):
            c = int(cnt[q])
            dist = ham[q, :c].astype(np.float32) / pbits[q, :c].astype(np.float32)
            out.append(Matches(keys[q, :c].copy(), dist, ham[q, :c].copy(), pbits[q, :c].copy()))
        return out[0] if single else BatchMatches(out)

    # -- lifecycle: the reference persists HNSW shard files (save/load); here a snapshot is the raw columns
    def save(self, path):
        # type: (str) -> None
        self._table.save(path)

    def load(self, path):
        # type: (str) -> None
        self._table.load(path)

    def reset(self):
        self._table.drop()

    close = reset


class HipHammingIndex:
    """
    Fixed-length binary vectors, Hamming metric, 64-bit keys: the subset of
    ``usearch.index.Index(ndim, metric=MetricKind.Hamming, dtype=ScalarKind.B1)`` that the reference's
    characterisation tests pin (tests/test_usearch_{add,get,contains,remove,search}.py), always exact.
    """

    def __init__(self, engine, ndim):
        # type: (object, int) -> None
        if ndim % 8 or not 8 <= ndim <= 256:
            raise ValueError("ndim must be a multiple of 8 bits up to 256")
        self.ndim = ndim
        self.nbytes = ndim // 8
        self._table = engine.open_table(METRIC_HAMMING, 1, self.nbytes)

    def _vectors(self, vectors):
        arr = np.asarray(vectors, dtype=np.uint8)
        if arr.ndim == 1:
            arr = arr.reshape(1, -1)
        if arr.ndim != 2 or arr.shape[1] != self.nbytes:
            raise ValueError(f"vectors must have {self.nbytes} bytes ({self.ndim} bits)")
        return arr

    def add(self, keys, vectors):
        # type: (int | list[int] | np.ndarray, np.ndarray) -> np.ndarray
        keys = np.atleast_1d(np.asarray(keys, dtype=np.uint64))
        arr = self._vectors(vectors)
        if len(keys) != arr.shape[0]:
            raise ValueError("keys and vectors differ in length")
        words, _ = pack_bytes(arr, self._table.max_words)
        k2, w2, _ = _drop_present(self._table, keys, words, None)
        if len(k2):
            self._table.add(k2, w2)
        return keys

    def remove(self, keys):
        # type: (int | list[int] | np.ndarray) -> int
        keys = np.atleast_1d(np.asarray(keys, dtype=np.uint64))
        return self._table.remove(keys) if len(keys) else 0

    def contains(self, keys):
        # type: (int | list[int] | np.ndarray) -> bool | np.ndarray
        if np.isscalar(keys):
            return bool(self._table.contains(np.asarray([keys], dtype=np.uint64))[0])
        return self._table.contains(np.asarray(keys, dtype=np.uint64))

    def __contains__(self, key):
        return self.contains(key)

    def get(self, keys):
        # type: (int | list[int]) -> np.ndarray | None | list
        single = np.isscalar(keys)
        arr = np.atleast_1d(np.asarray(keys, dtype=np.uint64))
        words, nb = self._table.get(arr)
        out = [np.frombuffer(unpack_words(words[i], self.nbytes), dtype=np.uint8) if nb[i] else None for i in range(len(arr))]
        return out[0] if single else out

    def __len__(self):
        return self._table.size

    size = property(lambda self: self._table.size)

    def search(self, vectors, count=10):
        # type: (np.ndarray, int) -> Matches | BatchMatches
        if count < 1:
            raise ValueError("`count` must be >= 1")
        single = np.asarray(vectors).ndim == 1
        arr = self._vectors(vectors)
        q_words, _ = pack_bytes(arr, self._table.max_words)
        keys, ham, pbits, cnt = self._table.search(q_words, None, count)
        out = [Matches(keys[q, : cnt[q]].copy(), ham[q, : cnt[q]].astype(np.float32), ham[q, : cnt[q]].copy(), pbits[q, : cnt[q]].copy())
               for q in range(arr.shape[0])]
        return out[0] if single else BatchMatches(out)

    def reset(self):
        self._table.drop()

    close = reset


def table_rows(table, chunk_rows=1 << 20):
    """(16-byte key, code bytes) of every row of a table with 128-bit keys, segment by segment."""
    for nbytes, total in table.segments().items():
        for first in range(0, total, chunk_rows):
            n = min(chunk_rows, total - first)
            keys, cols = table.export_rows(nbytes, first, n)
            kb = words_to_key128(keys)
            raw = np.ascontiguousarray(cols.T).astype(">u8").tobytes()
            stride = cols.shape[0] * 8
            for i in range(n):
                yield kb[i], raw[i * stride : i * stride + nbytes]


def key128_to_words(keys):
    # type: (list[bytes] | np.ndarray) -> np.ndarray
    """16-byte big-endian keys -> uint64 [n, 2] (hi, lo)."""
    if isinstance(keys, np.ndarray) and keys.dtype == np.uint64 and keys.ndim == 2:
        return np.ascontiguousarray(keys)
    raw = b"".join(bytes(k) for k in keys)
    if len(raw) != 16 * len(keys):
        raise ValueError("composite keys must be 16 bytes each")
    return np.frombuffer(raw, dtype=">u8").astype(np.uint64).reshape(len(keys), 2)


def words_to_key128(words):
    # type: (np.ndarray) -> list[bytes]
    raw = np.ascontiguousarray(words, dtype=np.uint64).astype(">u8").tobytes()
    return [raw[i : i + 16] for i in range(0, len(raw), 16)]


class HipIndex128:
    """Fixed-length binary vectors under the Hamming metric, 128-bit (16-byte) keys."""

    def __init__(self, engine, ndim):
        # type: (object, int) -> None
        if ndim % 8 or not 8 <= ndim <= 256:
            raise ValueError("ndim must be a multiple of 8 bits up to 256")
        self.ndim = ndim
        self.nbytes = ndim // 8
        self._table = engine.open_table(METRIC_HAMMING, 2, self.nbytes)

    def _vectors(self, vectors):
        arr = np.asarray(vectors, dtype=np.uint8) if not isinstance(vectors, np.ndarray) else vectors.astype(np.uint8, copy=False)
        if arr.ndim == 1:
            arr = arr.reshape(1, -1)
        if arr.shape[1] != self.nbytes:
            raise ValueError(f"vectors must have {self.nbytes} bytes ({self.ndim} bits), got {arr.shape[1]}")
        return arr

    def add(self, keys, vectors, trusted_unique=False):
        # type: (list[bytes] | np.ndarray, np.ndarray, bool) -> None
        kw = key128_to_words(keys)
        arr = self._vectors(vectors)
        if kw.shape[0] != arr.shape[0]:
            raise ValueError("keys and vectors differ in length")
        if kw.shape[0] == 0:
            return
        words, _ = pack_bytes(arr, self._table.max_words)
        self._table.add(kw, words, None, trusted_unique=trusted_unique)

    def remove(self, keys):
        # type: (list[bytes] | np.ndarray) -> int
        if len(keys) == 0:
            return 0
        return self._table.remove(key128_to_words(keys))

    def __contains__(self, key):
        return bool(self._table.contains(key128_to_words([key]))[0])

    def get(self, key):
        # type: (bytes) -> np.ndarray | None
        words, nb = self._table.get(key128_to_words([key]))
        if nb[0] == 0:
            return None
        return np.frombuffer(unpack_words(words[0], self.nbytes), dtype=np.uint8)

    def get_many(self, keys):
        # type: (list[bytes]) -> list[np.ndarray | None]
        """Stored vectors of many keys in one device round trip (None for absent keys)."""
        if not keys:
            return []
        words, nb = self._table.get(key128_to_words(keys))
        return [np.frombuffer(unpack_words(words[i], self.nbytes), dtype=np.uint8) if nb[i] else None for i in range(len(keys))]

    def __len__(self):
        return self._table.size

    @property
    def size(self):
        return self._table.size

    def search(self, vectors, count=10):
        # type: (np.ndarray, int) -> Matches | BatchMatches
        """``distances`` are raw differing-bit counts as float32 (``tests/test_usearch_search.py:141``)."""
        if count < 1:
            raise ValueError("`count` must be >= 1")
        single = isinstance(vectors, np.ndarray) and vectors.ndim == 1
        arr = self._vectors(vectors)
        q_words, _ = pack_bytes(arr, self._table.max_words)
        keys, ham, pbits, cnt = self._table.search(q_words, None, count)
        out = []
        for q in range(arr.shape[0]):
            c = int(cnt[q])
            out.append(Matches(words_to_key128(keys[q, :c]), ham[q, :c].astype(np.float32), ham[q, :c].copy(), pbits[q, :c].copy()))
        return out[0] if single else BatchMatches(out)

    def search_arrays(self, vectors, count=10, max_hamming=None):
        # type: (np.ndarray, int, int | None) -> tuple[np.ndarray, np.ndarray, np.ndarray]
        """
        The same search as raw arrays, for callers that filter before they materialise keys:
        (key words uint64 [nq, count, 2], differing bits uint32 [nq, count], valid entries uint32 [nq]).
        With ``max_hamming`` only rows within that radius are listed (nearest first).
        """
        if count < 1:
            raise ValueError("`count` must be >= 1")
        q_words, _ = pack_bytes(self._vectors(vectors), self._table.max_words)
        if max_hamming is None:
            keys, ham, _, cnt = self._table.search(q_words, None, count)
        else:
            keys, ham, _, cnt = self._table.search_within(q_words, None, count, int(max_hamming))
        return keys, ham, cnt

    def search_within(self, vectors, count, max_hamming=0):
        # type: (np.ndarray, int, int) -> Matches | BatchMatches
        """
        Only the rows within ``max_hamming`` bits, nearest first, ties by ascending key, at most ``count``.
        ``max_hamming=0`` lists the collisions of each query in the order LMDB iterates the duplicates of a
        simprint key (``lmdb_ops.py:197-203``).
        """
        if count < 1:
            raise ValueError("`count` must be >= 1")
        single = isinstance(vectors, np.ndarray) and vectors.ndim == 1
        arr = self._vectors(vectors)
        q_words, _ = pack_bytes(arr, self._table.max_words)
        keys, ham, pbits, cnt = self._table.search_within(q_words, None, count, max_hamming)
        out = []
        for q in range(arr.shape[0]):
            c = int(cnt[q])
            out.append(Matches(words_to_key128(keys[q, :c]), ham[q, :c].astype(np.float32), ham[q, :c].copy(), pbits[q, :c].copy()))
        return out[0] if single else BatchMatches(out)

    @property
    def scores_on_device(self):
        # type: () -> bool
        """Whether the table behind this index scores simprint matches itself (``isccsearch_simprint_score``; a sharded table does not)."""
        return hasattr(self._table, "simprint_score")

    @property
    def scores_exact_on_device(self):
        # type: () -> bool
        """Whether the table scores hard-boundary (collision) searches itself (``isccsearch_simprint_exact``)."""
        return hasattr(self._table, "simprint_exact")

    def score_assets(self, vectors, count, max_hamming, threshold, limit, total_assets, dup_limit, detailed):
        # type: (np.ndarray, int, int | None, float, int, int, int, bool) -> tuple
        """``HipTable.simprint_score`` for byte vectors: search + scoring of ``usearch_core.py:137-269`` in one device round trip."""
        q_words, _ = pack_bytes(self._vectors(vectors), self._table.max_words)
        return self._table.simprint_score(q_words, count, max_hamming, threshold, limit, total_assets, dup_limit, detailed)

    def exact_assets(self, vectors, given, queried, dup_limit, threshold, limit, detailed):
        # type: (np.ndarray, np.ndarray, int, int, float, int, bool) -> tuple
        """``HipTable.simprint_exact`` for byte vectors: collision search + coverage x quality scoring of ``lmdb_ops.py:169-301`` on the device."""
        q_words, _ = pack_bytes(self._vectors(vectors), self._table.max_words)
        return self._table.simprint_exact(q_words, given, queried, dup_limit, threshold, limit, detailed)

    def get_freq(self, keys, dup_limit=1000):
        # type: (list[bytes], int) -> np.ndarray
        """Document frequency of the vector stored under each key (0 for absent keys), from the frequency column."""
        if len(keys) == 0:
            return np.zeros(0, dtype=np.uint32)
        return self._table.get_freq(key128_to_words(keys), dup_limit)

    def doc_freq(self, vectors, dup_limit=1000):
        # type: (np.ndarray, int) -> np.ndarray
        """Distinct assets (key[:8]) among the first ``dup_limit`` rows equal to each vector (``lmdb_ops.py:139-166``)."""
        arr = self._vectors(vectors)
        q_words, _ = pack_bytes(arr, self._table.max_words)
        return self._table.doc_freq(q_words, None, dup_limit)

    def save(self, path):
        # type: (str) -> None
        self._table.save(path)

    def load(self, path):
        # type: (str) -> None
        self._table.load(path)

    def rows(self, chunk_rows=1 << 20):
        """Iterate (key bytes, vector bytes) over every stored row (snapshot restore of host-side maps)."""
        if hasattr(self._table, "rows"):          # a leader-front table gathers the rows of every shard itself
            return self._table.rows()
        return table_rows(self._table, chunk_rows)

    def reset(self):
        self._table.drop()

    close = reset
