"""
numpy-level wrapper of the C-ABI: one ``HipEngine`` per GPU, ``HipTable`` per code table.

This is the thinnest layer above ``include/isccsearch.h``; the reference-shaped objects
(``HipNphdIndex`` ~ ``iscc_usearch.ShardedNphdIndex``, ``HipIndex128`` ~ ``ShardedIndex128``) live in
``iscc_search_amd/nphd.py``.  Nothing here computes distances on the CPU.
"""

import ctypes
import threading

import numpy as np

from iscc_search_amd import _lib


def pack_bytes(codes, max_words):
    # type: (list[bytes] | np.ndarray, int) -> tuple[np.ndarray, np.ndarray]
    """
    Pack codes big-endian into zero-padded uint64 words (the C-ABI layout).

    :param codes: list of byte strings (any lengths 1..max_words*8) or a 2-D uint8 array (equal lengths)
    :param max_words: words per row in the output
    :return: (words uint64 [n, max_words], nbytes uint8 [n])
    """
    width = max_words * 8
    if isinstance(codes, np.ndarray) and codes.ndim == 2:
        arr = np.ascontiguousarray(codes, dtype=np.uint8)
        n, nb = arr.shape
        if not 1 <= nb <= width:
            raise ValueError(f"code length {nb} bytes outside 1..{width}")
        buf = np.zeros((n, width), dtype=np.uint8)
        buf[:, :nb] = arr
        lens = np.full(n, nb, dtype=np.uint8)
    else:
        n = len(codes)
        if n == 1 and type(codes[0]) is bytes:
            # one code -- the per-unit searches of a request -- without the row-by-row staging below (7.5 -> 2 us)
            c = codes[0]
            if not 1 <= len(c) <= width:
                raise ValueError(f"code length {len(c)} bytes outside 1..{width}")
            return np.frombuffer(c.ljust(width, b"\0"), dtype=">u8").astype(np.uint64).reshape(1, max_words), np.array([len(c)], dtype=np.uint8)
        buf = np.zeros((n, width), dtype=np.uint8)
        lens = np.zeros(n, dtype=np.uint8)
        for i, c in enumerate(codes):
            c = bytes(c) if not isinstance(c, np.ndarray) else c.astype(np.uint8).tobytes()
            if not 1 <= len(c) <= width:
                raise ValueError(f"code length {len(c)} bytes outside 1..{width}")
            buf[i, : len(c)] = np.frombuffer(c, dtype=np.uint8)
            lens[i] = len(c)
    words = buf.view(">u8").astype(np.uint64).reshape(n, max_words)
    return words, lens


def unpack_words(words, nbytes):
    # type: (np.ndarray, int) -> bytes
    """Inverse of ``pack_bytes`` for one row."""
    return np.ascontiguousarray(words, dtype=np.uint64).astype(">u8").tobytes()[:nbytes]


class HipEngine:
    """One engine handle = one GPU = one HIP stream (``isccsearch_create``)."""

    def __init__(self, device_id=0):
        # type: (int) -> None
        self._lib = _lib.load_library()
        h = ctypes.c_void_p()
        _lib.check(self._lib.isccsearch_create(int(device_id), ctypes.byref(h)))
        self._h = h
        self.device_id = int(device_id)
        self._lock = threading.Lock()

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError("engine is closed")
        return self._h

    def open_table(self, metric, key_words, max_bytes):
        # type: (int, int, int) -> HipTable
        tid = ctypes.c_uint32()
        _lib.check(self._lib.isccsearch_table_open(self.handle, metric, key_words, max_bytes, ctypes.byref(tid)))
        return HipTable(self, tid.value, metric, key_words, max_bytes)

    def set_option(self, name, value):
        # type: (str, int) -> None
        _lib.check(self._lib.isccsearch_set_option(self.handle, name.encode(), int(value)))

    def stats(self, reset=False):
        # type: (bool) -> dict
        st = _lib.Stats()
        _lib.check(self._lib.isccsearch_stats_get(self.handle, ctypes.byref(st), 1 if reset else 0))
        return st.as_dict()

    def merge_device(self, n_lists, nq, k, key_words, d_records_ptr, d_counts_ptr, list_stride, count_stride, after_stream=None):
        # type: (int, int, int, int, int, int, int, int, int | None) -> tuple
        """
        k-way merge of per-shard result blocks held in device memory (``isccsearch_merge_device``).  With ``after_stream``
        (a HIP stream handle) the merge is ordered behind that stream on the device instead of by a host synchronisation
        (``isccsearch_merge_device_after``); a count of ``_lib.COUNT_OVERFLOW`` then marks a query some shard could not
        complete asynchronously.
        """
        out, outs = _alloc_out(nq, k, key_words)
        args = (self.handle, n_lists, nq, k, key_words, ctypes.c_void_p(d_records_ptr), ctypes.c_void_p(d_counts_ptr), list_stride, count_stride)
        if after_stream is None:
            _lib.check(self._lib.isccsearch_merge_device(*args, *outs))
        else:
            _lib.check(self._lib.isccsearch_merge_device_after(*args, ctypes.c_void_p(after_stream), *outs))
        return out

    def stream(self):
        # type: () -> int
        """The library's hipStream_t (``isccsearch_stream``): work a caller queues on it is in order with the library's own."""
        return int(self._lib.isccsearch_stream(self.handle) or 0)

    def merge_many(self, merges, after_stream):
        # type: (list[tuple], int) -> list[tuple]
        """
        Several ``merge_device`` calls behind ONE synchronisation (``isccsearch_merge_many_after``).
        ``merges`` = [(n_lists, nq, k, key_words, d_records_ptr, d_counts_ptr, list_stride, count_stride)].
        """
        arr = (_lib.MergeRequest * len(merges))()
        outs = []
        for r, (n_lists, nq, k, key_words, rec, cnt, ls, cs) in zip(arr, merges):
            out, addr = _alloc_out(nq, k, key_words)
            outs.append(out)
            r.n_lists, r.nq, r.k, r.key_words = n_lists, nq, k, key_words
            r.d_records, r.d_counts, r.list_stride, r.count_stride = rec, cnt, ls, cs
            r.out_keys, r.out_hamming, r.out_prefix_bits, r.out_count = addr
        _lib.check(self._lib.isccsearch_merge_many_after(self.handle, len(merges), arr, ctypes.c_void_p(after_stream)))
        return outs

    def search_many(self, requests):
        # type: (list[tuple]) -> list[tuple]
        """
        Several searches with ONE device synchronisation (``isccsearch_search_many``): the per-unit searches of
        one ``search_assets`` request (``usearch/index.py:786-806``).  ``requests`` = [(table, q_words, q_nbytes, k,
        max_hamming or None)]; returns [(keys, hamming, prefix_bits, count)] shaped as ``HipTable.search``.
        """
        n = len(requests)
        if n == 0:
            return []
        arr = (_lib.Request * n)()
        keep, outs = [], []
        for i, (table, q_words, q_nbytes, k, max_hamming) in enumerate(requests):
            if k < 1:
                raise ValueError("`count` must be >= 1")
            q_words = table._words(q_words)
            nq = q_words.shape[0]
            q_nbytes = table._nbytes(q_nbytes, nq)
            out, addr = _alloc_out(nq, k, table.key_words)
            keep.append((q_words, q_nbytes))
            outs.append(out)
            r = arr[i]
            r.table, r.nq, r.k = table.id, nq, k
            r.max_hamming = -1 if max_hamming is None else int(max_hamming)
            r.q_words = _lib.ptr(q_words)
            r.q_nbytes = _lib.ptr(q_nbytes)
            r.out_keys, r.out_hamming, r.out_prefix_bits, r.out_count = addr
        _lib.check(self._lib.isccsearch_search_many(self.handle, n, arr))
        return outs

    def close(self):
        # type: () -> None
        """Idempotent (``protocols/index.py:167-172``)."""
        with self._lock:
            if self._h is not None:
                self._lib.isccsearch_destroy(self._h)
                self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _alloc_out(nq, k, key_words):
    """
    The four result arrays of a search as views of ONE allocation, and their addresses (one lookup instead of four).

    The library writes EVERY slot (``unpack_records``: zeros beyond a query's count), so a large block is not zeroed first:
    ``np.zeros`` of the 3.3 MB a simprint-sized search returns (512 queries x 400) cost 0.2 ms of a 1.2 ms step (0.4 ms with
    128-bit keys); small blocks stay zeroed (microseconds).
    """
    nk = nq * k
    o_ham = nk * 8 * key_words
    o_pre = o_ham + nk * 4
    o_cnt = (o_pre + nk * 2 + 3) & ~3
    buf = (np.empty if nk >= 16384 else np.zeros)(o_cnt + nq * 4 + 8, dtype=np.uint8)
    base = buf.__array_interface__["data"][0]
    pad = -base & 7                                      # numpy aligns to 16 in practice; be exact anyway
    shape = (nq, k, 2) if key_words == 2 else (nq, k)
    out = (
        buf[pad : pad + o_ham].view(np.uint64).reshape(shape),
        buf[pad + o_ham : pad + o_pre].view(np.uint32).reshape(nq, k),
        buf[pad + o_pre : pad + o_pre + nk * 2].view(np.uint16).reshape(nq, k),
        buf[pad + o_cnt : pad + o_cnt + nq * 4].view(np.uint32),
    )
    return out, (base + pad, base + pad + o_ham, base + pad + o_pre, base + pad + o_cnt)


class HipTable:
    """A table of (key, code) rows resident in HBM."""

    def __init__(self, engine, table_id, metric, key_words, max_bytes):
        # type: (HipEngine, int, int, int, int) -> None
        self.engine = engine
        self.id = table_id
        self.metric = metric
        self.key_words = key_words
        self.max_bytes = max_bytes
        self.max_words = (max_bytes + 7) // 8
        self._open = True

    # -- helpers -----------------------------------------------------------------------------
    def _keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        if self.key_words == 2:
            if keys.ndim != 2 or keys.shape[1] != 2:
                raise ValueError("128-bit keys must be shaped [n, 2] (hi, lo)")
        elif keys.ndim != 1:
            raise ValueError("64-bit keys must be shaped [n]")
        return keys

    def _words(self, words, n=None):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        if words.ndim != 2 or words.shape[1] != self.max_words:
            raise ValueError(f"code words must be shaped [n, {self.max_words}]")
        if n is not None and words.shape[0] != n:
            raise ValueError("keys and codes differ in length")
        return words

    def _nbytes(self, nbytes, n):
        if self.metric == _lib.METRIC_HAMMING:
            if nbytes is not None and np.any(np.asarray(nbytes) != self.max_bytes):
                raise ValueError(f"Hamming table holds {self.max_bytes}-byte codes only")
            return None
        if nbytes is None:
            raise ValueError("nbytes is required for NPHD tables")
        nbytes = np.ascontiguousarray(nbytes, dtype=np.uint8)
        if nbytes.shape != (n,):
            raise ValueError("nbytes must be shaped [n]")
        return nbytes

    # -- C-ABI calls -------------------------------------------------------------------------
    def add(self, keys, words, nbytes=None, trusted_unique=False):
        # type: (np.ndarray, np.ndarray, np.ndarray | None, bool) -> None
        keys = self._keys(keys)
        n = keys.shape[0]
        if n == 0:
            return
        words = self._words(words, n)
        nbytes = self._nbytes(nbytes, n)
        flags = _lib.ADD_TRUSTED_UNIQUE if trusted_unique else 0
        lib = self.engine._lib
        _lib.check(
            lib.isccsearch_add(
                self.engine.handle, self.id, n, _lib.ptr(keys, ctypes.c_uint64), _lib.ptr(words, ctypes.c_uint64),
                _lib.ptr(nbytes, ctypes.c_uint8), flags,
            )
        )

    def add_synthetic(self, nbytes, n, seed, first_row=0, key_base=0):
        # type: (int, int, int, int, int) -> None
        lib = self.engine._lib
        _lib.check(lib.isccsearch_add_synthetic(self.engine.handle, self.id, nbytes, n, seed & (2**64 - 1), first_row, key_base))

    def reserve(self, nbytes, rows):
        # type: (int, int) -> None
        _lib.check(self.engine._lib.isccsearch_reserve(self.engine.handle, self.id, nbytes, rows))

    def remove(self, keys):
        # type: (np.ndarray) -> int
        keys = self._keys(keys)
        if keys.shape[0] == 0:
            return 0
        removed = ctypes.c_uint64()
        _lib.check(
            self.engine._lib.isccsearch_remove(
                self.engine.handle, self.id, keys.shape[0], _lib.ptr(keys, ctypes.c_uint64), ctypes.byref(removed)
            )
        )
        return int(removed.value)

    def contains(self, keys):
        # type: (np.ndarray) -> np.ndarray
        keys = self._keys(keys)
        out = np.zeros(keys.shape[0], dtype=np.uint8)
        if keys.shape[0]:
            _lib.check(
                self.engine._lib.isccsearch_contains(
                    self.engine.handle, self.id, keys.shape[0], _lib.ptr(keys, ctypes.c_uint64), _lib.ptr(out, ctypes.c_uint8)
                )
            )
        return out.astype(bool)

    def get(self, keys):
        # type: (np.ndarray) -> tuple[np.ndarray, np.ndarray]
        """Stored codes of ``keys``: (words [n, max_words], nbytes [n]; 0 = absent)."""
        keys = self._keys(keys)
        n = keys.shape[0]
        words = np.zeros((n, self.max_words), dtype=np.uint64)
        nb = np.zeros(n, dtype=np.uint8)
        if n:
            _lib.check(
                self.engine._lib.isccsearch_get(
                    self.engine.handle, self.id, n, _lib.ptr(keys, ctypes.c_uint64), _lib.ptr(words, ctypes.c_uint64),
                    _lib.ptr(nb, ctypes.c_uint8),
                )
            )
        return words, nb

    @property
    def size(self):
        # type: () -> int
        return int(self.engine._lib.isccsearch_size(self.engine.handle, self.id))

    def search(self, q_words, q_nbytes, k):
        # type: (np.ndarray, np.ndarray | None, int) -> tuple
        """Exact top-k: (keys [nq, k(,2)], hamming [nq, k], prefix_bits [nq, k], count [nq])."""
        q_words = self._words(q_words)
        nq = q_words.shape[0]
        if k < 1:
            raise ValueError("`count` must be >= 1")
        q_nbytes = self._nbytes(q_nbytes, nq)
        out, addr = _alloc_out(nq, k, self.key_words)
        if nq:
            _lib.check(self.engine._lib.isccsearch_search(self.engine.handle, self.id, nq, _lib.ptr(q_words), _lib.ptr(q_nbytes), k, *addr))
        return out

    def search_within(self, q_words, q_nbytes, k, max_hamming):
        # type: (np.ndarray, np.ndarray | None, int, int) -> tuple
        """
        Range-limited exact top-k: only rows within ``max_hamming`` bits over the compared prefix, nearest
        first, ties by ascending key.  ``max_hamming=0`` is the collision lookup of the reference's LMDB
        dupsort path (``lmdb_ops.py:169-249``).  Same return shape as ``search``; counts may be 0.
        """
        q_words = self._words(q_words)
        nq = q_words.shape[0]
        if k < 1:
            raise ValueError("`count` must be >= 1")
        q_nbytes = self._nbytes(q_nbytes, nq)
        out, addr = _alloc_out(nq, k, self.key_words)
        if nq:
            _lib.check(self.engine._lib.isccsearch_search_within(self.engine.handle, self.id, nq, _lib.ptr(q_words), _lib.ptr(q_nbytes), k, int(max_hamming), *addr))
        return out

    def doc_freq(self, q_words, q_nbytes=None, dup_limit=1000):
        # type: (np.ndarray, np.ndarray | None, int) -> np.ndarray
        """
        Distinct assets (first key word) among the first ``dup_limit`` rows equal to each code
        (``count_doc_freq``, ``lmdb_ops.py:139-166``) -> uint32 [nq].
        """
        q_words = self._words(q_words)
        nq = q_words.shape[0]
        q_nbytes = self._nbytes(q_nbytes, nq)
        out = np.zeros(nq, dtype=np.uint32)
        if nq:
            _lib.check(
                self.engine._lib.isccsearch_doc_freq(
                    self.engine.handle, self.id, nq, _lib.ptr(q_words, ctypes.c_uint64), _lib.ptr(q_nbytes, ctypes.c_uint8),
                    int(dup_limit), _lib.ptr(out, ctypes.c_uint32),
                )
            )
        return out

    def doc_freq_counted(self, q_words, q_nbytes=None, dup_limit=1000):
        # type: (np.ndarray, np.ndarray | None, int) -> tuple
        """``doc_freq`` and, beside it, how many colliding rows each count was taken over (<= dup_limit) -> (uint32 [nq], uint32 [nq])."""
        q_words = self._words(q_words)
        nq = q_words.shape[0]
        q_nbytes = self._nbytes(q_nbytes, nq)
        freq, coll = np.zeros(nq, dtype=np.uint32), np.zeros(nq, dtype=np.uint32)
        if nq:
            _lib.check(
                self.engine._lib.isccsearch_doc_freq_counted(
                    self.engine.handle, self.id, nq, _lib.ptr(q_words, ctypes.c_uint64), _lib.ptr(q_nbytes, ctypes.c_uint8),
                    int(dup_limit), _lib.ptr(freq, ctypes.c_uint32), _lib.ptr(coll, ctypes.c_uint32),
                )
            )
        return freq, coll

    def get_freq(self, keys, dup_limit=1000):
        # type: (np.ndarray, int) -> np.ndarray
        """
        Document frequency of the code stored under each key (0 for absent keys), from the segment's
        frequency column (built lazily on the device after the rows changed) -> uint32 [n].
        """
        keys = self._keys(keys)
        n = keys.shape[0]
        out = np.zeros(n, dtype=np.uint32)
        if n:
            _lib.check(self.engine._lib.isccsearch_get_freq(
                self.engine.handle, self.id, n, _lib.ptr(keys, ctypes.c_uint64), int(dup_limit), _lib.ptr(out, ctypes.c_uint32)))
        return out

    def simprint_score(self, q_words, count, max_hamming, threshold, limit, total_assets, dup_limit, detailed):
        # type: (np.ndarray, int, int | None, float, int, int, int, bool) -> tuple
        """
        Neighbour search + asset scoring in one call, the lists never leaving the device (``isccsearch_simprint_score``:
        ``usearch_core.py:137-269``).  Returns (results [n] ``SIMPRINT_RESULT_DTYPE``, chunks [c] ``SIMPRINT_CHUNK_DTYPE`` or
        None, chunk words uint64 [c, max_words] or None, info = (results, assets matched, longest neighbour list, chunks)).
        """
        q_words = self._words(q_words)
        nq = q_words.shape[0]
        if count < 1:
            raise ValueError("`count` must be >= 1")
        limit = int(limit)
        results = np.empty(limit, dtype=_lib.SIMPRINT_RESULT_DTYPE)
        info = np.zeros(4, dtype=np.uint32)
        chunks = words = None
        if detailed:
            cap = min(limit * nq, nq * int(count))
            chunks = np.empty(cap, dtype=_lib.SIMPRINT_CHUNK_DTYPE)
            words = np.empty((cap, self.max_words), dtype=np.uint64)
        if nq:
            _lib.check(self.engine._lib.isccsearch_simprint_score(
                self.engine.handle, self.id, nq, _lib.ptr(q_words), int(count), -1 if max_hamming is None else int(max_hamming),
                float(threshold), limit, int(total_assets), int(dup_limit),
                _lib.ptr(results), _lib.ptr(chunks), _lib.ptr(words), _lib.ptr(info)))
        n, c = int(info[0]), int(info[3])
        return results[:n], (chunks[:c] if detailed else None), (words[:c] if detailed else None), tuple(int(x) for x in info)

    def simprint_exact(self, q_words, given, queried, dup_limit, threshold, limit, detailed):
        # type: (np.ndarray, np.ndarray, int, int, float, int, bool) -> tuple
        """
        Hard-boundary search + coverage x quality scoring in one call (``isccsearch_simprint_exact``: ``lmdb_ops.py:169-301``).
        ``q_words`` = the distinct query simprints, ``given`` = for every query simprint as given its index among them.
        Returns (results, chunks or None, info).
        """
        q_words = self._words(q_words)
        given = np.ascontiguousarray(given, dtype=np.uint32)
        limit = int(limit)
        results = np.empty(limit, dtype=_lib.SIMPRINT_RESULT_DTYPE)
        info = np.zeros(4, dtype=np.uint32)
        cap = int(min(dup_limit, _lib.MAX_K)) * given.shape[0]
        chunks = np.empty(cap, dtype=_lib.SIMPRINT_CHUNK_DTYPE) if detailed else None
        if q_words.shape[0] and given.shape[0]:
            _lib.check(self.engine._lib.isccsearch_simprint_exact(
                self.engine.handle, self.id, q_words.shape[0], _lib.ptr(q_words), given.shape[0], _lib.ptr(given), int(queried), int(dup_limit),
                float(threshold), limit, _lib.ptr(results), _lib.ptr(chunks), _lib.ptr(info)))
        n, c = int(info[0]), int(info[3])
        return results[:n], (chunks[:c] if detailed else None), tuple(int(x) for x in info)

    def search_device(self, q_words, q_nbytes, k, d_records_ptr, d_counts_ptr, max_hamming=None, consumer_stream=None, hint=None):
        # type: (np.ndarray, np.ndarray | None, int, int, int, int | None, int | None, int | None) -> None
        """
        Same search (range-limited when ``max_hamming`` is given), results left in caller-owned device memory.  With
        ``consumer_stream`` (a HIP stream handle) the call does not wait for the GPU: that stream is made to wait for the
        results instead, and overflowed queries come back with a count of ``_lib.COUNT_OVERFLOW`` (``isccsearch_search_device_async``).
        ``hint`` (asynchronous top-k searches only): the single pass starts under this Hamming distance instead of a bootstrap
        sample's threshold, so the lists hold this table's nearest rows WITHIN the hint -- fewer than k if it was too tight;
        the caller must check that (``ShardedTable`` does, on the merged lists).
        """
        q_words = self._words(q_words)
        nq = q_words.shape[0]
        q_nbytes = self._nbytes(q_nbytes, nq)
        lib, args = self.engine._lib, (self.engine.handle, self.id, nq, _lib.ptr(q_words, ctypes.c_uint64), _lib.ptr(q_nbytes, ctypes.c_uint8), k)
        out = (ctypes.c_void_p(d_records_ptr), ctypes.c_void_p(d_counts_ptr))
        if consumer_stream is not None:
            if hint is not None and max_hamming is None:
                self.engine.set_option("device_search_hint", int(hint))      # one-shot, consumed by the call below
            _lib.check(lib.isccsearch_search_device_async(*args, -1 if max_hamming is None else int(max_hamming), *out, ctypes.c_void_p(consumer_stream)))
            return
        if max_hamming is None:
            _lib.check(lib.isccsearch_search_device(*args, *out))
        else:
            _lib.check(lib.isccsearch_search_within_device(*args, int(max_hamming), *out))

    # -- snapshot (raw little-endian column files) --------------------------
    def segments(self):
        # type: () -> dict[int, int]
        """{code length in bytes: rows} of the non-empty segments."""
        out = np.zeros(_lib.MAX_BYTES + 1, dtype=np.uint64)
        _lib.check(self.engine._lib.isccsearch_segments(self.engine.handle, self.id, _lib.ptr(out, ctypes.c_uint64)))
        return {b: int(out[b]) for b in range(1, _lib.MAX_BYTES + 1) if out[b]}

    def export_rows(self, nbytes, first_row, n):
        # type: (int, int, int) -> tuple[np.ndarray, np.ndarray]
        """(keys [n(,2)], cols [W, n]) of rows [first_row, first_row+n) of one segment, device layout."""
        W = (nbytes + 7) // 8
        keys = np.zeros((n, 2) if self.key_words == 2 else n, dtype=np.uint64)
        cols = np.zeros((W, n), dtype=np.uint64)
        if n:
            _lib.check(
                self.engine._lib.isccsearch_export(
                    self.engine.handle, self.id, nbytes, first_row, n, _lib.ptr(keys, ctypes.c_uint64), _lib.ptr(cols, ctypes.c_uint64)
                )
            )
        return keys, cols

    def add_columns(self, nbytes, keys, cols, trusted_unique=False):
        # type: (int, np.ndarray, np.ndarray, bool) -> None
        keys = self._keys(keys)
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        n = keys.shape[0]
        if cols.shape != ((nbytes + 7) // 8, n):
            raise ValueError(f"cols must be shaped [{(nbytes + 7) // 8}, {n}]")
        if n:
            _lib.check(
                self.engine._lib.isccsearch_add_columns(
                    self.engine.handle, self.id, nbytes, n, _lib.ptr(keys, ctypes.c_uint64), _lib.ptr(cols, ctypes.c_uint64),
                    _lib.ADD_TRUSTED_UNIQUE if trusted_unique else 0,
                )
            )

    def save(self, path, chunk_rows=1 << 24):
        # type: (str, int) -> None
        """
        Write the table as raw little-endian files: ``table.json`` + per segment ``segNN.keys.u64`` and
        ``segNN.wI.u64`` (one file per 64-bit word column, the device layout), streamed in chunks.
        """
        import json
        import os

        os.makedirs(path, exist_ok=True)
        segs = self.segments()
        for nbytes, rows in segs.items():
            W = (nbytes + 7) // 8
            files = [open(os.path.join(path, f"seg{nbytes:02d}.w{w}.u64.tmp"), "wb") for w in range(W)]
            kf = open(os.path.join(path, f"seg{nbytes:02d}.keys.u64.tmp"), "wb")
            try:
                for first in range(0, rows, chunk_rows):
                    n = min(chunk_rows, rows - first)
                    keys, cols = self.export_rows(nbytes, first, n)
                    keys.astype("<u8", copy=False).tofile(kf)
                    for w in range(W):
                        cols[w].astype("<u8", copy=False).tofile(files[w])
            finally:
                kf.close()
                for f in files:
                    f.close()
            for w in range(W):
                os.replace(os.path.join(path, f"seg{nbytes:02d}.w{w}.u64.tmp"), os.path.join(path, f"seg{nbytes:02d}.w{w}.u64"))
            os.replace(os.path.join(path, f"seg{nbytes:02d}.keys.u64.tmp"), os.path.join(path, f"seg{nbytes:02d}.keys.u64"))
        meta = {"format": 1, "metric": self.metric, "key_words": self.key_words, "max_bytes": self.max_bytes,
                "segments": {str(b): r for b, r in segs.items()}}
        with open(os.path.join(path, "table.json.tmp"), "w") as f:
            json.dump(meta, f)
        os.replace(os.path.join(path, "table.json.tmp"), os.path.join(path, "table.json"))

    def load(self, path, chunk_rows=1 << 24):
        # type: (str, int) -> None
        """Append the rows of a snapshot written by ``save`` (memory-mapped, streamed host -> device)."""
        import json
        import os

        with open(os.path.join(path, "table.json")) as f:
            meta = json.load(f)
        if (meta["metric"], meta["key_words"], meta["max_bytes"]) != (self.metric, self.key_words, self.max_bytes):
            raise ValueError(f"snapshot at {path} does not match this table's metric / key width / code length")
        for b_str, rows in meta["segments"].items():
            nbytes, rows = int(b_str), int(rows)
            W = (nbytes + 7) // 8
            kmap = np.memmap(os.path.join(path, f"seg{nbytes:02d}.keys.u64"), dtype="<u8", mode="r")
            cmaps = [np.memmap(os.path.join(path, f"seg{nbytes:02d}.w{w}.u64"), dtype="<u8", mode="r") for w in range(W)]
            if kmap.shape[0] != rows * self.key_words or any(c.shape[0] != rows for c in cmaps):
                raise ValueError(f"snapshot at {path}: segment {nbytes} is truncated")
            self.reserve(nbytes, self.segments().get(nbytes, 0) + rows)
            for first in range(0, rows, chunk_rows):
                n = min(chunk_rows, rows - first)
                keys = np.array(kmap[first * self.key_words : (first + n) * self.key_words], dtype=np.uint64)
                if self.key_words == 2:
                    keys = keys.reshape(n, 2)
                cols = np.stack([np.asarray(c[first : first + n], dtype=np.uint64) for c in cmaps])
                self.add_columns(nbytes, keys, cols, trusted_unique=True)

    def drop(self):
        # type: () -> None
        if self._open and self.engine._h is not None:
            _lib.check(self.engine._lib.isccsearch_table_drop(self.engine.handle, self.id))
        self._open = False
