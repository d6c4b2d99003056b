"""
Oracle-backed stand-in for ``HipEngine`` / ``HipTable`` (TEST INFRASTRUCTURE).

Lets the host logic (``nphd.py``, ``simprint.py``, ``index.py``, ``sharded.py``) run in the CPU test
tier with the CPU oracle answering the searches.  Lives under ``tests/`` because only tests may
route through ``oracle/``; the product path has no such engine.
"""

import numpy as np

from oracle import np_within, oracle_topk

RECORD_DTYPE = np.dtype(
    [("key_hi", "<u8"), ("key_lo", "<u8"), ("dist_rank", "<u4"), ("hamming", "<u2"), ("prefix_bits", "<u2")]
)


def rank_table():
    """Order-preserving rank of h/(8p) over all p in 1..32 bytes (independent of the C++ table)."""
    from fractions import Fraction

    vals = sorted({Fraction(h, 8 * p) for p in range(1, 33) for h in range(8 * p + 1)})
    pos = {v: i for i, v in enumerate(vals)}
    return {(p, h): pos[Fraction(h, 8 * p)] for p in range(1, 33) for h in range(8 * p + 1)}


class OracleTable:
    def __init__(self, metric, key_words, max_bytes):
        self.metric, self.key_words, self.max_bytes = metric, key_words, max_bytes
        self.max_words = (max_bytes + 7) // 8
        self._rows = {}  # key tuple -> (words tuple, nbytes)
        self.engine = None

    @staticmethod
    def _kt(k):
        return tuple(int(x) for x in np.atleast_1d(k))

    def add(self, keys, words, nbytes=None, trusted_unique=False):
        keys = np.asarray(keys, dtype=np.uint64)
        words = np.asarray(words, dtype=np.uint64)
        n = keys.shape[0]
        kts = [self._kt(keys[i]) for i in range(n)]
        if not trusted_unique:
            if len(set(kts)) != n or any(k in self._rows for k in kts):
                raise KeyError("key already present")
        for i, k in enumerate(kts):
            nb = int(nbytes[i]) if nbytes is not None else self.max_bytes
            if self.metric == 0 and nb != self.max_bytes:
                raise ValueError("Hamming table holds fixed-length codes")
            self._rows[k] = (words[i].copy(), nb)

    def remove(self, keys):
        keys = np.asarray(keys, dtype=np.uint64)
        removed = 0
        for i in range(keys.shape[0]):
            if self._rows.pop(self._kt(keys[i]), None) is not None:
                removed += 1
        return removed

    def contains(self, keys):
        keys = np.asarray(keys, dtype=np.uint64)
        return np.array([self._kt(keys[i]) in self._rows for i in range(keys.shape[0])], dtype=bool)

    def get(self, keys):
        keys = np.asarray(keys, dtype=np.uint64)
        n = keys.shape[0]
        words = np.zeros((n, self.max_words), dtype=np.uint64)
        nb = np.zeros(n, dtype=np.uint8)
        for i in range(n):
            row = self._rows.get(self._kt(keys[i]))
            if row is not None:
                words[i], nb[i] = row
        return words, nb

    @property
    def size(self):
        return len(self._rows)

    def _arrays(self):
        n = len(self._rows)
        keys = np.zeros((n, 2) if self.key_words == 2 else n, dtype=np.uint64)
        words = np.zeros((n, self.max_words), dtype=np.uint64)
        nb = np.zeros(n, dtype=np.uint8)
        for i, (k, (w, b)) in enumerate(self._rows.items()):
            keys[i] = k if self.key_words == 2 else k[0]
            words[i] = w
            nb[i] = b
        return keys, words, nb

    def search(self, q_words, q_nbytes, k):
        if k < 1:
            raise ValueError("`count` must be >= 1")
        q_words = np.asarray(q_words, dtype=np.uint64)
        nq = q_words.shape[0]
        keys, words, nb = self._arrays()
        if self.metric == 0:
            return oracle_topk(0, keys, words, None, q_words, None, k, fixed_nbytes=self.max_bytes)
        if q_nbytes is None:
            raise ValueError("nbytes is required for NPHD tables")
        if len(self._rows) == 0:
            shape = (nq, k, 2) if self.key_words == 2 else (nq, k)
            return np.zeros(shape, np.uint64), np.zeros((nq, k), np.uint32), np.zeros((nq, k), np.uint16), np.zeros(nq, np.uint32)
        return oracle_topk(1, keys, words, nb, q_words, np.asarray(q_nbytes, dtype=np.uint8), k)

    def search_within(self, q_words, q_nbytes, k, max_hamming):
        """Range-limited top-k by the vectorised numpy restatement (``oracle.np_within``)."""
        if k < 1:
            raise ValueError("`count` must be >= 1")
        q_words = np.asarray(q_words, dtype=np.uint64).reshape(-1, self.max_words)
        nq = q_words.shape[0]
        kshape = (nq, k, 2) if self.key_words == 2 else (nq, k)
        out = (np.zeros(kshape, np.uint64), np.zeros((nq, k), np.uint32), np.zeros((nq, k), np.uint16), np.zeros(nq, np.uint32))
        if not self._rows:
            return out
        keys, words, nb = self._arrays()
        for q in range(nq):
            qb = self.max_bytes if self.metric == 0 else int(q_nbytes[q])
            kk, h, p = np_within(words, nb, keys, q_words[q], qb, k, max_hamming)
            c = len(h)
            out[0][q, :c], out[1][q, :c], out[2][q, :c], out[3][q] = kk, h, p, c
        return out

    def doc_freq(self, q_words, q_nbytes=None, dup_limit=1000):
        keys, _, _, cnt = self.search_within(q_words, q_nbytes, dup_limit, 0)
        out = np.zeros(len(cnt), dtype=np.uint32)
        for q, c in enumerate(cnt):
            assets = keys[q, :c, 0] if self.key_words == 2 else keys[q, :c]
            out[q] = len(np.unique(assets))
        return out

    def doc_freq_counted(self, q_words, q_nbytes=None, dup_limit=1000):
        keys, _, _, cnt = self.search_within(q_words, q_nbytes, dup_limit, 0)
        return self.doc_freq(q_words, q_nbytes, dup_limit), cnt.astype(np.uint32)

    def get_freq(self, keys, dup_limit=1000):
        words, nb = self.get(keys)
        out = np.zeros(len(nb), dtype=np.uint32)
        for i in np.nonzero(nb)[0]:
            out[i] = self.doc_freq(words[i : i + 1], nb[i : i + 1], dup_limit)[0]
        return out

    def simprint_score(self, q_words, count, max_hamming, threshold, limit, total_assets, dup_limit, detailed):
        """``HipTable.simprint_score`` answered by the oracle's neighbours and the plain-loop checker (``tests/simprint_checker.py``)."""
        from iscc_search_amd._lib import SIMPRINT_CHUNK_DTYPE, SIMPRINT_RESULT_DTYPE
        from simprint_checker import score_lists

        q_words = np.asarray(q_words, dtype=np.uint64).reshape(-1, self.max_words)
        nq = q_words.shape[0]
        if max_hamming is None:
            keys, ham, _, cnt = self.search(q_words, None, count)
        else:
            keys, ham, _, cnt = self.search_within(q_words, None, count, max_hamming)
        to_bytes = lambda w, n: np.ascontiguousarray(w, dtype=np.uint64).astype(">u8").tobytes()[:n]
        lists = [[(to_bytes(keys[q, i], 16), int(ham[q, i])) for i in range(int(cnt[q]))] for q in range(nq)]
        simprints = [to_bytes(q_words[q], self.max_bytes) for q in range(nq)]

        def stored(key):
            words, nb = self.get(np.frombuffer(key, dtype=">u8").astype(np.uint64).reshape(1, 2))
            return to_bytes(words[0], self.max_bytes) if nb[0] else None

        def freq(sp_bytes):
            buf = np.zeros(self.max_words * 8, dtype=np.uint8)
            buf[: len(sp_bytes)] = np.frombuffer(sp_bytes, dtype=np.uint8)
            return int(self.doc_freq(buf.view(">u8").astype(np.uint64).reshape(1, -1), None, dup_limit)[0])

        scored = score_lists(simprints, lists, 8 * self.max_bytes, limit, threshold, stored, freq if dup_limit else None, total_assets)
        results = np.zeros(len(scored), dtype=SIMPRINT_RESULT_DTYPE)
        chunk_rows, word_rows = [], []
        for r, (asset, score, matches, chunks) in enumerate(scored):
            results[r] = (int.from_bytes(asset, "big"), score, matches, len(chunk_rows))
            for qi, match, sim, offset, size, f in chunks:
                h = [int(d) for (_, d) in lists[qi] if 1.0 - (float(d) / (8 * self.max_bytes)) == sim][0]
                chunk_rows.append(((offset << 32) | size, qi, h, f, 0))
                buf = np.zeros(self.max_words * 8, dtype=np.uint8)
                buf[: len(match)] = np.frombuffer(match, dtype=np.uint8)
                word_rows.append(buf.view(">u8").astype(np.uint64))
        info = (len(scored), 0, int(cnt.max(initial=0)), len(chunk_rows))
        if not detailed:
            return results, None, None, info
        chunks = np.array(chunk_rows, dtype=SIMPRINT_CHUNK_DTYPE) if chunk_rows else np.zeros(0, dtype=SIMPRINT_CHUNK_DTYPE)
        words = np.stack(word_rows) if word_rows else np.zeros((0, self.max_words), dtype=np.uint64)
        return results, chunks, words, info

    def search_records(self, q_words, q_nbytes, k, max_hamming=None):
        """Structured records [nq, k] + counts, as the device exchange format."""
        if max_hamming is None:
            keys, ham, pbits, cnt = self.search(q_words, q_nbytes, k)
        else:
            keys, ham, pbits, cnt = self.search_within(q_words, q_nbytes, k, max_hamming)
        ranks = rank_table()
        nq = q_words.shape[0]
        rec = np.zeros((nq, k), dtype=RECORD_DTYPE)
        for q in range(nq):
            for i in range(int(cnt[q])):
                if self.key_words == 2:
                    rec[q, i]["key_hi"], rec[q, i]["key_lo"] = keys[q, i, 0], keys[q, i, 1]
                else:
                    rec[q, i]["key_lo"] = keys[q, i]
                h, p = int(ham[q, i]), int(pbits[q, i])
                rec[q, i]["hamming"], rec[q, i]["prefix_bits"] = h, p
                rec[q, i]["dist_rank"] = ranks[(p // 8, h)] if self.metric == 1 else h
        return rec, cnt.astype(np.int32)

    # snapshot interface (same duck type as HipTable; the file logic itself is the product's)
    def segments(self):
        out = {}
        for _, (_, b) in self._rows.items():
            out[b] = out.get(b, 0) + 1
        return dict(sorted(out.items()))

    def _seg_items(self, nbytes):
        return [(k, w) for k, (w, b) in self._rows.items() if b == nbytes]

    def export_rows(self, nbytes, first_row, n):
        items = self._seg_items(nbytes)[first_row : first_row + n]
        W = (nbytes + 7) // 8
        keys = np.zeros((n, 2) if self.key_words == 2 else n, dtype=np.uint64)
        cols = np.zeros((W, n), dtype=np.uint64)
        for i, (k, w) in enumerate(items):
            keys[i] = k if self.key_words == 2 else k[0]
            cols[:, i] = w[:W]
        return keys, cols

    def add_columns(self, nbytes, keys, cols, trusted_unique=False):
        keys = np.asarray(keys, dtype=np.uint64)
        n = keys.shape[0]
        words = np.zeros((n, self.max_words), dtype=np.uint64)
        words[:, : cols.shape[0]] = np.asarray(cols, dtype=np.uint64).T
        self.add(keys, words, np.full(n, nbytes, dtype=np.uint8) if self.metric == 1 else None, trusted_unique=trusted_unique)

    def reserve(self, nbytes, rows):
        pass

    def drop(self):
        self._rows.clear()


def _borrow_snapshot_io():
    from iscc_search_amd.engine import HipTable

    OracleTable.save = HipTable.save
    OracleTable.load = HipTable.load


_borrow_snapshot_io()


class OracleEngine:
    def open_table(self, metric, key_words, max_bytes):
        t = OracleTable(metric, key_words, max_bytes)
        t.engine = self
        return t

    def search_many(self, requests):
        out = []
        for table, q_words, q_nbytes, k, max_hamming in requests:
            q_words = np.asarray(q_words, dtype=np.uint64).reshape(-1, table.max_words)
            out.append(table.search(q_words, q_nbytes, k) if max_hamming is None else table.search_within(q_words, q_nbytes, k, max_hamming))
        return out

    def close(self):
        pass
