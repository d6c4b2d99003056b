"""
``hip:///?devices=N``: the index sharded over the N GPUs of a node behind the SAME host classes.

One process per GPU (``python -m torch.distributed.run --nproc-per-node N ...``), every rank constructs the same
``HipIndexManager`` and calls the protocol methods with the same arguments (SPMD, like every ``torch.distributed``
program); every rank returns the same answer.  The reference has nothing to compare with -- it is single-process by
contract (``iscc_search/indexes/usearch/manager.py:43-46``); what this replaces is the one ``ShardedNphdIndex`` /
``ShardedIndex128`` object per table of ``usearch/index.py:1617-1625`` and ``simprint/usearch_core.py:73-83``.

``ShardedEngine`` has the duck type of ``HipEngine`` and hands out ``ShardedHipTable`` objects with the duck type of
``HipTable``, so ``HipNphdIndex``, ``HipIndex128``, ``HipSimprintIndex`` and ``HipIndex`` run unchanged on top:

* rows are ROUTED by key hash (``sharded.shard_of_key``; 128-bit chunk keys by their asset word, so one asset's chunks
  share a rank): ``add`` keeps the rows this rank owns, ``remove`` removes them;
* ``search`` / ``search_within`` / ``doc_freq`` are the data path: local exact top-k per shard, ONE all-gather of
  ``{records | counts}`` blocks, k-way merge on every rank (``sharded.ShardedTable``);
* ``contains`` / ``get`` / ``size`` / ``remove``'s count are answered by the owner and combined with one small
  all-reduce each -- the key -> rank map is the hash itself, no host table is needed.

SPMD discipline (as for any ``torch.distributed`` program): every rank must make the same calls in the same order.  Two ways
to get that:

* ONE calling process (the reference's server and CLI): ``HipIndexManager("hip:///path?devices=N")`` constructed where no
  process group exists becomes the leader (``shard_front.LeaderEngine``): it starts the other ranks itself, runs the host
  logic alone and broadcasts every TABLE operation to the workers, which serve them on their ``ShardedHipTable`` -- the order
  is the leader's, whatever the caller's threads do;
* ``python -m torch.distributed.run --nproc-per-node N script.py`` where the script itself makes identical calls on every rank
  (``bench.py``): then each rank's manager holds one lock around EVERY protocol method (also flush, close, the lazy snapshot
  load and delete_index: all of them reach collectives), which keeps one rank's threads from interleaving two requests'
  collectives -- it cannot order two threads the same way on two ranks, so such a script must issue its calls from one thread.

An exception raised on ONE rank only (a device allocation failing on one GPU) leaves the others waiting in their collective
until the group's timeout: the leader's watchdog (first way) or the launcher (second way) tears the job down.
"""

import os

import numpy as np

from iscc_search_amd.sharded import ShardedTable

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def owner_of(route_words, world_size):
    # type: (np.ndarray, int) -> np.ndarray
    """Vectorised ``sharded.shard_of_key``: owner rank of every key."""
    with np.errstate(over="ignore"):
        x = np.asarray(route_words, dtype=np.uint64) * _GOLDEN
    return ((x >> np.uint64(32)) % np.uint64(world_size)).astype(np.int64)


class ShardedEngine:
    """Engine facade over one local engine per rank (``HipEngine`` in production)."""

    def __init__(self, local_engine, ops_factory=None, group=None, device=None, ctrl_group=None):
        # type: (object, object | None, object | None, object | None, object | None) -> None
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("a sharded index needs an initialised torch.distributed process group (one process per GPU)")
        self.dist = dist
        self.local = local_engine
        self.group = group
        # small host-side collectives (owner lookups, counts, snapshot barriers) go over `ctrl_group` when given -- a gloo group
        # beside an RCCL data group: no staging through the GPU, no RCCL launch for 8 bytes
        self.ctrl_group = ctrl_group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        if ops_factory is None:
            from iscc_search_amd.sharded import HipShardOps

            ops_factory = lambda table: HipShardOps(table, self.device if self.device is not None else "cuda:%d" % local_engine.device_id)  # noqa: E731
        self._ops_factory = ops_factory

    # -- small control-path collectives -----------------------------------------------------------
    def all_reduce(self, arr, op="sum"):
        # type: (np.ndarray, str) -> np.ndarray
        """Element-wise reduction of a small host array over the ranks (staged on the GPU when the backend is RCCL)."""
        import torch

        a = np.ascontiguousarray(arr)
        view = a.view(np.int64) if a.dtype == np.uint64 else a
        t = torch.from_numpy(view.copy())
        group = self.ctrl_group if self.ctrl_group is not None else self.group
        if self.dist.get_backend(group) == "nccl":
            t = t.to(self.device if self.device is not None else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX, group=group)
        out = t.cpu().numpy()
        return out.view(np.uint64) if a.dtype == np.uint64 else out.astype(a.dtype, copy=False)

    def all_gather_object(self, obj):
        out = [None] * self.world_size
        self.dist.all_gather_object(out, obj, group=self.ctrl_group if self.ctrl_group is not None else self.group)
        return out

    # -- HipEngine duck type ---------------------------------------------------------------------------
    def open_table(self, metric, key_words, max_bytes):
        return ShardedHipTable(self, self.local.open_table(metric, key_words, max_bytes))

    def set_option(self, name, value):
        self.local.set_option(name, value)

    def stats(self, reset=False):
        return self.local.stats(reset)

    def search_many(self, requests):
        # type: (list[tuple]) -> list[tuple]
        """
        The per-unit searches of one ``search_assets`` request: local searches back to back, ONE all-gather for all of them, one
        merge each (``ShardedTable.search_many``); requests that cannot share an exchange (queries of several lengths in one
        request, two requests on one table) run one by one.
        """
        items = []
        for table, q_words, q_nbytes, k, max_hamming in requests:
            if k < 1:
                raise ValueError("`count` must be >= 1")
            q_words = np.ascontiguousarray(q_words, dtype=np.uint64).reshape(-1, table.max_words)
            q_nbytes = None if q_nbytes is None else np.ascontiguousarray(q_nbytes, dtype=np.uint8)
            items.append((table._sharded, q_words, q_nbytes, k, None if max_hamming is None else int(max_hamming)))
        if all(qn is None or len(np.unique(qn)) <= 1 for _, _, qn, _, _ in items) and all(q.shape[0] for _, q, _, _, _ in items):
            fused = ShardedTable.search_many(items)
            if fused is not None:
                return fused
        out = []
        for table, q_words, q_nbytes, k, max_hamming in requests:
            out.append(table.search(q_words, q_nbytes, k) if max_hamming is None else table.search_within(q_words, q_nbytes, k, max_hamming))
        return out

    def close(self):
        self.local.close()


class ShardedHipTable:
    """One logical table; this rank holds the rows whose key hashes to it."""

    def __init__(self, engine, local_table):
        # type: (ShardedEngine, object) -> None
        self.engine = engine
        self.local = local_table
        self.metric = local_table.metric
        self.key_words = local_table.key_words
        self.max_bytes = local_table.max_bytes
        self.max_words = local_table.max_words
        ops = engine._ops_factory(local_table)
        if not hasattr(ops, "all_reduce_sum"):
            ops.all_reduce_sum = lambda arr: engine.all_reduce(np.ascontiguousarray(arr, dtype=np.int64))      # the engine's control transport
        self._sharded = ShardedTable(ops, group=engine.group, assets_share_a_rank=True)

    # -- routing -----------------------------------------------------------------------------------------
    def _mine(self, keys):
        keys = np.asarray(keys, dtype=np.uint64)
        route = keys[:, 0] if self.key_words == 2 else keys      # chunk keys: the asset word, so an asset's chunks share a rank
        return owner_of(route, self.engine.world_size) == self.engine.rank

    # -- mutation ------------------------------------------------------------------------------------------
    def add(self, keys, words, nbytes=None, trusted_unique=False):
        keys = np.asarray(keys, dtype=np.uint64)
        if keys.shape[0] == 0:
            return
        mine = self._mine(keys)
        if mine.any():
            self.local.add(keys[mine], np.asarray(words, dtype=np.uint64)[mine], None if nbytes is None else np.asarray(nbytes)[mine],
                           trusted_unique=trusted_unique)

    def remove_local(self, keys):
        """This rank's part of ``remove`` (no collective): how many of the keys it held and removed."""
        keys = np.asarray(keys, dtype=np.uint64)
        removed = 0
        if keys.shape[0]:
            mine = self._mine(keys)
            if mine.any():
                removed = self.local.remove(keys[mine])
        return removed

    def remove(self, keys):
        return int(self.engine.all_reduce(np.array([self.remove_local(keys)], dtype=np.int64))[0])

    # -- lookups answered by the owner -----------------------------------------------------------------------
    def contains(self, keys):
        keys = np.asarray(keys, dtype=np.uint64)
        if keys.shape[0] == 0:
            return np.zeros(0, dtype=bool)
        return self.engine.all_reduce(self.local.contains(keys).astype(np.uint8), op="max").astype(bool)

    def get(self, keys):
        keys = np.asarray(keys, dtype=np.uint64)
        if keys.shape[0] == 0:
            return np.zeros((0, self.max_words), dtype=np.uint64), np.zeros(0, dtype=np.uint8)
        words, nb = self.local.get(keys)       # zero where this rank does not hold the key: the sum is the owner's row
        return self.engine.all_reduce(words), self.engine.all_reduce(nb.astype(np.int64)).astype(np.uint8)

    @property
    def size(self):
        return int(self.engine.all_reduce(np.array([self.local.size], dtype=np.int64))[0])

    # -- the data path -----------------------------------------------------------------------------------------
    def _by_length(self, q_words, q_nbytes, run):
        """The device search takes queries of ONE byte length per call (one compared prefix per segment)."""
        q_words = np.ascontiguousarray(q_words, dtype=np.uint64).reshape(-1, self.max_words)
        if q_nbytes is None or len(np.unique(q_nbytes)) <= 1:
            return run(q_words, None if q_nbytes is None else np.ascontiguousarray(q_nbytes, dtype=np.uint8))
        q_nbytes = np.asarray(q_nbytes, dtype=np.uint8)
        out = None
        for length in np.unique(q_nbytes):
            sel = np.nonzero(q_nbytes == length)[0]
            part = run(np.ascontiguousarray(q_words[sel]), np.ascontiguousarray(q_nbytes[sel]))
            if out is None:
                out = tuple(np.zeros((q_words.shape[0],) + p.shape[1:], dtype=p.dtype) for p in part)
            for o, p in zip(out, part):
                o[sel] = p
        return out

    def search(self, q_words, q_nbytes, k):
        if k < 1:
            raise ValueError("`count` must be >= 1")
        return self._by_length(q_words, q_nbytes, lambda qw, qn: self._sharded.search(qw, qn, k))

    def search_within(self, q_words, q_nbytes, k, max_hamming):
        if k < 1:
            raise ValueError("`count` must be >= 1")
        return self._by_length(q_words, q_nbytes, lambda qw, qn: self._sharded.search_within(qw, qn, k, int(max_hamming)))

    def doc_freq(self, q_words, q_nbytes=None, dup_limit=1000):
        q_words = np.ascontiguousarray(q_words, dtype=np.uint64).reshape(-1, self.max_words)
        if q_words.shape[0] == 0:
            return np.zeros(0, dtype=np.uint32)
        return self._sharded.doc_freq(q_words, q_nbytes, dup_limit)

    def get_freq(self, keys, dup_limit=1000):
        """
        Document frequency of the code stored under each key (0 for absent keys).  The owner's frequency COLUMN cannot answer
        it: the other shards hold rows with the same code under other keys.  So the owner supplies the code (one all-reduce) and
        every shard counts it on the device, batched (``doc_freq``: the counts add, see there).
        """
        words, nb = self.get(keys)
        out = np.zeros(len(nb), dtype=np.uint32)
        present = np.nonzero(nb)[0]
        if len(present):
            out[present] = self.doc_freq(words[present], None, dup_limit)
        return out

    # -- snapshot: every rank keeps its own shard ------------------------------------------------------------------
    def segments(self):
        return self.local.segments()

    def export_rows(self, nbytes, first_row, n):
        return self.local.export_rows(nbytes, first_row, n)

    def gathered_rows(self):
        """Sorted (key bytes, code bytes) of the rows of EVERY shard (a collective; restore of old snapshots only)."""
        from iscc_search_amd.nphd import table_rows

        return sorted(r for part in self.engine.all_gather_object(list(table_rows(self.local))) for r in part)

    def add_columns(self, nbytes, keys, cols, trusted_unique=False):
        self.local.add_columns(nbytes, keys, cols, trusted_unique=trusted_unique)

    def reserve(self, nbytes, rows):
        self.local.reserve(nbytes, rows)

    def _shard_dir(self, path):
        return os.path.join(path, "shard-%d-of-%d" % (self.engine.rank, self.engine.world_size))

    def save(self, path, chunk_rows=1 << 24):
        self.local.save(self._shard_dir(path), chunk_rows)

    def load(self, path, chunk_rows=1 << 24):
        if not os.path.isdir(self._shard_dir(path)):
            raise ValueError(f"snapshot at {path} was not written by {self.engine.world_size} ranks")
        self.local.load(self._shard_dir(path), chunk_rows)

    def drop(self):
        self.local.drop()
