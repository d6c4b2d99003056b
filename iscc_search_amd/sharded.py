"""
Row-range sharding of one table across the GPUs of a node (one process per GPU).

The reference has no distributed path at all (single process, ``usearch/manager.py:43-46``); this is
the MI355X-native addition SURVEY.md section 8e describes.  Rows are independent and top-k is a
decomposable reduction, so the only exchange is ONE all-gather of the per-shard results
(one block of ``nq * k`` 24-byte records + ``nq`` counts per rank; RCCL over xGMI when the process group's backend
is ``nccl``), followed by a k-way merge on every rank (``merge_kernel``).  No ring all-reduce, no
second collective: at 160 KB per rank for 1 024 queries the exchange is latency bound.

``ShardedTable`` is transport- and engine-agnostic: it drives a small "ops" object.
``HipShardOps`` is the product implementation (device buffers owned by torch, kernels by the
C-ABI); tests drive the same class over gloo with an oracle-backed ops object.
"""

import contextlib
import os

import numpy as np

RECORD_BYTES = 24
COUNT_OVERFLOW = 0xFFFFFFFF   # _lib.COUNT_OVERFLOW (kept here so that this module imports without the HIP library)


def shard_range(n_rows, rank, world_size):
    # type: (int, int, int) -> tuple[int, int]
    """Contiguous row range [lo, hi) of ``rank``: sizes differ by at most one row."""
    lo = n_rows * rank // world_size
    hi = n_rows * (rank + 1) // world_size
    return lo, hi


def shard_of_key(key_lo, world_size):
    # type: (int, int) -> int
    """Owner rank of a key for hash-routed (mutable) tables."""
    x = (int(key_lo) * 0x9E3779B97F4A7C15) & (2**64 - 1)
    return (x >> 32) % world_size


def hint_margin(k):
    # type: (int) -> int
    """Bits above the previous step's worst k-th distance a hinted step starts at: as the engine's own hints (isccsearch.hip, SpecHint::margin)."""
    return 1 if k >= 64 else 2


def block_bytes(nq, k):
    # type: (int, int) -> tuple[int, int]
    """Per-rank exchange block {records [nq][k] | counts [nq] | pad}: (record bytes, block bytes)."""
    rec = nq * k * RECORD_BYTES
    return rec, rec + (nq * 4 + 7) // 8 * 8


def _exchange_scope(ops):
    """The ops' scope for one search step (``HipShardOps.exchange_scope``: torch's current stream := the library's), if it has one."""
    make = getattr(ops, "exchange_scope", None)
    return make() if make is not None else contextlib.nullcontext()


class _OnLibraryStream:
    """``with``: the calling thread's current torch stream on the ops' device is the library's stream; restored on the way out."""

    __slots__ = ("ops", "prev")

    def __init__(self, ops):
        self.ops = ops
        self.prev = None

    def __enter__(self):
        ops = self.ops
        if ops._stream() != ops._lib_stream:
            self.prev = ops._get_current(ops._device_index)           # (stream id, device index, device type)
            ops._set_current(stream_id=ops._ext.stream_id, device_index=ops._ext.device_index, device_type=ops._ext.device_type)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            self.ops._set_current(stream_id=self.prev[0], device_index=self.prev[1], device_type=self.prev[2])
        return False


class HipShardOps:
    """Local search + merge on one GPU through the C-ABI; buffers are torch tensors on that GPU."""

    def __init__(self, table, device):
        import torch

        self.torch = torch
        self.table = table
        self.engine = table.engine
        self.device = torch.device(device)
        self.key_words = table.key_words
        self._buffers = {}
        self._ext = None
        self._lib_stream = self.engine.stream() if hasattr(self.engine, "stream") else 0
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._get_current = getattr(torch._C, "_cuda_getCurrentStream", None)
        self._set_current = getattr(torch._C, "_cuda_setStream", None)
        self._device_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _stream(self):
        # (torch.cuda.current_stream() builds a Stream object: 5.5 us a call, three calls a step; the raw accessor returns the pointer)
        if self._raw_stream is not None:
            return self._raw_stream(self._device_index)
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def exchange_scope(self):
        """
        torch's current stream := the LIBRARY's stream (``isccsearch_stream`` as a ``torch.cuda.ExternalStream``) for one search
        step: local search, all-gather (a collective without ``async_op`` is launched on the current stream) and merge are then
        one in-order queue -- no event between two queues in front of the exchange and in front of the merge (~12 + ~21 us of
        hand-over per step, ``profiles/r04_step_timelines.txt``).  Outside the scope the calls still order themselves by events.
        (``torch.cuda.stream()`` does the same in ~10 us of Stream objects a step; with torch's raw accessors it is ~2.)
        """
        if not self._lib_stream or os.environ.get("ISCC_HIP_SHARD_ONE_QUEUE", "1") == "0":
            return contextlib.nullcontext()              # the step's stages on torch's own current stream, ordered by events
        if self._ext is None:
            self._ext = self.torch.cuda.ExternalStream(self._lib_stream, device=self.device)
        if self._raw_stream is None or self._get_current is None or self._set_current is None:
            return self.torch.cuda.stream(self._ext)
        return _OnLibraryStream(self)

    supports_hint = True

    def local_search(self, q_words, q_nbytes, k, max_hamming=None, synchronous=False, hint=None):
        """
        One device block {records | counts} holding this shard's exact top-k (within ``max_hamming`` if given; with ``hint``: its
        nearest rows within that distance -- fewer than k when the hint was too tight, which ``ShardedTable`` notices after the merge).

        Asynchronous by default: the search is queued on the library's stream and torch's current stream -- the one the
        all-gather is issued on -- waits for it on the device; nothing waits on the host.  ``synchronous`` runs the
        classic path, which also completes queries whose candidate list overflowed (the asynchronous one marks them).
        """
        nq = q_words.shape[0]
        rec_bytes, blk = block_bytes(nq, k)
        buf = self.buffer("block", blk)
        self.table.search_device(q_words, q_nbytes, k, buf.data_ptr(), buf.data_ptr() + rec_bytes, max_hamming=max_hamming,
                                 consumer_stream=None if synchronous else self._stream(), hint=None if synchronous else hint)
        return buf

    def buffer(self, name, nbytes):
        """
        A device buffer of ``nbytes`` kept between steps (one per role and size; a step's buffers are consumed before the next
        step of the same shape starts: the merge synchronises).  Allocating them per step cost ~15 us of host time each.
        """
        key = (name, nbytes)
        buf = self._buffers.get(key)
        if buf is None:
            if len(self._buffers) >= 16:
                self._buffers.clear()
            buf = self._buffers[key] = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
        return buf

    def search_single(self, q_words, q_nbytes, k, max_hamming=None):
        """The whole answer when this shard is the only one: the host entry point (no result block, no merge launch)."""
        if max_hamming is None:
            return self.table.search(q_words, q_nbytes, k)
        return self.table.search_within(q_words, q_nbytes, k, max_hamming)

    def local_doc_freq(self, q_words, q_nbytes, dup_limit):
        """This shard's (distinct assets, colliding rows looked at) per code: lookup and reduction on the device."""
        return self.table.doc_freq_counted(q_words, q_nbytes, dup_limit)

    def merge(self, gathered, n_lists, nq, k):
        """Merge ordered behind torch's current stream (where the gathered blocks were produced): one copy, one synchronisation."""
        rec_bytes, blk = block_bytes(nq, k)
        base = gathered.data_ptr()
        return self.engine.merge_device(n_lists, nq, k, self.key_words, base, base + rec_bytes, blk, blk, after_stream=self._stream())

    def merge_many(self, gathered, stride, parts):
        """Several parts of a fused exchange merged behind ONE synchronisation; parts = [(offset, n_lists, nq, k, key_words)]; None = too large for one call."""
        base = gathered.data_ptr()
        merges = []
        total = 0
        for offset, n_lists, nq, k, key_words in parts:
            rec_bytes, _ = block_bytes(nq, k)
            merges.append((n_lists, nq, k, key_words, base + offset, base + offset + rec_bytes, stride, stride))
            total += rec_bytes + nq * 4 + 16
        if total > (1 << 20):
            return None
        return self.engine.merge_many(merges, self._stream())

    def merge_strided(self, gathered, offset, stride, n_lists, nq, k):
        """The same for a block that is one PART of every rank's share of a fused exchange: list l sits at offset + l x stride."""
        rec_bytes, _ = block_bytes(nq, k)
        base = gathered.data_ptr() + offset
        return self.engine.merge_device(n_lists, nq, k, self.key_words, base, base + rec_bytes, stride, stride, after_stream=self._stream())


class ShardedTable:
    """
    One logical table whose rows are spread over the ranks of a ``torch.distributed`` process group.

    Every rank calls ``search`` with the SAME queries and gets the SAME global top-k back.
    """

    def __init__(self, ops, group=None, always_gather=False, assets_share_a_rank=False):
        import torch.distributed as dist

        self.dist = dist
        self.ops = ops
        self.group = group
        self.always_gather = always_gather   # run the collective even with one rank (rehearsal on one GPU)
        # rows routed by the asset word of their key (sharded_engine.ShardedHipTable): per-shard document frequencies add.
        # Row-RANGE shards (bench.py, the engine-level tests) may split an asset's chunks: the merged list is reduced instead
        self.assets_share_a_rank = assets_share_a_rank
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # (batch-size class, k, query lengths) -> [hint, batches still to skip, penalty]: the GLOBAL k-th distance the previous step of that shape
        # ended at + 2.  Every shard starts its pass under it (no bootstrap sample; the lists hold the shard's rows within the hint),
        # and the step stands if every merged list holds k rows -- they all lie within the hint, so nothing nearer was left out.
        # The merged lists are the same on every rank, hence so is every decision taken from them.
        self.use_hints = bool(getattr(ops, "supports_hint", False)) and not os.environ.get("ISCC_NO_SHARD_HINT")
        self._hints = {}
        self.hint_hits = self.hint_misses = 0
        self._stages_on_host = None

    def _staged(self):
        """The rehearsal transport (several ranks sharing one GPU cannot use RCCL: gloo, blocks staged through the host)?  Asked once."""
        if self._stages_on_host is None:
            self._stages_on_host = self.dist.get_backend(self.group) == "gloo"
        return self._stages_on_host

    def search(self, q_words, q_nbytes, k):
        # type: (np.ndarray, np.ndarray | None, int) -> tuple
        return self._search(q_words, q_nbytes, k, None)

    def search_within(self, q_words, q_nbytes, k, max_hamming):
        # type: (np.ndarray, np.ndarray | None, int, int) -> tuple
        """
        Range-limited search over all shards: each rank lists its rows within ``max_hamming`` (nearest first, at most
        k), the same all-gather + merge keeps the k nearest overall -- the lists are ordered by (distance, key), so
        the merged list equals the unsharded one.
        """
        return self._search(q_words, q_nbytes, k, int(max_hamming))

    def doc_freq(self, q_words, q_nbytes=None, dup_limit=1000):
        # type: (np.ndarray, np.ndarray | None, int) -> np.ndarray
        """
        Distinct assets among the first ``dup_limit`` collisions of each code across ALL shards (``count_doc_freq``,
        ``lmdb_ops.py:139-166``).

        Rows are routed by their ASSET word, so an asset's chunks share a rank and the shards' distinct-asset counts ADD --
        as long as every collision is looked at, i.e. the collisions of all shards together stay within ``dup_limit`` (the
        normal case: each shard reduces its own list on the device, one small all-reduce sums counts and lengths).  A code
        with more collisions than that is cut at ``dup_limit`` in key order ACROSS the shards: for those codes only, the
        merged collision list is reduced (the lists are ordered by key, so distinct assets are the changes of the asset word).
        """
        local = getattr(self.ops, "local_doc_freq", None)
        if self.world_size == 1 and not self.always_gather and local is not None:
            return local(q_words, q_nbytes, dup_limit)[0]
        nq = q_words.shape[0]
        if local is None or not self.assets_share_a_rank:
            slow = np.arange(nq)
            out = np.zeros(nq, dtype=np.uint32)
        else:
            freq, coll = local(q_words, q_nbytes, dup_limit)
            both = self._all_reduce_sum(np.concatenate([freq.astype(np.int64), coll.astype(np.int64)]))
            out = both[:nq].astype(np.uint32)
            slow = np.nonzero(both[nq:] > dup_limit)[0]          # the same on every rank: they all take the slow path together
        if len(slow):
            qn = None if q_nbytes is None else np.ascontiguousarray(np.asarray(q_nbytes)[slow])
            keys, _, _, cnt = self.search_within(np.ascontiguousarray(q_words[slow]), qn, dup_limit, 0)
            assets = keys[..., 0] if keys.ndim == 3 else keys
            valid = np.arange(assets.shape[1])[None, :] < cnt[:, None]
            change = np.ones_like(valid)
            change[:, 1:] = assets[:, 1:] != assets[:, :-1]
            out[slow] = (valid & change).sum(axis=1).astype(np.uint32)
        return out

    def _all_reduce_sum(self, arr):
        # type: (np.ndarray) -> np.ndarray
        """Small host array summed over the ranks (the ops object may bring its own transport: ``ShardedEngine.all_reduce``)."""
        reducer = getattr(self.ops, "all_reduce_sum", None)
        if reducer is not None:
            return reducer(arr)
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64).copy())
        if self.dist.get_backend(self.group) == "nccl":
            t = t.to(getattr(self.ops, "device", "cuda"))
        self.dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    def _search(self, q_words, q_nbytes, k, max_hamming):
        nq = q_words.shape[0]
        alone = self.world_size == 1 and not (self.always_gather and self.dist.is_initialized())
        state = None
        if self.use_hints and max_hamming is None and nq and not alone:
            qlen = None if q_nbytes is None else tuple(sorted(set(int(b) for b in np.asarray(q_nbytes))))      # (a prefix length has its own distances)
            state = self._hints.setdefault((int(nq).bit_length(), int(k), qlen), [None, 0, 0])
            if state[0] is not None and state[1] == 0:
                out = self._exchange(q_words, q_nbytes, k, None, {"hint": state[0]})
                cnt = out[3]
                if not np.any(cnt == COUNT_OVERFLOW) and int(cnt.min()) >= k:
                    worst = int(out[1][:, k - 1].max())
                    state[0], state[2] = max(worst + hint_margin(k), state[0] - 1), 0          # decays by one bit per step towards what the batches need
                    self.hint_hits += 1
                    return out
                self.hint_misses += 1
                state[2] = min(2 * state[2] + 1, 15)                             # a miss: 0, 2, 6, 14 steps without a hint for 1, 2, 3, 4 in a row
                state[1] = state[2] - 1
            elif state[1]:
                state[1] -= 1
        out = self._exchange(q_words, q_nbytes, k, max_hamming, {})
        if np.any(out[3] == COUNT_OVERFLOW):
            # some shard's candidate list overflowed and the asynchronous search could only mark it; the merged counts are
            # the same on every rank, so every rank repeats the step through the synchronous path (exact fallback) together
            out = self._exchange(q_words, q_nbytes, k, max_hamming, {"synchronous": True})
        if state is not None:
            cnt = out[3]
            full = not np.any(cnt == COUNT_OVERFLOW) and int(cnt.min()) >= k
            state[0] = int(out[1][:, k - 1].max()) + hint_margin(k) if full else None       # (a table with fewer than k rows never gets a hint)
        return out

    @staticmethod
    def search_many(items):
        """
        Several searches -- the per-unit searches of one ``search_assets`` request (``usearch/index.py:786-806``), each on its own
        table -- with ONE exchange: every shard runs its local searches back to back, their blocks travel in one all-gather, each
        part is merged on its own.  ``items`` = [(ShardedTable, q_words, q_nbytes, k, max_hamming | None)], queries of one length
        per item; returns the results in order, or None when the items cannot share an exchange (the caller then runs them one
        by one).
        """
        import torch

        first = items[0][0]
        if first.world_size == 1 or len(items) < 2 or len({(id(t.ops), q.shape[0], k) for t, q, _, k, _ in items}) != len(items):
            return None            # (two items on one table with one block size would share that table's block buffer)
        if any(t.world_size != first.world_size or t.group is not first.group for t, *_ in items):
            return None
        with _exchange_scope(first.ops):
            # every item's shards start under the GLOBAL k-th distance its table's previous search of this shape ended at (as `_search`):
            # the per-unit searches of a request then cost each shard one range-limited pass instead of bootstrap + levels
            blocks, states = [], []
            for t, q_words, q_nbytes, k, max_hamming in items:
                state, how = None, {}
                if t.use_hints and max_hamming is None:
                    qlen = None if q_nbytes is None else tuple(sorted(set(int(b) for b in np.asarray(q_nbytes))))
                    state = t._hints.setdefault((int(q_words.shape[0]).bit_length(), int(k), qlen), [None, 0, 0])
                    if state[0] is not None and state[1] == 0:
                        how = {"hint": state[0]}
                    elif state[1]:
                        state[1] -= 1
                states.append((state, bool(how)))
                blocks.append(t.ops.local_search(q_words, q_nbytes, k, **how) if max_hamming is None else t.ops.local_search(q_words, q_nbytes, k, max_hamming))
            share = torch.cat(blocks)
            total = share.numel()
            if share.is_cuda and first._staged():
                host = torch.empty(first.world_size * total, dtype=share.dtype)      # rehearsal transport: staged through the host
                first.dist.all_gather_into_tensor(host, share.cpu(), group=first.group)
                gathered = host.to(share.device)
            else:
                make = getattr(first.ops, "buffer", None)
                gathered = make("gathered", first.world_size * total) if make else torch.empty(first.world_size * total, dtype=share.dtype, device=share.device)
                first.dist.all_gather_into_tensor(gathered, share, group=first.group)
            # the merges: queued back to back behind ONE synchronisation when the ops share an engine that can (HipShardOps.merge_many)
            merged = None
            many = getattr(first.ops, "merge_many", None)
            if many is not None and all(getattr(t.ops, "engine", None) is first.ops.engine for t, *_ in items):
                parts, offset = [], 0
                for (t, q_words, _, k, _), block in zip(items, blocks):
                    parts.append((offset, t.world_size, q_words.shape[0], k, t.ops.key_words))
                    offset += block.numel()
                merged = many(gathered, total, parts)
            out, offset = [], 0
            for (t, q_words, q_nbytes, k, max_hamming), block in zip(items, blocks):
                nq = q_words.shape[0]
                strided = getattr(t.ops, "merge_strided", None)
                if merged is not None:
                    res = merged[len(out)]
                elif strided is not None:
                    res = strided(gathered, offset, total, t.world_size, nq, k)
                else:
                    part = gathered.view(t.world_size, total)[:, offset : offset + block.numel()].contiguous().view(-1)
                    res = t.ops.merge(part, t.world_size, nq, k)
                state, hinted = states[len(out)]
                cnt = res[3]
                full = not np.any(cnt == COUNT_OVERFLOW) and int(cnt.min()) >= k
                if hinted and not full:
                    # the hint was too tight for some query (or a list overflowed under it): the merged lists are the same on every rank, so
                    # every rank repeats THIS item without it together
                    t.hint_misses += 1
                    state[2] = min(2 * state[2] + 1, 15)
                    state[1] = state[2] - 1
                    res = t._exchange(q_words, q_nbytes, k, max_hamming, {})
                    cnt = res[3]
                    hinted = False
                if np.any(cnt == COUNT_OVERFLOW):
                    # a shard could only mark an overflowed candidate list: every rank repeats THIS item through the synchronous path together
                    res = t._exchange(q_words, q_nbytes, k, max_hamming, {"synchronous": True})
                    cnt = res[3]
                if state is not None:
                    full = not np.any(cnt == COUNT_OVERFLOW) and int(cnt.min()) >= k
                    if hinted:
                        t.hint_hits += 1
                        state[0], state[2] = max(int(res[1][:, k - 1].max()) + hint_margin(k), state[0] - 1), 0
                    else:
                        state[0] = int(res[1][:, k - 1].max()) + hint_margin(k) if full else None
                out.append(res)
                offset += block.numel()
        return out

    def _exchange(self, q_words, q_nbytes, k, max_hamming, how):
        nq = q_words.shape[0]
        alone = self.world_size == 1 and not (self.always_gather and self.dist.is_initialized())
        single = getattr(self.ops, "search_single", None)
        if alone and single is not None:
            return single(q_words, q_nbytes, k, max_hamming)
        with _exchange_scope(self.ops):
            block = self.ops.local_search(q_words, q_nbytes, k, **how) if max_hamming is None else self.ops.local_search(q_words, q_nbytes, k, max_hamming, **how)
            if alone:
                return self.ops.merge(block, 1, nq, k)
            import torch

            if block.is_cuda and self._staged():
                # rehearsal transport (several ranks sharing one GPU cannot use RCCL): stage the blocks through the host
                host = torch.empty(self.world_size * block.numel(), dtype=block.dtype)
                self.dist.all_gather_into_tensor(host, block.cpu(), group=self.group)
                gathered = host.to(block.device)
            else:
                make = getattr(self.ops, "buffer", None)
                gathered = make("gathered", self.world_size * block.numel()) if make else torch.empty(self.world_size * block.numel(), dtype=block.dtype, device=block.device)
                # the one exchange step of the path: [world] x {records [nq][k] | counts [nq]}
                self.dist.all_gather_into_tensor(gathered, block, group=self.group)
            return self.ops.merge(gathered, self.world_size, nq, k)
