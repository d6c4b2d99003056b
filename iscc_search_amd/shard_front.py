"""
``hip:///path?devices=N`` from ONE calling process: the leader front of a sharded index.

The reference's callers are single processes: the FastAPI lifespan builds one index object
(``iscc_search/server/__init__.py:75-135``), the CLI another (``iscc_search/cli/common.py:41-97``), and the usearch manager
is single-process by contract (``iscc_search/indexes/usearch/manager.py:43-46, :201-217``).  A sharded index nevertheless
needs one process per GPU (RCCL, one HIP context each).  So the process that constructs ``HipIndexManager`` becomes the
LEADER (rank 0) and starts N - 1 shard WORKERS -- fresh interpreters (``python -m iscc_search_amd.shard_worker``), started
before the leader itself touches the GPU.

Division of labour:

* the LEADER alone runs the host logic -- ``HipIndexManager`` / ``HipIndex``: assets, chunk lists, normalisation, aggregation,
  scoring -- over a ``LeaderEngine``, which has the duck type of ``HipEngine`` and hands out ``LeaderTable`` objects;
* the WORKERS hold no host state of the index at all: they serve TABLE operations (open / add / remove / contains / get /
  size / search / search_within / doc_freq / get_freq / save / load / drop) on their shard (``sharded_engine.ShardedHipTable``),
  joining the collectives of each (local top-k -> ONE all-gather -> merge, owner lookups -> one small all-reduce);
* a table operation travels as ONE write per worker on that worker's request PIPE: a 64-byte header {op, table, counts,
  flags} + keys, codes and queries as raw array bytes -- no pickling, no collective, microseconds;
* concurrent callers are COMBINED, as ``isccsearch_search`` combines them on one GPU: the first searching thread leads one
  round, takes every request waiting on the same (table, k, radius), runs them as one request + one collective step and
  hands each caller its slice.  Everything else is serialised by the engine's lock, in the leader's order.

Failure handling:

* arguments are validated on the leader BEFORE anything is broadcast, so a bad request never reaches a worker;
* operations whose local part can fail on one rank alone (open, add, remove, save, load, ...) end their local part with an
  exchange of outcomes (one tiny all-reduce): all ranks fine -> go on; all ranks failed alike -> the leader re-raises to its
  caller and the front stays up; outcomes differ -> the front is torn down.  Read operations (search, lookups) exchange
  nothing: a failure there is a device or transport fault;
* a worker that dies is noticed by the watchdog thread, a leader-side fault marks the front broken: every further call raises
  ``RuntimeError("the sharded index is down: ...")``, the workers are stopped.  No process that has initialised the GPU is
  ever re-executed;
* an IDLE front stays up: a worker waits for its next request in a blocking pipe read, which has no deadline -- the process
  group (whose collectives do carry one) is used only inside an operation; a leader that dies closes the pipes,
  which ends the workers.

The data path keeps its own backend: ``nccl`` (RCCL over xGMI) by default, ``gloo`` for CPU tests and for ranks that share
one GPU (``backend=gloo`` in the URI's query or the constructor).
"""

import datetime
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ENV_FACTORY = "ISCC_HIP_SHARD_ENGINE_FACTORY"     # "module:callable" -> callable(local_rank) = (local engine, ops factory | None, device | None)
ENV_BACKEND = "ISCC_HIP_SHARD_BACKEND"
ENV_SAME_GPU = "ISCC_HIP_SHARD_SAME_GPU"          # "1": every rank uses GPU 0 (rehearsal on a one-GPU box, backend gloo)
ENV_TIMEOUT = "ISCC_HIP_SHARD_TIMEOUT_S"

ENV_REQUEST_FD = "ISCC_HIP_SHARD_REQUEST_FD"      # the read end of this worker's request pipe
# -- the wire format of one table operation ---------------------------------------------------------------------------
HEAD_WORDS = 8                                      # int64: op, table, n, a, b, c, payload bytes, sequence number
(OP_NOP, OP_SHUTDOWN, OP_OPEN, OP_DROP, OP_RESERVE, OP_ADD, OP_REMOVE, OP_CONTAINS, OP_GET, OP_SIZE, OP_SEARCH, OP_DOC_FREQ,
 OP_GET_FREQ, OP_SAVE, OP_LOAD, OP_SET_OPTION, OP_ROWS, OP_SEARCH_MANY) = range(18)
# operations that end their local part with an exchange of outcomes (their local part holds no collective)
STATUS_OPS = frozenset((OP_OPEN, OP_DROP, OP_RESERVE, OP_ADD, OP_REMOVE, OP_SAVE, OP_LOAD, OP_SET_OPTION))
NO_RADIUS = -1
ERROR_CODES = {ValueError: 1, FileNotFoundError: 2, FileExistsError: 3, KeyError: 4, LookupError: 5, MemoryError: 6}


def error_code(exc):
    if exc is None:
        return 0
    for cls, code in ERROR_CODES.items():
        if type(exc) is cls:
            return code
    return 99


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def build_rank_engine(factory_spec, same_gpu, timeout_s):
    """The ``ShardedEngine`` one rank runs over its local engine, and the control group."""
    import importlib

    import torch.distributed as dist

    from iscc_search_amd.sharded_engine import ShardedEngine

    local_rank = 0 if same_gpu else int(os.environ.get("LOCAL_RANK", dist.get_rank()))
    ops_factory = device = None
    if factory_spec:
        module, _, name = factory_spec.partition(":")
        local, ops_factory, device = getattr(importlib.import_module(module), name)(local_rank)
    else:
        from iscc_search_amd.engine import HipEngine   # raises loudly without library / GPU: no CPU fallback

        local, device = HipEngine(local_rank), f"cuda:{local_rank}"
    # host-side control traffic stays off RCCL; its deadline is explicit (the default of a new group is torch's 30 minutes)
    ctrl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s)) if dist.get_backend() != "gloo" else None
    return ShardedEngine(local, ops_factory=ops_factory, device=device, ctrl_group=ctrl), ctrl


def pack(arrays):
    """Raw bytes of some arrays back to back (None = absent)."""
    return b"".join(np.ascontiguousarray(a).tobytes() for a in arrays if a is not None)


def _write_all(fd, data):
    view = memoryview(data)
    while view:
        view = view[os.write(fd, view) :]


def _read_exact(fd, n):
    parts = []
    while n:
        part = os.read(fd, min(n, 1 << 20))
        if not part:
            raise EOFError("the leader closed the request pipe")
        parts.append(part)
        n -= len(part)
    return b"".join(parts)


class Channel:
    """
    Requests from the leader to every worker, and the exchange of outcomes.

    A request is a 64-byte header {op, table, n, a, b, c, payload bytes, sequence number} followed by its payload, written to
    ONE PIPE PER WORKER (created before the workers are started, inherited by them).  A worker waits for its next request in a
    plain blocking read: no deadline runs while the index is idle (the wait used to sit inside a gloo collective,
    whose timeout took an idle index down), a pipe write costs microseconds where a gloo broadcast costs 0.1-0.3 ms, and a
    leader that dies closes the pipe, which ends the worker.  The process group is used only INSIDE an operation.
    """

    def __init__(self, dist, group, write_fds=(), read_fd=None):
        import torch

        self.dist, self.group, self.torch = dist, group, torch
        self.write_fds, self.read_fd = list(write_fds), read_fd
        self.seq = 0

    def send(self, op, table=0, n=0, a=0, b=0, c=0, payload=b""):
        self.seq += 1
        data = np.array([op, table, n, a, b, c, len(payload), self.seq], dtype=np.int64).tobytes() + payload
        for fd in self.write_fds:
            _write_all(fd, data)

    def recv(self):
        op, table, n, a, b, c, nbytes, _ = (int(x) for x in np.frombuffer(_read_exact(self.read_fd, 8 * HEAD_WORDS), dtype=np.int64))
        return op, table, n, a, b, c, (_read_exact(self.read_fd, nbytes) if nbytes else b"")

    def close(self):
        for fd in self.write_fds:
            try:
                os.close(fd)
            except OSError:
                pass
        self.write_fds = []

    def outcomes_agree(self, code):
        """One tiny all-reduce: (every rank reported `code`?, the largest code).  All ranks call it at the same point of an operation."""
        t = self.torch.tensor([code, -code], dtype=self.torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t[0]) == -int(t[1]), int(t[0])


def run_table_op(engine, tables, op, table_id, n, a, b, c, payload):
    """
    The LOCAL part of one table operation on this rank's ``ShardedEngine`` -- the same function on the leader and on every worker.
    Returns the operation's result (the workers drop it).
    """
    if op == OP_OPEN:
        tables[table_id] = engine.open_table(n, a, b)
        return None
    if op == OP_SET_OPTION:
        engine.set_option(payload.decode(), n)
        return None
    if op == OP_SEARCH_MANY:
        requests, off = [], 0
        for _ in range(n):
            tid, nq, k, radius, has_len = (int(x) for x in np.frombuffer(payload, dtype=np.int64, count=5, offset=off))
            off += 40
            t = tables[tid]
            qw = np.frombuffer(payload, dtype=np.uint64, count=nq * t.max_words, offset=off).reshape(nq, t.max_words)
            off += nq * t.max_words * 8
            qn = None
            if has_len:
                qn = np.frombuffer(payload, dtype=np.uint8, count=nq, offset=off)
                off += (nq + 7) // 8 * 8
            requests.append((t, qw, qn, k, None if radius == NO_RADIUS else radius))
        return engine.search_many(requests)       # local searches back to back, ONE all-gather, one merge each
    t = tables[table_id]
    kw, mw = t.key_words, t.max_words
    if op == OP_DROP:
        tables.pop(table_id).drop()
        return None
    if op == OP_RESERVE:
        t.reserve(a, n)
        return None
    if op in (OP_ADD, OP_REMOVE, OP_CONTAINS, OP_GET, OP_GET_FREQ):
        keys = np.frombuffer(payload, dtype=np.uint64, count=n * kw).reshape((n, 2) if kw == 2 else (n,))
        if op == OP_ADD:
            words = np.frombuffer(payload, dtype=np.uint64, count=n * mw, offset=n * kw * 8).reshape(n, mw)
            nbytes = np.frombuffer(payload, dtype=np.uint8, count=n, offset=n * (kw + mw) * 8) if b else None
            t.add(keys, words, nbytes, trusted_unique=bool(a))
            return None
        if op == OP_REMOVE:
            return t.remove_local(keys)
        if op == OP_CONTAINS:
            return t.contains(keys)
        if op == OP_GET:
            return t.get(keys)
        return t.get_freq(keys, a)
    if op == OP_SIZE:
        return t.size
    if op in (OP_SEARCH, OP_DOC_FREQ):
        qw = np.frombuffer(payload, dtype=np.uint64, count=n * mw).reshape(n, mw)
        qn = np.frombuffer(payload, dtype=np.uint8, count=n, offset=n * mw * 8) if c else None
        if op == OP_DOC_FREQ:
            return t.doc_freq(qw, qn, a)
        return t.search(qw, qn, a) if b == NO_RADIUS else t.search_within(qw, qn, a, b)
    if op == OP_SAVE:
        t.save(payload.decode())
        return None
    if op == OP_LOAD:
        t.load(payload.decode())
        return None
    if op == OP_ROWS:
        return t.gathered_rows()
    raise RuntimeError(f"unknown table operation {op}")


class LeaderTable:
    """One logical table of the sharded index, as the leader's host classes see it: the duck type of ``HipTable``."""

    def __init__(self, engine, table_id, metric, key_words, max_bytes):
        self.engine, self.id = engine, table_id
        self.metric, self.key_words, self.max_bytes = metric, key_words, max_bytes
        self.max_words = (max_bytes + 7) // 8

    # -- argument checks (nothing invalid is ever broadcast) ---------------------------------------------------------
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
        from iscc_search_amd._lib import METRIC_HAMMING

        if self.metric == METRIC_HAMMING:
            if nbytes is not None and np.any(np.asarray(nbytes) != self.max_bytes):
                raise ValueError(f"Hamming table holds {self.max_bytes}-byte codes only")
            return None
        if nbytes is None:
            raise ValueError("nbytes is required for NPHD tables")
        nbytes = np.ascontiguousarray(nbytes, dtype=np.uint8)
        if nbytes.shape != (n,):
            raise ValueError("nbytes must be shaped [n]")
        if n and (int(nbytes.min()) < 1 or int(nbytes.max()) > self.max_bytes):
            raise ValueError(f"code length outside 1..{self.max_bytes} bytes")
        return nbytes

    @staticmethod
    def _k(k):
        from iscc_search_amd._lib import MAX_K

        if k < 1:
            raise ValueError("`count` must be >= 1")
        if k > MAX_K:
            raise ValueError(f"count {k} exceeds ISCCSEARCH_MAX_K ({MAX_K})")
        return int(k)

    # -- mutation -----------------------------------------------------------------------------------------------------
    def add(self, keys, words, nbytes=None, trusted_unique=False):
        keys = self._keys(keys)
        n = keys.shape[0]
        if n == 0:
            return
        words = self._words(words, n)
        nbytes = self._nbytes(nbytes, n)
        self.engine.run(OP_ADD, self.id, n, int(bool(trusted_unique)), int(nbytes is not None), payload=pack([keys, words, nbytes]))

    def remove(self, keys):
        keys = self._keys(keys)
        if keys.shape[0] == 0:
            return 0
        return self.engine.run(OP_REMOVE, self.id, keys.shape[0], payload=pack([keys]))

    def reserve(self, nbytes, rows):
        self.engine.run(OP_RESERVE, self.id, int(rows), int(nbytes))

    def drop(self):
        self.engine.run(OP_DROP, self.id)

    # -- lookups ------------------------------------------------------------------------------------------------------
    def contains(self, keys):
        keys = self._keys(keys)
        if keys.shape[0] == 0:
            return np.zeros(0, dtype=bool)
        return self.engine.run(OP_CONTAINS, self.id, keys.shape[0], payload=pack([keys]))

    def get(self, keys):
        keys = self._keys(keys)
        if keys.shape[0] == 0:
            return np.zeros((0, self.max_words), dtype=np.uint64), np.zeros(0, dtype=np.uint8)
        return self.engine.run(OP_GET, self.id, keys.shape[0], payload=pack([keys]))

    @property
    def size(self):
        return self.engine.run(OP_SIZE, self.id)

    def get_freq(self, keys, dup_limit=1000):
        from iscc_search_amd._lib import METRIC_HAMMING

        if self.metric != METRIC_HAMMING:
            raise ValueError("get_freq is defined for fixed-length (Hamming) tables")
        keys = self._keys(keys)
        if keys.shape[0] == 0:
            return np.zeros(0, dtype=np.uint32)
        return self.engine.run(OP_GET_FREQ, self.id, keys.shape[0], self._k(dup_limit), payload=pack([keys]))

    def doc_freq(self, q_words, q_nbytes=None, dup_limit=1000):
        q_words = self._words(q_words)
        nq = q_words.shape[0]
        if nq == 0:
            return np.zeros(0, dtype=np.uint32)
        q_nbytes = self._nbytes(q_nbytes, nq)
        return self.engine.run(OP_DOC_FREQ, self.id, nq, self._k(dup_limit), 0, int(q_nbytes is not None), payload=pack([q_words, q_nbytes]))

    # -- the data path ------------------------------------------------------------------------------------------------
    def search(self, q_words, q_nbytes, k):
        return self._search(q_words, q_nbytes, k, NO_RADIUS)

    def search_within(self, q_words, q_nbytes, k, max_hamming):
        if not 0 <= int(max_hamming) <= 256:
            raise ValueError(f"max_hamming {max_hamming} outside 0..256")
        return self._search(q_words, q_nbytes, k, int(max_hamming))

    def _search(self, q_words, q_nbytes, k, radius):
        q_words = self._words(q_words)
        nq = q_words.shape[0]
        k = self._k(k)
        q_nbytes = self._nbytes(q_nbytes, nq)
        if nq == 0:
            shape = (0, k, 2) if self.key_words == 2 else (0, k)
            return np.zeros(shape, np.uint64), np.zeros((0, k), np.uint32), np.zeros((0, k), np.uint16), np.zeros(0, np.uint32)
        return self.engine.combined_search(self, q_words, q_nbytes, k, radius)

    # -- snapshot -----------------------------------------------------------------------------------------------------
    def save(self, path, chunk_rows=1 << 24):
        self.engine.run(OP_SAVE, self.id, payload=str(path).encode())

    def load(self, path, chunk_rows=1 << 24):
        self.engine.run(OP_LOAD, self.id, payload=str(path).encode())

    def rows(self):
        """(key bytes, code bytes) of every stored row of every shard (restore of snapshots without their host_chunks file)."""
        return iter(self.engine.run(OP_ROWS, self.id))


class _PendingSearch:
    __slots__ = ("table", "q_words", "q_nbytes", "k", "radius", "done", "result", "error")

    def __init__(self, table, q_words, q_nbytes, k, radius):
        self.table, self.q_words, self.q_nbytes, self.k, self.radius = table, q_words, q_nbytes, k, radius
        self.done, self.result, self.error = False, None, None


class LeaderEngine:
    """
    Rank 0 of a sharded index inside the one calling process: the duck type of ``HipEngine`` for the leader's host classes,
    the owner of the shard workers.  One per process (``leader_engine``); managers share it.
    """

    rank = 0

    def __init__(self, devices, backend="nccl", engine_factory=None, same_gpu=False, timeout_s=None):
        # type: (int, str, str | None, bool, float | None) -> None
        import torch.distributed as dist

        if dist.is_initialized():
            raise RuntimeError("LeaderEngine starts its own process group; under torch.distributed.run construct HipIndexManager on every rank instead")
        self.devices = self.world_size = devices
        self.params = (devices, backend, engine_factory, bool(same_gpu))
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get(ENV_TIMEOUT, 300))
        self._lock = threading.RLock()
        self._qcv = threading.Condition()
        self._pending = []
        self._round_active = False
        self._broken = None          # why the front is down
        self._closed = False
        self._users = 0
        self._tables = {}
        self._next_table = 1
        port = free_port()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(devices))
        env[ENV_BACKEND], env[ENV_TIMEOUT] = backend, str(self.timeout_s)
        env[ENV_SAME_GPU] = "1" if same_gpu else "0"
        if engine_factory:
            env[ENV_FACTORY] = engine_factory
        else:
            env.pop(ENV_FACTORY, None)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = os.pathsep.join([root] + [p for p in sys.path if p] + [env.get("PYTHONPATH", "")])
        # fresh interpreters, started BEFORE this process makes its first GPU call (a process that has initialised the GPU must
        # not be the parent of an exec on this platform); each gets its rank through the environment
        self.workers = []
        write_fds = []
        for rank in range(1, devices):
            r, w = os.pipe()
            e = dict(env, RANK=str(rank), LOCAL_RANK=str(0 if same_gpu else rank))
            e[ENV_REQUEST_FD] = str(r)
            self.workers.append(subprocess.Popen([sys.executable, "-m", "iscc_search_amd.shard_worker"], env=e, pass_fds=(r,)))
            os.close(r)
            write_fds.append(w)
        for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE"):
            os.environ[k] = env[k]
        os.environ["RANK"] = "0"
        os.environ.setdefault("LOCAL_RANK", "0")
        try:
            kwargs = {}
            if backend == "nccl":
                import torch

                kwargs["device_id"] = torch.device("cuda", 0)
            dist.init_process_group(backend=backend, rank=0, world_size=devices, timeout=datetime.timedelta(seconds=self.timeout_s), **kwargs)
            self.dist = dist
            self.engine, self.ctrl = build_rank_engine(engine_factory, same_gpu, self.timeout_s)
            self.channel = Channel(dist, self.ctrl, write_fds=write_fds)
        except BaseException:
            for fd in write_fds:
                os.close(fd)
            self._stop_workers()
            raise
        self._watchdog = threading.Thread(target=self._watch, name="hip-shard-watchdog", daemon=True)
        self._watchdog.start()

    # -- failure handling --------------------------------------------------------------------------------------------
    def _watch(self):
        while not self._closed and self._broken is None:
            for rank, proc in enumerate(self.workers, start=1):
                code = proc.poll()
                if code is not None and not self._closed:
                    self._broken = f"shard worker {rank} exited with code {code}"
                    self._stop_workers()
                    return
            time.sleep(0.2)

    def _stop_workers(self):
        for proc in self.workers:
            if proc.poll() is None:
                proc.terminate()
        deadline = time.time() + 10
        for proc in self.workers:
            try:
                proc.wait(max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                proc.kill()

    def check(self):
        if self._broken is not None:
            raise RuntimeError(f"the sharded index is down: {self._broken}")
        if self._closed:
            raise RuntimeError("the sharded index is closed")

    @property
    def broken(self):
        return self._broken

    def _fault(self, exc):
        if self._broken is None:
            self._broken = f"{type(exc).__name__}: {exc}"
        self._stop_workers()
        return RuntimeError(f"the sharded index is down: {self._broken}")

    def _send(self, op, *args, **kw):
        self.channel.send(op, *args, **kw)

    # -- one table operation, in the leader's order -------------------------------------------------------------------
    def run(self, op, table=0, n=0, a=0, b=0, c=0, payload=b""):
        with self._lock:
            self.check()
            try:
                self._send(op, table, n, a, b, c, payload)
            except BaseException as exc:          # noqa: BLE001 -- the frame did not reach everybody: nothing sane can follow
                raise self._fault(exc) from exc
            error = None
            try:
                result = run_table_op(self.engine, self._tables, op, table, n, a, b, c, payload)
            except BaseException as exc:          # noqa: BLE001
                error, result = exc, None
            if op in STATUS_OPS:
                try:
                    same, worst = self.channel.outcomes_agree(error_code(error))
                except BaseException as exc:      # noqa: BLE001
                    raise self._fault(exc) from exc
                if not same:
                    raise self._fault(RuntimeError(f"the ranks disagree about the outcome of table operation {op} "
                                                   f"(here: {type(error).__name__ if error else 'ok'}; worst code {worst})"))
                if error is not None:
                    raise error                   # every rank failed alike: the caller's problem, the front stays up
                if op == OP_REMOVE:               # the local parts are done everywhere: now the count (one small all-reduce)
                    try:
                        result = int(self.engine.all_reduce(np.array([result], dtype=np.int64))[0])
                    except BaseException as exc:  # noqa: BLE001
                        raise self._fault(exc) from exc
            elif error is not None:
                raise self._fault(error) from error
            return result

    # -- HipEngine duck type ------------------------------------------------------------------------------------------
    def open_table(self, metric, key_words, max_bytes):
        from iscc_search_amd._lib import MAX_BYTES, METRIC_HAMMING, METRIC_NPHD

        if metric not in (METRIC_HAMMING, METRIC_NPHD) or key_words not in (1, 2) or not 1 <= max_bytes <= MAX_BYTES:
            raise ValueError("bad table shape")
        with self._lock:
            table_id = self._next_table
            self._next_table += 1
            self.run(OP_OPEN, table_id, metric, key_words, max_bytes)
        return LeaderTable(self, table_id, metric, key_words, max_bytes)

    def set_option(self, name, value):
        self.run(OP_SET_OPTION, 0, int(value), payload=str(name).encode())

    def stats(self, reset=False):
        return self.engine.stats(reset)

    def search_many(self, requests):
        # type: (list[tuple]) -> list[tuple]
        """The per-unit searches of one ``search_assets`` request (``usearch/index.py:786-806``) as ONE frame: the ranks run them back to back."""
        if not requests:
            return []
        parts = []
        for table, q_words, q_nbytes, k, max_hamming in requests:
            q_words = table._words(q_words)
            nq = q_words.shape[0]
            q_nbytes = table._nbytes(q_nbytes, nq)
            radius = NO_RADIUS if max_hamming is None else int(max_hamming)
            parts.append(np.array([table.id, nq, table._k(k), radius, int(q_nbytes is not None)], dtype=np.int64).tobytes())
            parts.append(q_words.tobytes())
            if q_nbytes is not None:
                parts.append(q_nbytes.tobytes() + b"\0" * (-nq % 8))
        return self.run(OP_SEARCH_MANY, 0, len(requests), payload=b"".join(parts))

    def combined_search(self, table, q_words, q_nbytes, k, radius):
        """
        Searches arriving from many threads (the reference calls ``search`` per query unit from FastAPI's thread pool,
        ``usearch/index.py:786-806``): the first caller leads exactly one round, takes every request waiting on the same
        (table, k, radius), runs them as ONE broadcast + ONE collective step and scatters the slices; then hands over.
        """
        me = _PendingSearch(table, q_words, q_nbytes, k, radius)
        with self._qcv:
            self._pending.append(me)
            while True:
                if me.done:
                    break
                if not self._round_active:
                    self._round_active = True
                    break
                self._qcv.wait()
            if me.done:
                if me.error is not None:
                    raise me.error
                return me.result
            batch, self._pending = self._pending, []
        try:
            groups = {}
            for r in batch:
                groups.setdefault((r.table.id, r.k, r.radius), []).append(r)
            for (tid, kk, rad), reqs in groups.items():
                try:
                    t = reqs[0].table
                    qw = reqs[0].q_words if len(reqs) == 1 else np.concatenate([r.q_words for r in reqs])
                    qn = None
                    if reqs[0].q_nbytes is not None:
                        qn = reqs[0].q_nbytes if len(reqs) == 1 else np.concatenate([r.q_nbytes for r in reqs])
                    out = self.run(OP_SEARCH, tid, qw.shape[0], kk, rad, int(qn is not None), payload=pack([qw, qn]))
                    off = 0
                    for r in reqs:
                        m = r.q_words.shape[0]
                        r.result = out if len(reqs) == 1 else tuple(np.ascontiguousarray(x[off : off + m]) for x in out)
                        off += m
                    del t
                except BaseException as exc:      # noqa: BLE001 -- every caller of the group learns why
                    for r in reqs:
                        r.error = exc
        finally:
            with self._qcv:
                for r in batch:
                    r.done = True
                self._round_active = False
                self._qcv.notify_all()
        if me.error is not None:
            raise me.error
        return me.result

    # -- lifetime -----------------------------------------------------------------------------------------------------
    def retain(self):
        with self._lock:
            self._users += 1
        return self

    def close(self):
        """Called by every manager that used the engine; the last one shuts the workers down."""
        global _LEADER
        with self._lock:
            self._users -= 1
            if self._users > 0 or self._closed:
                return
            if self._broken is None:
                try:
                    self._send(OP_SHUTDOWN)
                    self.engine.close()
                except BaseException as exc:      # noqa: BLE001 -- closing must not hang on a dead peer
                    self._broken = f"{type(exc).__name__}: {exc}"
            self.channel.close()                  # (a worker still waiting for a request sees the end of its pipe)
            self._closed = True
            if _LEADER is self:
                _LEADER = None
        deadline = time.time() + 30
        for proc in self.workers:
            try:
                proc.wait(max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                proc.kill()
        try:
            self.dist.destroy_process_group()
        except BaseException:      # noqa: BLE001
            pass


_LEADER = None          # the process's one leader front (a process group is process-wide state)
_LEADER_LOCK = threading.Lock()


def leader_engine(devices, backend="nccl", engine_factory=None, same_gpu=False):
    """
    The leader front of this process, started on first use and shared by every ``HipIndexManager("hip:///...?devices=N")`` of
    the process (a second sharded manager used to fall through to the SPMD path and issue collectives no worker
    joined).  Managers that ask for another shape than the running front's are refused.
    """
    global _LEADER
    with _LEADER_LOCK:
        if _LEADER is not None and not _LEADER._closed:
            if _LEADER.params != (devices, backend, engine_factory, bool(same_gpu)):
                raise ValueError(f"this process already leads a sharded index of {_LEADER.params[0]} ranks ({_LEADER.params[1]}); "
                                 f"a second front of another shape cannot share its process group")
            return _LEADER.retain()
        _LEADER = LeaderEngine(devices, backend=backend, engine_factory=engine_factory, same_gpu=same_gpu)
        return _LEADER.retain()


def leads_this_process():
    """Whether the process group of this process belongs to a leader front (as opposed to a launcher's SPMD group)."""
    return _LEADER is not None and not _LEADER._closed
