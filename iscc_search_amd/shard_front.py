"""
``hip:///path?devices=N`` from ONE calling process.

The reference's callers are single processes: the FastAPI lifespan builds one index object
(``iscc_search/server/__init__.py:75-135``), the CLI another (``iscc_search/cli/common.py:41-97``), and the usearch manager
is single-process by contract (``iscc_search/indexes/usearch/manager.py:43-46, :201-217``).  A sharded index nevertheless
needs one process per GPU (RCCL, one HIP context each).  So the process that constructs ``HipIndexManager`` becomes the
LEADER (rank 0) and starts N - 1 shard WORKERS -- fresh interpreters (``python -m iscc_search_amd.shard_worker``), started
before the leader itself touches the GPU -- and

* every protocol call is SEQUENCED by the leader: one lock, one ``broadcast_object_list`` of ``(method, args, kwargs)`` on
  a gloo control group, then every rank runs that call on its own ``HipIndexManager`` over a ``ShardedEngine`` (rows routed
  by key hash, local top-k + one all-gather + merge: ``sharded_engine.py``).  All ranks therefore make the same calls in
  the same order whatever the caller's threads do, and only the leader returns anything;
* errors that every rank raises alike (invalid input, unknown index / asset: raised before any collective) travel to the
  caller from the leader's own call; a worker that fails differently EXITS, and a watchdog thread in the leader notices a
  worker that is gone: the front is marked broken (every further call raises), the other workers are stopped, the process
  group -- whose collectives carry a timeout -- is torn down.  No process that has initialised the GPU is ever re-executed.

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

ENV_FACTORY = "ISCC_HIP_SHARD_ENGINE_FACTORY"     # "module:callable" -> callable(local_rank) = (local engine, ops factory | None, device | None)
ENV_URI = "ISCC_HIP_SHARD_URI"
ENV_BACKEND = "ISCC_HIP_SHARD_BACKEND"
ENV_SAME_GPU = "ISCC_HIP_SHARD_SAME_GPU"          # "1": every rank uses GPU 0 (rehearsal on a one-GPU box, backend gloo)
ENV_TIMEOUT = "ISCC_HIP_SHARD_TIMEOUT_S"
SHUTDOWN = "__shutdown__"
# raised alike on every rank, before any collective: the leader's own call reports them
DETERMINISTIC = (ValueError, FileNotFoundError, FileExistsError)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def build_rank_manager(uri, factory_spec, same_gpu):
    """The ``HipIndexManager`` one rank runs: the SPMD manager of ``sharded_engine.py`` over this rank's local engine."""
    import importlib

    import torch.distributed as dist

    from iscc_search_amd.index import HipIndexManager
    from iscc_search_amd.sharded_engine import ShardedEngine

    local_rank = 0 if same_gpu else int(os.environ.get("LOCAL_RANK", dist.get_rank()))
    ops_factory = device = None
    if factory_spec:
        module, _, name = factory_spec.partition(":")
        local, ops_factory, device = getattr(importlib.import_module(module), name)(local_rank)
    else:
        from iscc_search_amd.engine import HipEngine   # raises loudly without library / GPU: no CPU fallback

        local, device = HipEngine(local_rank), f"cuda:{local_rank}"
    ctrl = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None      # host-side control traffic stays off RCCL
    engine = ShardedEngine(local, ops_factory=ops_factory, device=device, ctrl_group=ctrl)
    manager = HipIndexManager(uri, engine=engine)
    manager._owns_engine = True
    return manager, ctrl


class ShardLeader:
    """Rank 0 of a sharded index inside the one calling process; owns the workers."""

    def __init__(self, uri, devices, backend="nccl", engine_factory=None, same_gpu=False, timeout_s=None):
        # type: (str, int, str, str | None, bool, float | None) -> None
        import torch.distributed as dist

        if dist.is_initialized():
            raise RuntimeError("ShardLeader starts its own process group; under torch.distributed.run construct HipIndexManager on every rank instead")
        self.devices = devices
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get(ENV_TIMEOUT, 300))
        self._lock = threading.Lock()
        self._broken = None          # why the front is down
        self._closed = False
        port = free_port()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(devices))
        env[ENV_URI], env[ENV_BACKEND], env[ENV_TIMEOUT] = uri, backend, str(self.timeout_s)
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
        for rank in range(1, devices):
            e = dict(env, RANK=str(rank), LOCAL_RANK=str(0 if same_gpu else rank))
            self.workers.append(subprocess.Popen([sys.executable, "-m", "iscc_search_amd.shard_worker"], env=e))
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
            self.manager, self.ctrl = build_rank_manager(uri, engine_factory, same_gpu)
        except BaseException:
            self._stop_workers()
            raise
        self._watchdog = threading.Thread(target=self._watch, name="hip-shard-watchdog", daemon=True)
        self._watchdog.start()

    # -- failure handling ---------------------------------------------------------------------------------------------
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

    def _check(self):
        if self._broken is not None:
            raise RuntimeError(f"the sharded index is down: {self._broken}")
        if self._closed:
            raise RuntimeError("the sharded index is closed")

    # -- the one entry point --------------------------------------------------------------------------------------------
    def call(self, method, *args, **kwargs):
        with self._lock:           # the leader sequences: one request at a time, the same order on every rank
            self._check()
            try:
                self.dist.broadcast_object_list([(method, args, kwargs)], src=0, group=self.ctrl)
                return getattr(self.manager, method)(*args, **kwargs)
            except DETERMINISTIC:
                raise
            except BaseException as exc:
                # a collective that lost its peer, a device error on this rank: nothing sane can follow on this group
                if self._broken is None:
                    self._broken = f"{type(exc).__name__}: {exc}"
                self._stop_workers()
                raise RuntimeError(f"the sharded index is down: {self._broken}") from exc

    def close(self):
        with self._lock:
            if self._closed:
                return
            if self._broken is None:
                try:
                    self.dist.broadcast_object_list([("close", (), {})], src=0, group=self.ctrl)
                    self.manager.close()
                    self.dist.broadcast_object_list([(SHUTDOWN, (), {})], src=0, group=self.ctrl)
                except BaseException as exc:      # noqa: BLE001 -- closing must not hang on a dead peer
                    self._broken = f"{type(exc).__name__}: {exc}"
            self._closed = True
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
