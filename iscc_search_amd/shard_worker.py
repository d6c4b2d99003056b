"""
One shard worker of ``hip:///path?devices=N`` (rank 1 .. N-1): started by ``shard_front.LeaderEngine`` as a fresh interpreter.

It holds NO host state of the index -- no assets, no chunk lists, no scoring: it builds a ``ShardedEngine`` over its local
engine and serves the TABLE operations the leader broadcasts (``shard_front.run_table_op``: the very function the leader runs
on its own rank), joining the collectives of each, until told to shut down.  It never returns anything to anybody.  Operations
whose local part can fail on one rank alone end that part with the exchange of outcomes described in ``shard_front``; any
other failure makes the worker exit, which the leader's watchdog notices.
"""

import datetime
import os
import sys
import traceback


def main():
    import numpy as np
    import torch.distributed as dist

    from iscc_search_amd import shard_front as front

    backend = os.environ.get(front.ENV_BACKEND, "nccl")
    same_gpu = os.environ.get(front.ENV_SAME_GPU, "0") == "1"
    timeout_s = float(os.environ.get(front.ENV_TIMEOUT, 300))
    kwargs = {}
    if backend == "nccl":
        import torch

        kwargs["device_id"] = torch.device("cuda", 0 if same_gpu else int(os.environ.get("LOCAL_RANK", os.environ["RANK"])))
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]),
                            timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    engine, ctrl = front.build_rank_engine(os.environ.get(front.ENV_FACTORY), same_gpu, timeout_s)
    channel = front.Channel(dist, ctrl, read_fd=int(os.environ[front.ENV_REQUEST_FD]))
    tables = {}
    code = 0
    trace = os.environ.get("ISCC_HIP_SHARD_TRACE") == "1"      # per-operation wall time of this rank on stderr (tools/probe_leader_search.py)
    import time

    try:
        while True:
            t_wait = time.perf_counter()
            op, table, n, a, b, c, payload = channel.recv()
            t_op = time.perf_counter()
            if op == front.OP_NOP:
                continue
            if op == front.OP_SHUTDOWN:
                break
            error = result = None
            try:
                result = front.run_table_op(engine, tables, op, table, n, a, b, c, payload)
            except BaseException as exc:  # noqa: BLE001
                error = exc
            if op in front.STATUS_OPS:
                same, _ = channel.outcomes_agree(front.error_code(error))
                if not same:
                    raise RuntimeError(f"the ranks disagree about the outcome of table operation {op}") from error
                if error is None and op == front.OP_REMOVE:
                    engine.all_reduce(np.array([result], dtype=np.int64))
            elif error is not None:
                raise error
            if trace:
                print(f"[shard worker {os.environ['RANK']}] op {op} n {n}: waited {(t_op - t_wait) * 1e3:.3f} ms, ran {(time.perf_counter() - t_op) * 1e3:.3f} ms", file=sys.stderr)
    except BaseException:                # noqa: BLE001 -- this shard is gone, and says so by exiting
        traceback.print_exc()
        code = 3
    finally:
        try:
            engine.close()
        except BaseException:            # noqa: BLE001
            pass
    if code:
        os._exit(code)                   # no collective teardown with peers that may be waiting for us
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
