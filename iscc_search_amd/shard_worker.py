"""
One shard worker of ``hip:///path?devices=N`` (rank 1 .. N-1): started by ``shard_front.ShardLeader`` as a fresh interpreter,
it builds the same ``HipIndexManager`` over a ``ShardedEngine`` as the leader and replays the protocol calls the leader
broadcasts -- joining their collectives -- until told to shut down.  It never returns anything to anybody; if a call fails
in a way the leader's own call does not (anything but invalid input / unknown index), it exits, which the leader notices.
"""

import datetime
import os
import sys
import traceback


def main():
    import torch.distributed as dist

    from iscc_search_amd import shard_front as front

    backend = os.environ.get(front.ENV_BACKEND, "nccl")
    same_gpu = os.environ.get(front.ENV_SAME_GPU, "0") == "1"
    timeout = datetime.timedelta(seconds=float(os.environ.get(front.ENV_TIMEOUT, 300)))
    kwargs = {}
    if backend == "nccl":
        import torch

        kwargs["device_id"] = torch.device("cuda", 0 if same_gpu else int(os.environ.get("LOCAL_RANK", os.environ["RANK"])))
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), timeout=timeout, **kwargs)
    manager, ctrl = front.build_rank_manager(os.environ[front.ENV_URI], os.environ.get(front.ENV_FACTORY), same_gpu)
    code = 0
    try:
        while True:
            msg = [None]
            dist.broadcast_object_list(msg, src=0, group=ctrl)
            method, args, kwargs_ = msg[0]
            if method == front.SHUTDOWN:
                break
            try:
                getattr(manager, method)(*args, **kwargs_)
            except front.DETERMINISTIC:
                pass                     # the leader raised the same to its caller
    except BaseException:                # noqa: BLE001 -- anything else: this shard is gone, and says so by exiting
        traceback.print_exc()
        code = 3
    finally:
        try:
            manager.close()
        except BaseException:            # noqa: BLE001
            pass
    if code:
        os._exit(code)                   # no collective teardown with peers that may be waiting for us
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
