"""
Runs in a process of its own (tests/test_shard_leader.py): ONE process constructs ``HipIndexManager("hip://...?devices=N")``,
which becomes the leader of its shard workers, and drives the index through protocol calls only.

usage: run_leader_scenario.py <uri> <engine factory | -> <out.json> [kill | idle | threads | outcomes]
"""

import json
import os
import signal
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from helpers import flip_bits, make_asset, sp  # noqa: E402
from iscc_search_amd import codec  # noqa: E402
from iscc_search_amd.index import HipIndexManager  # noqa: E402
from iscc_search_amd.schema import IsccIndex, IsccQuery  # noqa: E402


def protocol_scenario(make):
    """Protocol calls only -- what the reference's server and CLI can do with an index object; returns every answer."""
    rng = np.random.default_rng(11)
    out = []
    dump = lambda r: json.dumps(r.model_dump(mode="json"), sort_keys=True)       # noqa: E731
    s = [rng.integers(0, 256, size=16, dtype=np.uint8).tobytes() for _ in range(4)]
    assets = [make_asset(rng, i, metadata={"source": f"https://example.com/{i}"}) for i in range(60)]
    for i in range(0, 40, 3):
        chunks = [sp(s[i % 4], 0, 10), sp(flip_bits(s[(i + 1) % 4], i % 5), 10, 20), sp(s[0], 30, 5)]
        assets[i] = assets[i].model_copy(update={"simprints": {"CONTENT_TEXT_V0": chunks}})
    manager = make()
    manager.create_index(IsccIndex(name="main"))
    manager.create_index(IsccIndex(name="scratch"))
    try:
        manager.create_index(IsccIndex(name="main"))
    except FileExistsError as e:
        out.append(str(e))
    out.append([r.status.value for r in manager.add_assets("main", assets[:35])])
    out.append([r.status.value for r in manager.add_assets("main", assets[30:])])
    out.append([r.status.value for r in manager.add_assets("scratch", assets[:5])])
    out.append(manager.get_index("main").assets)
    newer = make_asset(rng, 3, simprints={"CONTENT_TEXT_V0": [sp(s[2], 0, 7), sp(s[3], 7, 7)]})
    inst = [u for u in assets[7].units if codec.Iscc(u).unit_type.startswith("INSTANCE_")][0]
    no_inst = assets[7].model_copy(update={"units": [u for u in assets[7].units if u != inst], "iscc_code": None})
    out.append([r.status.value for r in manager.add_assets("main", [newer, no_inst])])
    queries = [
        IsccQuery(iscc_code=assets[10].iscc_code),
        IsccQuery(iscc_code=newer.iscc_code),
        IsccQuery(units=[inst]),
        IsccQuery(units=assets[20].units[:2]),
        IsccQuery(iscc_id=assets[12].iscc_id),
        IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(s[0]), codec.encode_base64(s[1])]}),
        IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(flip_bits(s[2], 2))]}),
    ]
    for q in queries:
        out.append(dump(manager.search_assets("main", q, limit=10)))
    out.append(manager.get_asset("main", assets[3].iscc_id).iscc_code)
    for bad in (lambda: manager.get_asset("main", assets[59].iscc_id.replace("A", "B", 1)), lambda: manager.search_assets("nope", queries[0]),
                lambda: manager.search_assets("main", queries[0], limit=0)):
        try:
            bad()
            out.append("no error")
        except (FileNotFoundError, ValueError) as e:
            out.append(type(e).__name__ + ": " + str(e))
    manager.delete_index("scratch")
    out.append(sorted((i.name, i.assets) for i in manager.list_indexes()))
    manager.flush()
    manager.close()
    manager.close()                      # idempotent
    manager = make()                     # a new leader over the snapshot the first one left
    out.append(sorted((i.name, i.assets) for i in manager.list_indexes()))
    for q in queries:
        out.append(dump(manager.search_assets("main", q, limit=10)))
    out.append([r.status.value for r in manager.add_assets("main", [make_asset(rng, 3)])])
    out.append(dump(manager.search_assets("main", IsccQuery(iscc_code=newer.iscc_code), limit=5)))
    manager.close()
    return out


def main():
    uri, factory, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    factory = None if factory == "-" else factory
    make = lambda: HipIndexManager(uri, shard_engine_factory=factory)      # noqa: E731
    if len(sys.argv) > 4 and sys.argv[4] == "kill":
        m = make()
        m.create_index(IsccIndex(name="main"))
        rng = np.random.default_rng(1)
        m.add_assets("main", [make_asset(rng, i) for i in range(8)])
        os.kill(m._leader.workers[0].pid, signal.SIGKILL)
        time.sleep(1.0)
        t0 = time.time()
        try:
            m.search_assets("main", IsccQuery(iscc_code=make_asset(rng, 99).iscc_code))
            result = "no error"
        except RuntimeError as e:
            result = str(e)
        try:
            m.list_indexes()
            again = "no error"
        except RuntimeError as e:
            again = str(e)
        m.close()
        out = {"first": result, "again": again, "seconds": time.time() - t0}
    elif len(sys.argv) > 4 and sys.argv[4] == "idle":
        # an idle front must outlive the deadline of the collective its workers wait in (ISCC_HIP_SHARD_TIMEOUT_S)
        m = make()
        m.create_index(IsccIndex(name="main"))
        rng = np.random.default_rng(1)
        assets = [make_asset(rng, i) for i in range(8)]
        m.add_assets("main", assets)
        idle = 2.2 * float(os.environ["ISCC_HIP_SHARD_TIMEOUT_S"])
        time.sleep(idle)
        r = m.search_assets("main", IsccQuery(iscc_code=assets[3].iscc_code), limit=3)
        out = {"idle_seconds": idle, "top": r.global_matches[0].iscc_id, "want": assets[3].iscc_id, "score": r.global_matches[0].score}
        m.close()
    elif len(sys.argv) > 4 and sys.argv[4] == "threads":
        # eight caller threads, as FastAPI's thread pool calls the sync protocol methods: every answer must be the single-threaded one
        import threading

        m = make()
        m.create_index(IsccIndex(name="main"))
        rng = np.random.default_rng(5)
        assets = [make_asset(rng, i) for i in range(64)]
        m.add_assets("main", assets)
        queries = [IsccQuery(iscc_code=a.iscc_code) for a in assets]
        dump = lambda r: json.dumps(r.model_dump(mode="json"), sort_keys=True)       # noqa: E731
        want = [dump(m.search_assets("main", q, limit=5)) for q in queries]
        got = [None] * len(queries)
        errors = []

        def worker(tid):
            try:
                for rep in range(3):
                    for i in range(tid, len(queries), 8):
                        got[i] = dump(m.search_assets("main", queries[i], limit=5))
                    if tid == 0 and rep == 1:
                        m.add_assets("main", [assets[0]])             # a writer between the searches (idempotent re-add)
            except BaseException as e:      # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        t0 = time.time()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        out = {"equal": got == want, "errors": errors, "seconds": time.time() - t0, "searches": 3 * len(queries)}
        m.close()
    elif len(sys.argv) > 4 and sys.argv[4] == "outcomes":
        # a failure every rank raises alike goes to the caller and the front stays up; outcomes that differ take it down
        m = make()
        m.create_index(IsccIndex(name="main"))
        rng = np.random.default_rng(1)
        assets = [make_asset(rng, i) for i in range(8)]
        m.add_assets("main", assets)
        eng = m._leader
        t = eng.open_table(0, 1, 8)
        out = {}
        try:
            t.load(os.path.join(os.path.dirname(out_path), "no-such-snapshot"))
            out["alike"] = "no error"
        except ValueError as e:
            out["alike"] = "ValueError: " + str(e)
        out["still_up"] = m.search_assets("main", IsccQuery(iscc_code=assets[2].iscc_code), limit=3).global_matches[0].iscc_id == assets[2].iscc_id
        for bad in (lambda: t.search(np.zeros((1, 2), dtype=np.uint64), None, 3), lambda: t.search(np.zeros((1, 1), dtype=np.uint64), None, 0),
                    lambda: t.add(np.zeros((2, 2), dtype=np.uint64), np.zeros((2, 1), dtype=np.uint64))):
            try:
                bad()
                out.setdefault("invalid", []).append("no error")
            except ValueError as e:
                out.setdefault("invalid", []).append("ValueError")
        out["up_after_invalid"] = len(m.list_indexes()) == 1
        # a snapshot only rank 0 can see: its shard directory exists, the other rank's does not
        lone = os.path.join(os.path.dirname(out_path), "lonely")
        t.add(np.arange(4, dtype=np.uint64), np.arange(4, dtype=np.uint64).reshape(4, 1))
        t.save(lone)
        import shutil

        shutil.rmtree(os.path.join(lone, "shard-1-of-2"))
        t2 = eng.open_table(0, 1, 8)
        try:
            t2.load(lone)
            out["differ"] = "no error"
        except RuntimeError as e:
            out["differ"] = str(e)
        try:
            m.list_indexes()
            out["after"] = "no error"
        except RuntimeError as e:
            out["after"] = str(e)
        m.close()
    else:
        out = protocol_scenario(make)
    with open(out_path, "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
