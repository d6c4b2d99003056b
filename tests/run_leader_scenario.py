"""
Runs in a process of its own (tests/test_shard_leader.py): ONE process constructs ``HipIndexManager("hip://...?devices=N")``,
which becomes the leader of its shard workers, and drives the index through protocol calls only.

usage: run_leader_scenario.py <uri> <engine factory | -> <out.json> [kill]
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
    else:
        out = protocol_scenario(make)
    with open(out_path, "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
