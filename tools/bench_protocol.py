#!/usr/bin/env python3
"""
BASELINE config 1 through the real backend: random 64-bit units via the full IsccIndexProtocol (HipIndexManager on the GPU):
assets/s for add_assets, ms per search_assets (one thread, and 8 threads), resident memory of the process and of its shard workers.

  python tools/bench_protocol.py [assets, default 2500] [--uri hip:///]              one GPU
  python tools/bench_protocol.py 2500 --uri 'hip:///?devices=2&backend=gloo&same_gpu=1'   the leader front, two ranks on one GPU (rehearsal)

Run both and compare: the difference per search_assets call is the front's overhead, the ratio of the
add_assets rates its ingest cost; the workers' RSS must not grow with the number of assets (they hold no host state).
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd import codec  # noqa: E402
from iscc_search_amd.index import HipIndexManager  # noqa: E402
from iscc_search_amd.schema import IsccEntry, IsccIndex, IsccQuery  # noqa: E402


def rss_mb(pid):
    try:
        with open(f"/proc/{pid}/status") as f:
            for line in f:
                if line.startswith("VmRSS:"):
                    return int(line.split()[1]) / 1024.0
    except OSError:
        pass
    return float("nan")


def make_asset(rng, i):
    units = [codec.encode_unit(mt, 0, 0, rng.integers(0, 256, size=8, dtype=np.uint8).tobytes())
             for mt in (codec.MT_META, codec.MT_CONTENT, codec.MT_DATA, codec.MT_INSTANCE)]
    return IsccEntry(iscc_id=codec.iscc_id_from_int(((1_000_000 + i) << 12) | (i & 0xFFF), 0), iscc_code=codec.gen_iscc_code(units), units=units)


def main():
    args = [a for a in sys.argv[1:]]
    uri = "hip:///"
    if "--uri" in args:
        i = args.index("--uri")
        uri = args[i + 1]
        del args[i : i + 2]
    n_assets = int(args[0]) if args else 2500      # x 4 units = 10 000 codes
    rng = np.random.default_rng(0)
    assets = [make_asset(rng, i) for i in range(n_assets)]
    m = HipIndexManager(uri)
    m.create_index(IsccIndex(name="c1"))
    workers = [p.pid for p in m._leader.workers] if m._leader is not None else []
    print(f"uri {uri}: {len(workers) + 1} rank(s)")
    rss0 = [rss_mb(p) for p in workers]
    t0 = time.perf_counter()
    for i in range(0, n_assets, 500):
        m.add_assets("c1", assets[i : i + 500])
    dt = time.perf_counter() - t0
    print(f"add_assets: {n_assets} assets ({n_assets * 4} units) in {dt:.2f} s = {n_assets / dt:.0f} assets/s")
    queries = [IsccQuery(iscc_code=a.iscc_code) for a in assets[:200]]
    one_unit = [IsccQuery(units=[a.units[0]]) for a in assets[:200]]
    for name, qs in (("search_assets (4 units per query, limit 10)", queries), ("search_assets (1 unit per query, limit 10)", one_unit)):
        m.search_assets("c1", qs[0], limit=10)
        t0 = time.perf_counter()
        for q in qs:
            r = m.search_assets("c1", q, limit=10)
            assert r.global_matches[0].score == 1.0
        dt = time.perf_counter() - t0
        print(f"{name}, 1 thread: {len(qs) / dt:.0f} searches/s ({dt / len(qs) * 1e3:.3f} ms each)")

    def worker(chunk):
        for q in chunk:
            m.search_assets("c1", q, limit=10)

    threads = [threading.Thread(target=worker, args=(queries[i::8],)) for i in range(8)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    print(f"search_assets (4 units), 8 threads: {len(queries) / dt:.0f} searches/s")
    print(f"resident memory: this process {rss_mb(os.getpid()):.0f} MB; shard workers before / after ingest: "
          f"{', '.join(f'{a:.0f} / {rss_mb(p):.0f} MB' for a, p in zip(rss0, workers)) or '-'}")
    m.close()


if __name__ == "__main__":
    main()
