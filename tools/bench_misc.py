#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench): latency, small tables, large k, mixed lengths.

  python tools/bench_misc.py [latency] [small] [simprint] [mixed] [threads] [within]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iscc_search_amd import _lib  # noqa: E402
from iscc_search_amd.engine import HipEngine  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    what = set(sys.argv[1:]) or {"latency", "small", "simprint", "mixed", "threads", "within"}
    eng = HipEngine(0)
    for item in filter(None, os.environ.get("ISCC_HIP_OPTS", "").split(",")):      # e.g. ISCC_HIP_OPTS=mfma=0
        eng.set_option(item.split("=")[0].strip(), int(item.split("=")[1]))
    rng = np.random.default_rng(0)
    if "latency" in what:
        t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
        t.add_synthetic(8, 100_000_000, 1)
        for nq in (1, 8, 16, 64):
            q = rng.integers(0, 2**64, size=(nq, 1), dtype=np.uint64)
            dt = timeit(lambda: t.search(q, None, 10))
            print(f"latency: 100M x 64-bit, nq={nq:3d}, k=10: {dt*1e3:8.3f} ms/call  ({nq/dt:9.0f} qps)")
        for k in (100, 1000, 4000):
            q = rng.integers(0, 2**64, size=(16, 1), dtype=np.uint64)
            dt = timeit(lambda: t.search(q, None, k), reps=5, warm=1)
            print(f"large k: 100M x 64-bit, nq=16, k={k}: {dt*1e3:8.3f} ms/call")
        t.drop()
    if "small" in what:
        # config 2 cold pass: evict the 8 MB table from L2 / Infinity Cache by streaming a 2.4 GB table, then time
        # ONE search (queries_per_pass 16 -> a single pass for 16 queries) before the warm loop below
        flush = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
        flush.add_synthetic(8, 300_000_000, 99)
        t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
        t.add_synthetic(8, 1_000_000, 2)
        q16 = rng.integers(0, 2**64, size=(16, 1), dtype=np.uint64)
        t.search(q16, None, 10)                       # buffers sized, kernels loaded
        cold = []
        for _ in range(5):
            flush.search(q16, None, 10)
            t0 = time.perf_counter()
            t.search(q16, None, 10)
            cold.append(time.perf_counter() - t0)
        warm = timeit(lambda: t.search(q16, None, 10), reps=20)
        print(f"config 2 cold: 1M x 64-bit, nq=16 after streaming 2.4 GB: {min(cold)*1e3:.3f} ms (median {sorted(cold)[2]*1e3:.3f}); warm {warm*1e3:.3f} ms")
        t.drop()
        flush.drop()
        for n in (10_000, 1_000_000, 10_000_000):
            t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
            t.add_synthetic(8, n, 2)
            for nq in (1, 16, 1024):
                q = rng.integers(0, 2**64, size=(nq, 1), dtype=np.uint64)
                dt = timeit(lambda: t.search(q, None, 10))
                print(f"small: {n:>10d} x 64-bit, nq={nq:4d}, k=10: {dt*1e3:8.3f} ms/call  ({nq/dt:10.0f} qps, {nq/8*n*8/dt/1e9:7.0f} GB/s algorithmic at T_q=8)")
            t.drop()
    if "simprint" in what:
        # config 5: 10 M segments in three tables by ndim (64/128/256), 128-bit keys, count = 40 x limit
        for nbytes, n in ((8, 4_000_000), (16, 4_000_000), (32, 2_000_000)):
            t = eng.open_table(_lib.METRIC_HAMMING, 2, nbytes)
            t.add_synthetic(nbytes, n, 3 + nbytes)
            for nq, k in ((64, 400), (512, 400), (64, 4000)):
                q = rng.integers(0, 2**64, size=(nq, t.max_words), dtype=np.uint64)
                dt = timeit(lambda: t.search(q, None, k), reps=3, warm=1)
                print(f"simprint: {n} x {nbytes*8}-bit (128-bit keys), nq={nq}, k={k}: {dt*1e3:9.3f} ms/call ({nq/dt:8.0f} qps)")
            t.drop()
    if "within" in what:
        # range-limited search (collision lookup) and the document-frequency column
        t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
        t.add_synthetic(8, 100_000_000, 1)
        _, cols = t.export_rows(8, 12345, 1024)
        stored = cols.T.copy()
        for nq in (1, 16, 1024):
            q = stored[:nq]
            for r, k in ((0, 1000), (4, 1000)):
                dt = timeit(lambda: t.search_within(q, None, k, r), reps=5, warm=1)
                print(f"within: 100M x 64-bit, nq={nq:4d}, max_hamming={r}, k={k}: {dt*1e3:8.3f} ms/call ({nq/dt:9.0f} qps)")
            dt = timeit(lambda: t.search(q, None, 10), reps=5, warm=1)
            print(f"        same queries, plain top-10 search:              {dt*1e3:8.3f} ms/call")
        t.drop()
        for nbytes, n in ((16, 10_000_000), (32, 10_000_000), (8, 50_000_000)):
            t = eng.open_table(_lib.METRIC_HAMMING, 2, nbytes)
            t.add_synthetic(nbytes, n, 7)
            probe = np.array([[0, 5]], dtype=np.uint64)
            t0 = time.perf_counter()
            t.get_freq(probe)                       # host key index + first column build
            first = time.perf_counter() - t0
            extra = np.array([[1, 1]], dtype=np.uint64)
            w = np.zeros((1, t.max_words), dtype=np.uint64)
            times = []
            for i in range(3):
                t.remove(extra) if i else None
                t.add(extra, w)                     # rows changed -> the column is rebuilt by the next lookup
                t0 = time.perf_counter()
                t.get_freq(probe)
                times.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            for _ in range(20):
                t.get_freq(probe)
            look = (time.perf_counter() - t0) / 20
            print(f"freq column: {n} x {nbytes*8}-bit rows (128-bit keys): first call {first*1e3:8.1f} ms (incl. host key index), "
                  f"rebuild {min(times)*1e3:7.2f} ms, lookup {look*1e3:6.3f} ms")
            t.drop()
    if "threads" in what:
        # the reference's call shape under load: many threads, ONE query per call (k = 100 = its default limit)
        import threading

        t = eng.open_table(_lib.METRIC_HAMMING, 1, 8)
        t.add_synthetic(8, 100_000_000, 1)
        for nthreads in (1, 4, 16, 64):
            stop = time.perf_counter() + 2.0
            counts = [0] * nthreads

            def loop(i):
                r = np.random.default_rng(i)
                while time.perf_counter() < stop:
                    t.search(r.integers(0, 2**64, size=(1, 1), dtype=np.uint64), None, 100)
                    counts[i] += 1

            ths = [threading.Thread(target=loop, args=(i,)) for i in range(nthreads)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt = time.perf_counter() - t0
            print(f"threads: {nthreads:3d} callers x 1 query/call, k=100, 100M x 64-bit: {sum(counts)/dt:9.0f} queries/s")
        t.drop()
    if "mixed" in what:
        t = eng.open_table(_lib.METRIC_NPHD, 1, 32)
        for nb, n in ((8, 40_000_000), (16, 20_000_000), (24, 10_000_000), (32, 30_000_000)):
            t.add_synthetic(nb, n, 100 + nb, key_base=nb * 10**9)
        for qb in (8, 32):
            nq = 256
            q = rng.integers(0, 2**64, size=(nq, 4), dtype=np.uint64)
            for j in range(4):
                if qb <= 8 * j:
                    q[:, j] = 0
            qn = np.full(nq, qb, dtype=np.uint8)
            dt = timeit(lambda: t.search(q, qn, 10), reps=3, warm=1)
            print(f"mixed NPHD: 100M rows in 4 length segments, {qb*8}-bit queries, nq={nq}, k=10: {dt*1e3:9.3f} ms/call ({nq/dt:8.0f} qps)")
        t.drop()
    print(eng.stats())
    eng.close()


if __name__ == "__main__":
    main()
