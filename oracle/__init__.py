"""
oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's hot path (exact NPHD / Hamming k-NN and the scoring
pipelines around it).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; ``iscc_search_amd`` never does.

Parity status: pinned against every literal known-answer the reference's tests hold for this
boundary (``tests/golden/kat_*.json``).  Mixed-length NPHD is pinned only by
the prose formula ``docs/explanation/similarity-search.md:24-29`` -- "NPHD mixed-length parity:
unpinned by reference tests".  The reference itself cannot be imported here (Python 3.10 vs
``requires-python >=3.11`` and missing third-party wheels: ordinary ModuleNotFoundError).

Two independent implementations are kept so they can check each other:
  * ``nphd_oracle.c``  -- popcount over packed 64-bit words, heap top-k (fast; OpenMP)
  * ``nphd_ref.py``    -- numpy ``unpackbits`` bit arrays and a full lexsort (slow; small cases)
"""

from oracle.clib import load_oracle, oracle_topk, oracle_splitmix64_fill, oracle_num_threads  # noqa: F401
from oracle.nphd_ref import pack_codes, ref_topk, ref_distance_pairs, ref_within, ref_doc_freq, np_within  # noqa: F401
