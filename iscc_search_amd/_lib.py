"""
ctypes binding of ``libisccsearch_hip.so`` (C-ABI: ``include/isccsearch.h``).

The library is the only compute path of this package: if it cannot be loaded, or no gfx950 device
can be opened, the functions here raise -- there is no CPU fallback.
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISCC_HIP_LIB: another build of the same library (kernel experiments); the default is the in-tree product build
LIB_PATH = os.environ.get("ISCC_HIP_LIB") or os.path.join(_HERE, "csrc", "libisccsearch_hip.so")

METRIC_HAMMING = 0
METRIC_NPHD = 1
MAX_BYTES = 32
MAX_K = 4096
COUNT_OVERFLOW = 0xFFFFFFFF   # ISCCSEARCH_COUNT_OVERFLOW: a count the asynchronous device search could not complete
ADD_TRUSTED_UNIQUE = 1
MAX_SCORED_SIMPRINTS = 8192   # ISCCSEARCH_MAX_SCORED_SIMPRINTS: query simprints one isccsearch_simprint_score call takes

# every symbol include/isccsearch.h declares
EXPORTS = (
    "isccsearch_create", "isccsearch_destroy", "isccsearch_last_error", "isccsearch_set_option",
    "isccsearch_stats_get", "isccsearch_table_open", "isccsearch_table_drop", "isccsearch_reserve",
    "isccsearch_size", "isccsearch_add", "isccsearch_remove", "isccsearch_contains", "isccsearch_get",
    "isccsearch_segments", "isccsearch_export", "isccsearch_add_columns",
    "isccsearch_add_synthetic", "isccsearch_search", "isccsearch_search_within", "isccsearch_search_many", "isccsearch_doc_freq", "isccsearch_doc_freq_counted", "isccsearch_get_freq",
    "isccsearch_simprint_score", "isccsearch_simprint_exact",
    "isccsearch_search_device", "isccsearch_search_within_device", "isccsearch_merge_device",
    "isccsearch_search_device_async", "isccsearch_merge_device_after", "isccsearch_merge_many_after", "isccsearch_stream",
)

RECORD_DTYPE = np.dtype(
    [("key_hi", "<u8"), ("key_lo", "<u8"), ("dist_rank", "<u4"), ("hamming", "<u2"), ("prefix_bits", "<u2")]
)
assert RECORD_DTYPE.itemsize == 24
# isccsearch_simprint_result / isccsearch_simprint_chunk (isccsearch_simprint_score)
SIMPRINT_RESULT_DTYPE = np.dtype([("asset", "<u8"), ("score", "<f8"), ("matches", "<u4"), ("first_chunk", "<u4")])
SIMPRINT_CHUNK_DTYPE = np.dtype([("key_lo", "<u8"), ("query", "<u4"), ("hamming", "<u4"), ("freq", "<u4"), ("reserved", "<u4")])
assert SIMPRINT_RESULT_DTYPE.itemsize == 24 and SIMPRINT_CHUNK_DTYPE.itemsize == 24


class Stats(ctypes.Structure):
    _fields_ = [
        ("searches", ctypes.c_uint64),
        ("queries", ctypes.c_uint64),
        ("scan_launches", ctypes.c_uint64),
        ("scan_passes", ctypes.c_uint64),
        ("scan_bytes", ctypes.c_uint64),
        ("scan_ms", ctypes.c_double),
        ("sample_bytes", ctypes.c_uint64),
        ("fallback_queries", ctypes.c_uint64),
        ("queries_per_pass", ctypes.c_uint32),
        ("compute_units", ctypes.c_uint32),
        ("freq_builds", ctypes.c_uint64),
        ("mfma_launches", ctypes.c_uint64),
        ("mfma_pair_words", ctypes.c_uint64),
        ("scan_pair_words", ctypes.c_uint64),
        ("scan_mfma_launches", ctypes.c_uint64),
        ("level_launches", ctypes.c_uint64),
        ("level_pair_words", ctypes.c_uint64),
        ("level_mfma_launches", ctypes.c_uint64),
        ("level_ms", ctypes.c_double),
        ("self_retries", ctypes.c_uint64),
        ("mfma_pack_launches", ctypes.c_uint64),
        ("spec_hits", ctypes.c_uint64),
        ("spec_misses", ctypes.c_uint64),
        ("candidates", ctypes.c_uint64),
        ("candidate_batches", ctypes.c_uint64),
    ]

    def as_dict(self):
        # type: () -> dict
        return {name: getattr(self, name) for name, _ in self._fields_}


class Request(ctypes.Structure):
    """``isccsearch_request``: one search of an ``isccsearch_search_many`` call."""

    _fields_ = [
        ("table", ctypes.c_uint32),
        ("nq", ctypes.c_uint32),
        ("k", ctypes.c_uint32),
        ("max_hamming", ctypes.c_int32),
        # pointers as plain addresses: filled from ndarray.ctypes.data (a typed POINTER cast costs ~2 us per field)
        ("q_words", ctypes.c_void_p),
        ("q_nbytes", ctypes.c_void_p),
        ("out_keys", ctypes.c_void_p),
        ("out_hamming", ctypes.c_void_p),
        ("out_prefix_bits", ctypes.c_void_p),
        ("out_count", ctypes.c_void_p),
        ("status", ctypes.c_int32),
    ]


class MergeRequest(ctypes.Structure):
    """``isccsearch_merge_request``: one merge of an ``isccsearch_merge_many_after`` call."""

    _fields_ = [
        ("n_lists", ctypes.c_uint32),
        ("nq", ctypes.c_uint32),
        ("k", ctypes.c_uint32),
        ("key_words", ctypes.c_int32),
        ("d_records", ctypes.c_void_p),
        ("d_counts", ctypes.c_void_p),
        ("list_stride", ctypes.c_uint64),
        ("count_stride", ctypes.c_uint64),
        ("out_keys", ctypes.c_void_p),
        ("out_hamming", ctypes.c_void_p),
        ("out_prefix_bits", ctypes.c_void_p),
        ("out_count", ctypes.c_void_p),
    ]


_LIB = None


def load_library():
    # type: () -> ctypes.CDLL
    """Load the HIP library and declare its signatures.  Raises RuntimeError when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C iscc_search_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1
    # under the same sonames as /opt/rocm's; whichever is loaded first serves every later DT_NEEDED.
    # If this library pulled in /opt/rocm's copy first, a later `import torch` would initialise against
    # a runtime it was not built for ("No HIP GPUs are available").  So when torch is installed, let it
    # load its runtime first and share it (set ISCC_HIP_NO_TORCH=1 for a torch-free process).
    if not os.environ.get("ISCC_HIP_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}. There is no CPU fallback.") from e
    # array arguments are declared void*: `ptr()` hands over plain addresses.  (ndarray.ctypes.data_as(POINTER(...)) costs 2.6 us
    # per argument -- 16 us of a 185 us single-query search for its six arrays; the address alone 1.2 us, and the result block
    # of a search is ONE allocation with one address, engine._alloc_out.)
    vp = ctypes.c_void_p
    u64p = u32p = u16p = u8p = ctypes.c_void_p
    i, u32, u64 = ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64
    sig = {
        "isccsearch_create": (i, [i, ctypes.POINTER(vp)]),
        "isccsearch_destroy": (i, [vp]),
        "isccsearch_last_error": (ctypes.c_char_p, []),
        "isccsearch_set_option": (i, [vp, ctypes.c_char_p, ctypes.c_int64]),
        "isccsearch_stats_get": (i, [vp, ctypes.POINTER(Stats), i]),
        "isccsearch_table_open": (i, [vp, i, i, i, ctypes.POINTER(ctypes.c_uint32)]),
        "isccsearch_table_drop": (i, [vp, u32]),
        "isccsearch_reserve": (i, [vp, u32, i, u64]),
        "isccsearch_size": (u64, [vp, u32]),
        "isccsearch_add": (i, [vp, u32, u64, u64p, u64p, u8p, u32]),
        "isccsearch_remove": (i, [vp, u32, u64, u64p, u64p]),
        "isccsearch_contains": (i, [vp, u32, u64, u64p, u8p]),
        "isccsearch_get": (i, [vp, u32, u64, u64p, u64p, u8p]),
        "isccsearch_segments": (i, [vp, u32, u64p]),
        "isccsearch_export": (i, [vp, u32, i, u64, u64, u64p, u64p]),
        "isccsearch_add_columns": (i, [vp, u32, i, u64, u64p, u64p, u32]),
        "isccsearch_add_synthetic": (i, [vp, u32, i, u64, u64, u64, u64]),
        "isccsearch_search": (i, [vp, u32, u32, u64p, u8p, u32, u64p, u32p, u16p, u32p]),
        "isccsearch_search_within": (i, [vp, u32, u32, u64p, u8p, u32, u32, u64p, u32p, u16p, u32p]),
        "isccsearch_search_many": (i, [vp, u32, ctypes.POINTER(Request)]),
        "isccsearch_doc_freq": (i, [vp, u32, u32, u64p, u8p, u32, u32p]),
        "isccsearch_doc_freq_counted": (i, [vp, u32, u32, u64p, u8p, u32, u32p, u32p]),
        "isccsearch_get_freq": (i, [vp, u32, u64, u64p, u32, u32p]),
        "isccsearch_simprint_score": (i, [vp, u32, u32, u64p, u32, ctypes.c_int32, ctypes.c_double, u32, ctypes.c_int64, u32, vp, vp, vp, u32p]),
        "isccsearch_simprint_exact": (i, [vp, u32, u32, u64p, u32, u32p, u32, u32, ctypes.c_double, u32, vp, vp, u32p]),
        "isccsearch_search_device": (i, [vp, u32, u32, u64p, u8p, u32, vp, vp]),
        "isccsearch_search_within_device": (i, [vp, u32, u32, u64p, u8p, u32, u32, vp, vp]),
        "isccsearch_merge_device": (i, [vp, u32, u32, u32, i, vp, vp, u64, u64, u64p, u32p, u16p, u32p]),
        "isccsearch_search_device_async": (i, [vp, u32, u32, u64p, u8p, u32, ctypes.c_int32, vp, vp, vp]),
        "isccsearch_merge_device_after": (i, [vp, u32, u32, u32, i, vp, vp, u64, u64, vp, u64p, u32p, u16p, u32p]),
        "isccsearch_merge_many_after": (i, [vp, u32, ctypes.POINTER(MergeRequest), vp]),
        "isccsearch_stream": (vp, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def last_error():
    # type: () -> str
    msg = load_library().isccsearch_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    # type: (int) -> None
    """Map the C-ABI's errno-style codes onto the exceptions the reference's callers expect."""
    if rc == 0:
        return
    msg = last_error() or f"isccsearch error {rc}"
    import errno

    if rc == -errno.EINVAL:
        raise ValueError(msg)
    if rc == -errno.EEXIST:
        raise KeyError(msg)
    if rc == -errno.ENOENT:
        raise LookupError(msg)
    if rc == -errno.ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def ptr(arr, ctype=None):
    """Address of a C-contiguous numpy array for a void* argument (None passes through as NULL).  `ctype` documents the element type."""
    if arr is None:
        return None
    return arr.__array_interface__["data"][0]
